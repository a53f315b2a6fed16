"""Where the time of ONE rank's band frame goes on the GPU's clock: kernel start / end times of band 4 of 8 of the 1920x1080 bench frame
(no-op exchange callback, pipelined frames as bench.py --gpus N runs them) from a rocprofv3 kernel trace.
  rocprofv3 --kernel-trace --output-format csv -d DIR -- python tools/band_timeline.py run [--config4] [--bands 8] [--frames 40]
  python tools/band_timeline.py read DIR        -> one JSON object: per kernel mean duration, and per frame the busy time of the frame's own
                                                    stream, its idle gaps (no kernel of ANY stream running) and the frame period"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def arg(name, default):
    return sys.argv[sys.argv.index(name) + 1] if name in sys.argv else default


def run():
    import torch
    from gfxexp_amd import api, scenes, tilesplit
    W, H = 1920, 1080
    nb, frames = int(arg("--bands", "8")), int(arg("--frames", "40"))
    config4 = "--config4" in sys.argv
    ctx = api.Context(0)
    scenes.bench_street(textured=True).upload(ctx)
    cam = api.make_camera(W, H, pos=(1.5, 2.2, 52.0), pitch=4.0, yaw=181.5)
    cfg = api.RestirRenderer.default_config(W, H, api.RENDERER_UNBIASED if config4 else api.RENDERER_BIASED)
    cfg.camera = cam
    cfg.rowBegin, cfg.rowEnd = tilesplit.band_rows(H, nb)[nb // 2]
    cfg.enableBumpMapping = 1
    r = api.RestirRenderer(ctx, cfg)
    if config4:
        r.set_env(api.env_make_sky(2048, 1024), 2048, 1024, 0.6, 0.4)
    r.set_exchange(lambda stream, d: None, 0)
    for _ in range(frames):
        r.render_frame()
    torch.cuda.synchronize()
    r.close()


def short(name):
    n = name.split("(")[0].replace("void ", "").replace("gfx::", "")
    return n[:60]


def read(d):
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                rows.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]), short(row["Kernel_Name"]), row.get("Queue_Id", ""), row.get("Stream_Id", "")))
    rows.sort()
    # frames: one k_initial_fused per frame; steady state = the last 20
    starts = [i for i, r in enumerate(rows) if r[2].startswith("k_initial_fused")]
    starts = starts[-21:]
    per_kernel, periods, idle, frames = {}, [], [], []
    for a, b in zip(starts[:-1], starts[1:]):
        seg = rows[a:b]
        t0, t1 = seg[0][0], rows[b][0]
        periods.append((t1 - t0) / 1e3)
        # idle = time in [t0, t1) covered by no kernel at all
        ev = sorted((max(s, t0), min(e, t1)) for s, e, *_ in rows[max(0, a - 8):b + 8] if e > t0 and s < t1)
        covered, cur = 0, t0
        for s, e in ev:
            if e > cur:
                covered += e - max(s, cur)
                cur = e
        idle.append((t1 - t0 - covered) / 1e3)
        for s, e, n, q, st in seg:
            per_kernel.setdefault(n, []).append((e - s) / 1e3)
        frames.append([(n, round((s - t0) / 1e3, 1), round((e - s) / 1e3, 1), q, st) for s, e, n, q, st in seg])
    out = {"frames": len(periods), "frame_period_us": round(sum(periods) / len(periods), 1), "gpu_idle_us_per_frame": round(sum(idle) / len(idle), 1),
           "kernel_us": {k: {"mean": round(sum(v) / len(v), 1), "per_frame": round(len(v) / len(periods), 2)} for k, v in sorted(per_kernel.items(), key=lambda kv: -sum(kv[1]))},
           "one_frame (kernel, start us, duration us, queue, stream)": frames[-1]}
    print(json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "read":
        read(sys.argv[2])
    else:
        run()
