"""One scheduling knob of the context against the band size: wall time per frame and per-kernel GPU time of ONE rank's band of the
1920x1080 bench frame (strip mode, no-op exchange, pipelined frames as bench.py --gpus N runs them) for every value of a
gfx_tunable_set knob.  JSON lines.
usage: band_sweep.py --knob trace_refill --values 4,8,16 [--plain] [--bands 8,4,2,1] [--ranks all|mid] [--serial]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gfxexp_amd import api, scenes, tilesplit  # noqa: E402


def arg(name, default):
    return sys.argv[sys.argv.index(name) + 1] if name in sys.argv else default


def measure(ctx, cam, W, H, band, steps=30, kernels=True):
    import torch
    config4 = "--config4" in sys.argv       # BASELINE configs[4]: unbiased estimator + 2048 x 1024 environment map
    cfg = api.RestirRenderer.default_config(W, H, api.RENDERER_UNBIASED if config4 else api.RENDERER_BIASED)
    cfg.camera = cam
    cfg.rowBegin, cfg.rowEnd = band
    cfg.enableBumpMapping = int("--plain" not in sys.argv)
    if "--serial" in sys.argv:
        os.environ["GFX_SERIAL_FRAMES"] = "1"
    r = api.RestirRenderer(ctx, cfg)
    os.environ.pop("GFX_SERIAL_FRAMES", None)
    if config4:
        r.set_env(api.env_make_sky(2048, 1024), 2048, 1024, 0.6, 0.4)
    if band != (0, 0):
        r.set_exchange(lambda stream, d: None, 0)
    for _ in range(6):
        r.render_frame()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        r.render_frame()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / steps * 1e3
    per = {}
    if kernels:
        ctx.timing_enable(True)
        for _ in range(8):
            r.render_frame()
        torch.cuda.synchronize()
        timings = ctx.timing_collect()
        ctx.timing_enable(False)
        per = {k: round(ms / 8, 4) for k, (ms, calls) in sorted(timings.items(), key=lambda kv: -kv[1][0])}
    r.close()
    return round(wall, 4), per


def main():
    W, H = 1920, 1080
    ctx = api.Context(0)
    scenes.bench_street(textured="--plain" not in sys.argv).upload(ctx)
    cam = api.make_camera(W, H, pos=(1.5, 2.2, 52.0), pitch=4.0, yaw=181.5)
    band_counts = [int(x) for x in arg("--bands", "8,4,2,1").split(",")]
    knob = arg("--knob", "trace_refill")
    values = [int(x) for x in arg("--values", "8").split(",")]
    ranks = arg("--ranks", "mid")
    for n in band_counts:
        bands = [(0, 0)] if n == 1 else tilesplit.band_rows(H, n)
        chosen = bands if ranks == "all" else [bands[(len(bands) - 1) // 2 + (1 if n == 8 else 0)] if n > 1 else bands[0]]
        for v in values:
            ctx.tunable_set(knob, v)
            walls, kern = [], None
            for b in chosen:
                w, per = measure(ctx, cam, W, H, b, kernels=(b == chosen[-1]))
                walls.append(w)
                kern = per
            print(json.dumps({"bands": n, "rows": [list(b) for b in chosen], knob: v, "wall_ms": walls, "worst_ms": max(walls), "kernels_ms_last": kern}), flush=True)


if __name__ == "__main__":
    main()
