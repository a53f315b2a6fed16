"""Where one band's frame time goes: per-kernel GPU time (HIP events around every launch) against the wall clock of the
frame, for the whole 1080p bench frame and for band 3 of an 8-way split in strip mode (no-op exchange).  JSON lines."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gfxexp_amd import api, scenes, tilesplit  # noqa: E402


def measure(ctx, cam, W, H, band, steps=30):
    import torch
    cfg = api.RestirRenderer.default_config(W, H, api.RENDERER_BIASED)
    cfg.camera = cam
    cfg.rowBegin, cfg.rowEnd = band
    os.environ["GFX_SERIAL_FRAMES"] = "1"
    r = api.RestirRenderer(ctx, cfg)
    del os.environ["GFX_SERIAL_FRAMES"]
    if band != (0, 0):
        r.set_exchange(lambda stream, d: None, 0)
    for _ in range(5):
        r.render_frame()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        r.render_frame()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / steps * 1e3
    # host time to enqueue one frame (no sync)
    t0 = time.perf_counter()
    for _ in range(steps):
        r.render_frame()
    enqueue = (time.perf_counter() - t0) / steps * 1e3
    torch.cuda.synchronize()
    ctx.timing_enable(True)
    for _ in range(8):
        r.render_frame()
    torch.cuda.synchronize()
    timings = ctx.timing_collect()
    ctx.timing_enable(False)
    per = {k: round(ms / 8, 4) for k, (ms, calls) in sorted(timings.items(), key=lambda kv: -kv[1][0])}
    launches = sum(calls for _, (ms, calls) in timings.items()) / 8
    r.close()
    return {"band": list(band), "wall_ms": round(wall, 4), "host_enqueue_ms": round(enqueue, 4), "kernel_sum_ms": round(sum(per.values()), 4),
            "timed_launches_per_frame": launches, "kernels_ms": per}


def main():
    W, H = 1920, 1080
    ctx = api.Context(0)
    scenes.bench_street(textured="--textured" in sys.argv).upload(ctx)
    cam = api.make_camera(W, H, pos=(1.5, 2.2, 52.0), pitch=4.0, yaw=181.5)
    print(json.dumps(measure(ctx, cam, W, H, (0, 0))))
    for rank in (0, 3):
        print(json.dumps(measure(ctx, cam, W, H, tilesplit.band_rows(H, 8)[rank])))


if __name__ == "__main__":
    main()
