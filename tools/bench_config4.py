"""BASELINE configs[4] on one GPU: ReSTIR DI unbiased + 2048x1024 environment map (analytic sky + sun) on the
street stand-in, 1920x1080.  One JSON line with the frame time and the per-kernel split."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gfxexp_amd import api  # noqa: E402
from gfxexp_amd import scenes  # noqa: E402


def main():
    import torch
    W, H = 1920, 1080
    ctx = api.Context(0)
    scenes.bench_street().upload(ctx)
    cfg = api.RestirRenderer.default_config(W, H, api.RENDERER_UNBIASED)
    cfg.camera = api.make_camera(W, H, pos=(1.5, 2.2, 52.0), pitch=4.0, yaw=181.5)
    r = api.RestirRenderer(ctx, cfg)
    sky = api.env_make_sky(2048, 1024)
    r.set_env(sky, 2048, 1024, 0.6, 0.4)
    for _ in range(4):
        r.render_frame()
    torch.cuda.synchronize()
    ctx.timing_enable(True)
    ctx.timing_collect()
    steps = 20
    t0 = time.perf_counter()
    for _ in range(steps):
        r.render_frame()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    kern = {k: round(v[0] / steps, 4) for k, v in sorted(ctx.timing_collect().items(), key=lambda kv: -kv[1][0])}
    print(json.dumps({"workload": "configs[4] on 1 GPU: unbiased ReSTIR DI + env map 2048x1024, street stand-in, 1920x1080",
                      "ms_per_frame": round(dt * 1e3, 4), "Mpaths_per_s": round(W * H / dt / 1e6, 2), "kernels_ms_per_frame": kern}))


if __name__ == "__main__":
    main()
