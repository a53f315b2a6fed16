#!/usr/bin/env python
"""The NRC frame (BASELINE configs[3] stand-in: street, 1920x1080, hash grid, 2 hidden layers, training on) with the four
training steps overlapped with the next frame (the default) and serial on the caller's stream (GFX_NRC_SERIAL_TRAINING=1):
wall-clock frame time and the per-kernel HIP-event table of each, to tell overlap inflation (kernels that share the GPU with
another stream read longer than they are) from a real slowdown; a third line with the frames not pipelined either (the next
frame's G-buffer pass otherwise runs under this frame's inference).  One JSON line per mode.

    python tools/bench_nrc_frame.py [--steps K] [--textured]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gfxexp_amd import api, scenes  # noqa: E402


def run(ctx, cfg, steps, serial, serial_frames=False):
    import torch
    if serial:
        os.environ["GFX_NRC_SERIAL_TRAINING"] = "1"
    if serial_frames:
        os.environ["GFX_SERIAL_FRAMES"] = "1"       # nor the next frame's G-buffer pass under this frame's inference
    r = api.NrcRenderer(ctx, cfg)
    os.environ.pop("GFX_NRC_SERIAL_TRAINING", None)
    os.environ.pop("GFX_SERIAL_FRAMES", None)
    for _ in range(6):
        r.render_frame()
    r.network()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        r.render_frame()
    r.network()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    ctx.timing_enable(True)
    ctx.timing_collect()
    n = 10
    for _ in range(n):
        r.render_frame()
    r.network()
    torch.cuda.synchronize()
    k = {a: round(b[0] / n, 4) for a, b in sorted(ctx.timing_collect().items(), key=lambda kv: -kv[1][0])}
    ctx.timing_enable(False)
    stats = r.stats()
    r.close()
    train = sum(v for a, v in k.items() if a in ("nrc_train_fwd_bwd", "nrc_optimizer", "nrc_pack"))
    mode = "overlapped with the next frame (second stream)"
    if serial:
        mode = "serial (one stream)" + (", frames not pipelined either: the kernel table holds undisturbed durations" if serial_frames else "")
    return {"training": mode, "frame_ms": round(ms, 4),
            "mpaths_s": round(1920 * 1080 / ms / 1e3, 1), "kernels_ms": k, "kernel_sum_ms": round(sum(k.values()), 4),
            "training_kernels_ms": round(train, 4), "other_kernels_ms": round(sum(k.values()) - train, 4), "stats": stats}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--textured", action="store_true")
    ap.add_argument("--nee", default="lights", choices=["lights", "regir", "restir"], help="next-event estimation of the tracer (gfxh_nrc_config::neeSampler)")
    args = ap.parse_args()
    hs = scenes.bench_street(textured=args.textured)
    ctx = api.Context(0)
    hs.upload(ctx)
    w, h = 1920, 1080
    cfg = api.NrcRenderer.default_config(w, h, hs.bounds())
    cfg.camera = api.make_camera(w, h, pos=(2.0, 5.0, 26.0), pitch=6.0, yaw=184.0)
    cfg.neeSampler = {"lights": api.NRC_NEE_LIGHTS, "regir": api.NRC_NEE_REGIR, "restir": api.NRC_NEE_RESTIR}[args.nee]
    for serial, serial_frames in ((False, False), (True, False), (True, True)):
        out = run(ctx, cfg, args.steps, serial, serial_frames)
        out["nee"] = args.nee
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
