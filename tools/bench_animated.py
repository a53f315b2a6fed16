"""Cost of an animated light in the bench frame: the street stand-in plus the reference command line's moving
rectangle light (restir_di_main.cpp:7-12: begin/end position, cosine ease, 5 s period), 1920x1080, ReSTIR DI biased.
Per frame: gfx_instance_set_transform + in-place BVH update + frame.  One JSON line."""
import json
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gfxexp_amd import api  # noqa: E402
from gfxexp_amd import scenes  # noqa: E402


def light_transform(t_seconds):
    t = 0.5 - 0.5 * math.cos(2 * math.pi * (t_seconds % 5.0) / 5.0)      # InstanceController::updateBody
    pos = ((1 - t) * -6.0 + t * 7.0, 4.5, (1 - t) * 30.0 + t * 12.0)
    return api.make_transform(pitch=-90.0, yaw=(1 - t) * 150.0 + t * 30.0, pos=pos)


def run(animated, steps=40, declare=True):
    import torch
    W, H = 1920, 1080
    hs = scenes.bench_street()
    light = hs.add_rectangle(1.5, 1.5, (60, 60, 60))
    slot = hs.add_instance(light, light_transform(0.0))
    ctx = api.Context(0)
    hs.upload(ctx)
    if animated and declare:
        ctx.instance_set_dynamic(slot)
    cfg = api.RestirRenderer.default_config(W, H, api.RENDERER_BIASED)
    cfg.camera = api.make_camera(W, H, pos=(1.5, 2.2, 52.0), pitch=4.0, yaw=181.5)
    r = api.RestirRenderer(ctx, cfg)
    frame = 0

    def step():
        nonlocal frame
        if animated:
            ctx.instance_set_transform(slot, light_transform(frame / 60.0))
            r.rebuild_accel()
        r.render_frame()
        frame += 1

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3, ctx.accel_stats(r.accel())


def main():
    static_ms, _ = run(False)
    anim_ms, stats = run(True)
    print(json.dumps({"workload": "bench street + moving rectangle light, 1920x1080, ReSTIR DI biased", "static_ms_per_frame": round(static_ms, 4),
                      "animated_ms_per_frame": round(anim_ms, 4), "Mpaths_per_s_animated": round(1920 * 1080 / anim_ms / 1e3, 2), "bvh": stats}))


if __name__ == "__main__":
    main()
