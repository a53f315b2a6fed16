"""How much of a k_trace launch is its tail?  The primary rays of bench.py's camera at 1920x1080 (and a shadow-like any-hit
set: the same rays cut off at 0.9 of their hit distance -> never occluded -> full traversal) are traced in three orders:
natural (8 x 8 pixel tiles like the renderer's queue), descending cost (the items each ray fetched in a first counting
launch: the longest rays start first, the launch ends on short ones) and ascending cost (worst case).  Same rays, same total
work; the differences are scheduling only.  Diagnostic for profiles/r03_experiments.txt."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gfxexp_amd import api, scenes  # noqa: E402
from tools.bvh_quality import camera_rays  # noqa: E402


def main():
    import torch
    w, h = 1920, 1080
    hs = scenes.bench_street(textured=True)
    cam = api.make_camera(w, h, pos=(1.5, 2.2, 52.0), pitch=4.0, yaw=181.5)
    org, dirs = camera_rays(cam, w, h)
    # 8 x 8 tiles, row-major tiles: the order the renderer's wave64 pixel mapping produces
    idx = np.arange(w * h).reshape(h // 8, 8, w // 8, 8).transpose(0, 2, 1, 3).reshape(-1)
    org, dirs = org[idx], dirs[idx]
    ctx = api.Context(0)
    hs.upload(ctx)
    accel = ctx.accel_build()
    n = w * h
    out = torch.zeros(n * 16, dtype=torch.uint8, device="cuda")

    def timed(mode, o, d, reps=10):
        d_org, d_dir = torch.from_numpy(np.ascontiguousarray(o)).cuda(), torch.from_numpy(np.ascontiguousarray(d)).cuda()
        for _ in range(3):
            ctx.trace(accel, mode, d_org.data_ptr(), d_dir.data_ptr(), n, out.data_ptr())
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ctx.trace(accel, mode, d_org.data_ptr(), d_dir.data_ptr(), n, out.data_ptr())
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    def cost(mode, o, d):
        d_org, d_dir = torch.from_numpy(np.ascontiguousarray(o)).cuda(), torch.from_numpy(np.ascontiguousarray(d)).cuda()
        counters = torch.zeros(4, dtype=torch.int64, device="cuda")
        per_ray = torch.zeros(n, dtype=torch.int32, device="cuda")
        ctx.trace(accel, mode, d_org.data_ptr(), d_dir.data_ptr(), n, out.data_ptr(), d_counters=counters.data_ptr(), d_per_ray_items=per_ray.data_ptr())
        torch.cuda.synchronize()
        return per_ray.cpu().numpy().astype(np.int64), out.view(torch.float32).view(n, 4)[:, 0].cpu().numpy().copy()

    res = {}
    items, dist = cost(api.TRACE_CLOSEST, org, dirs)
    for name, mode, o, d, it in [("closest", api.TRACE_CLOSEST, org, dirs, items)]:
        order_desc = np.argsort(-it, kind="stable")
        res[name] = {"items_mean": float(it.mean()), "items_p99": float(np.percentile(it, 99)), "items_max": int(it.max()),
                     "natural_ms": timed(mode, o, d), "descending_cost_ms": timed(mode, o[order_desc], d[order_desc]),
                     "ascending_cost_ms": timed(mode, o[order_desc[::-1]], d[order_desc[::-1]]),
                     "random_order_ms": timed(mode, o[np.random.RandomState(1).permutation(n)], d[np.random.RandomState(1).permutation(n)])}
    # unoccluded any-hit rays: cut every ray at 0.9 of its hit distance (misses keep their length)
    d2 = dirs.copy()
    hit = dist < 1e30
    d2[hit, 3] = dist[hit] * np.float32(0.9)
    it2, _ = cost(api.TRACE_ANY, org, d2)
    order_desc = np.argsort(-it2, kind="stable")
    res["any_unoccluded"] = {"items_mean": float(it2.mean()), "items_p99": float(np.percentile(it2, 99)), "items_max": int(it2.max()),
                             "natural_ms": timed(api.TRACE_ANY, org, d2), "descending_cost_ms": timed(api.TRACE_ANY, org[order_desc], d2[order_desc]),
                             "ascending_cost_ms": timed(api.TRACE_ANY, org[order_desc[::-1]], d2[order_desc[::-1]])}
    # per-ray overhead: the same rays cut off right behind the camera (one item each: the root) and with an empty interval (no item)
    d3 = dirs.copy(); d3[:, 3] = np.float32(1e-3)
    it3, _ = cost(api.TRACE_ANY, org, d3)
    res["any_root_only"] = {"items_mean": float(it3.mean()), "natural_ms": timed(api.TRACE_ANY, org, d3), "closest_natural_ms": timed(api.TRACE_CLOSEST, org, d3)}
    d4 = dirs.copy(); d4[:, 3] = np.float32(-1.0)
    res["empty_interval"] = {"natural_ms": timed(api.TRACE_ANY, org, d4)}
    for frac in (0.25, 0.5):
        d5 = dirs.copy(); d5[hit, 3] = dist[hit] * np.float32(frac)
        it5, _ = cost(api.TRACE_ANY, org, d5)
        res["any_cut_%g" % frac] = {"items_mean": float(it5.mean()), "natural_ms": timed(api.TRACE_ANY, org, d5)}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
