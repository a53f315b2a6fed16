"""Tree quality on the bench frame's primary rays: the reference-style SAH + spatial-split BVH8 (oracle
restatement, CPU; common/bvh_builder.cpp:656-1125) versus the product's GPU tree, same rays.

    python tools/bvh_quality.py [bench|small|bench-cluttered] [out.json] [--no-sah]

Rays: the pinhole rays of bench.py's camera (restir_di_shared.h:51-59 camera model) at 480x270.  The oracle counts
node fetches / triangle tests with its own traversal (distance-sorted children, bvh_builder.cpp:1272-1649); the
product counts with the counting instantiation of k_trace.  The JSON is what bench.py's
`roofline.frac_sah_normalised` reads from profiles/r02_bvh_quality.json.  Diagnostic: imports tests/ and oracle/."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gfxexp_amd import api  # noqa: E402
from tests import util  # noqa: E402


def camera_rays(cam, w, h, tmax=np.float32(3.0e38)):
    ori = np.array(list(cam.orientation), np.float32).reshape(3, 3)
    vh = np.float32(2 * np.tan(np.float32(cam.fovY) * np.float32(0.5)))
    vw = np.float32(cam.aspect) * vh
    xs = (np.arange(w, dtype=np.float32) + np.float32(0.5)) / np.float32(w)
    ys = (np.arange(h, dtype=np.float32) + np.float32(0.5)) / np.float32(h)
    X, Y = np.meshgrid(xs, ys)
    local = np.stack([vw * (np.float32(0.5) - X), vh * (np.float32(0.5) - Y), np.ones_like(X)], -1).reshape(-1, 3)
    d = local @ ori.T
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    n = w * h
    org = np.zeros((n, 4), np.float32)
    org[:, :3] = np.array(list(cam.position), np.float32)
    dirs = np.zeros((n, 4), np.float32)
    dirs[:, :3] = d
    dirs[:, 3] = tmax
    return org, dirs


def main():
    import torch
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    which = args[0] if len(args) > 0 else "small"
    out_path = args[1] if len(args) > 1 else None
    from gfxexp_amd import scenes
    hs = scenes.bench_street(cluttered=True) if which == "bench-cluttered" else util.bench_street() if which == "bench" else util.small_street()
    w, h = 480, 270
    cam = api.make_camera(w, h, pos=(1.5, 2.2, 52.0), pitch=4.0, yaw=181.5) if which.startswith("bench") else \
        api.make_camera(w, h, pos=(2.0, 5.0, 26.0), pitch=4.0, yaw=180.0)
    org, dirs = camera_rays(cam, w, h)
    ctx = api.Context(0)
    hs.upload(ctx)
    accel = ctx.accel_build()
    n = w * h
    d_org, d_dir = torch.from_numpy(org).cuda(), torch.from_numpy(dirs).cuda()
    out = torch.zeros(n * 16, dtype=torch.uint8, device="cuda")
    counters = torch.zeros(4, dtype=torch.int64, device="cuda")
    per_ray = torch.zeros(n, dtype=torch.int32, device="cuda")
    ctx.trace(accel, api.TRACE_CLOSEST, d_org.data_ptr(), d_dir.data_ptr(), n, out.data_ptr(), d_counters=counters.data_ptr(),
              d_per_ray_items=per_ray.data_ptr())
    torch.cuda.synchronize()
    c = counters.cpu().numpy()
    items = per_ray.cpu().numpy().astype(np.int64)
    assert items.sum() == int(c[0]) + int(c[1])
    edges = [0, 1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1 << 30]
    hist = np.histogram(items, bins=edges)[0]
    # rows of 8 image rows: where in the frame the expensive rays are (the band split cuts along these)
    by_rows = items.reshape(h, w).reshape(-1, 10 if h % 10 == 0 else 1, w).mean(axis=(1, 2)) if h % 10 == 0 else items.reshape(h, w).mean(axis=1)
    histogram = {"items_per_ray": {"mean": float(items.mean()), "p50": float(np.percentile(items, 50)), "p90": float(np.percentile(items, 90)),
                                   "p99": float(np.percentile(items, 99)), "p99.9": float(np.percentile(items, 99.9)), "max": int(items.max())},
                 "bins": {("%d-%d" % (edges[k], edges[k + 1] - 1) if k + 2 < len(edges) else ">=%d" % edges[k]): int(hist[k]) for k in range(len(hist))},
                 "mean_items_by_band_of_rows": [round(float(x), 2) for x in by_rows]}
    hits = out.view(torch.float32).view(n, 4)[:, 0].cpu().numpy()
    if "--no-sah" in sys.argv:
        print(json.dumps({"scene": which, "rays": n, "gpu_tree": {"nodes_per_ray": float(c[0]) / n, "tris_per_ray": float(c[1]) / n,
                                                                  "accel": ctx.accel_stats(accel)}, "histogram": histogram}))
        return
    t0 = time.time()
    osc = util.feed_oracle(hs)
    build_s = time.time() - t0
    _, stats = osc.trace(3, org, dirs, want_stats=True)
    res = {"histogram": histogram,"scene": which, "rays": n, "camera": "bench.py camera, 480x270 primary rays" if which.startswith("bench") else "small street",
           "hit_fraction": float(np.mean(hits < 1e30)),
           "gpu_tree": {"nodes_per_ray": float(c[0]) / n, "tris_per_ray": float(c[1]) / n, "accel": ctx.accel_stats(accel),
                        "builder": os.environ.get("GFX_BVH_VARIANT", "default")},
           "sah_tree": {"nodes_per_ray": int(stats[0]) / n, "tris_per_ray": int(stats[1]) / n, "cpu_build_s": round(build_s, 2),
                        "builder": "oracle restatement of common/bvh_builder.cpp (binned SAH + spatial splits), distance-sorted traversal"}}
    line = json.dumps(res)
    print(line)
    if out_path:
        with open(out_path, "w") as f:
            f.write(json.dumps(res, indent=1) + "\n")


if __name__ == "__main__":
    main()
