"""Node visits per primary ray: the reference-style SAH + spatial-split BVH8 (oracle restatement, CPU)
versus the GPU LBVH -> BVH8 of the product on the same rays.  Diagnostic, prints one JSON line."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gfxexp_amd import api  # noqa: E402
from tests import util  # noqa: E402


def main():
    import torch
    which = sys.argv[1] if len(sys.argv) > 1 else "small"
    hs = util.bench_street() if which == "bench" else util.small_street()
    w, h = 480, 270
    org, dirs = util.pinhole_rays(w, h, (2.0, 5.0, 26.0), (0.0, 3.0, 0.0), 50.0)
    ctx = api.Context(0)
    hs.upload(ctx)
    accel = ctx.accel_build()
    n = w * h
    d_org, d_dir = torch.from_numpy(org).cuda(), torch.from_numpy(dirs).cuda()
    out = torch.zeros(n * 16, dtype=torch.uint8, device="cuda")
    ctx.counters_enable(True)
    ctx.counters_read(True)
    ctx.trace(accel, api.TRACE_CLOSEST, d_org.data_ptr(), d_dir.data_ptr(), n, out.data_ptr())
    torch.cuda.synchronize()
    c = ctx.counters_read(True)
    t0 = time.time()
    osc = util.feed_oracle(hs)
    build_s = time.time() - t0
    _, stats = osc.trace(3, org, dirs, want_stats=True)
    print(json.dumps({"scene": which, "rays": n, "gpu_lbvh_nodes_per_ray": c["nodeFetches"] / n, "gpu_tris_per_ray": c["triFetches"] / n,
                      "cpu_sah_nodes_per_ray": int(stats[0]) / n, "cpu_sah_tris_per_ray": int(stats[1]) / n, "cpu_build_s": build_s,
                      "accel_stats": ctx.accel_stats(accel) if hasattr(ctx, "accel_stats") else None}))


if __name__ == "__main__":
    main()
