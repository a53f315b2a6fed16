"""The fixed cost of a k_trace launch: gfx_trace over N queue entries with EMPTY intervals (retired at the fetch: no traversal at all)
and with real primary rays of the bench camera, N from one ticket batch to a full frame.  HIP events around 20 launches.  JSON lines."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gfxexp_amd import api, scenes  # noqa: E402


def main():
    import torch
    ctx = api.Context(0)
    scenes.bench_street(textured=False).upload(ctx)
    accel = ctx.accel_build()
    W, H = 1920, 1080
    rng = np.random.default_rng(1)
    n_max = W * H
    # primary rays of the bench camera (pinhole), row-major
    cam_pos = np.array([1.5, 2.2, 52.0], np.float32)
    ys, xs = np.mgrid[0:H, 0:W]
    d = np.stack([(xs + 0.5) / W * 2 - 1, -((ys + 0.5) / H * 2 - 1) * H / W, -np.ones_like(xs, np.float32) * 1.2], -1).reshape(-1, 3).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    org = np.zeros((n_max, 4), np.float32); org[:, :3] = cam_pos
    dirs = np.zeros((n_max, 4), np.float32); dirs[:, :3] = d; dirs[:, 3] = 1e10
    d_org = torch.from_numpy(org).cuda()
    d_dir = torch.from_numpy(dirs).cuda()
    d_empty = d_dir.clone(); d_empty[:, 3] = -1.0
    out = torch.zeros(n_max * 4, dtype=torch.int32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    for mode, name in ((api.TRACE_CLOSEST, "closest"), (api.TRACE_ANY, "any")):
        for n in (64, 4096, 65536, 262144, n_max):
            for kind, dd in (("empty intervals", d_empty), ("primary rays", d_dir)):
                for _ in range(3):
                    ctx.trace(accel, mode, d_org.data_ptr(), dd.data_ptr(), n, out.data_ptr(), 0, stream=stream)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    ctx.trace(accel, mode, d_org.data_ptr(), dd.data_ptr(), n, out.data_ptr(), 0, stream=stream)
                e1.record()
                torch.cuda.synchronize()
                print(json.dumps({"mode": name, "rays": n, "kind": kind, "us_per_launch": round(e0.elapsed_time(e1) / 20 * 1e3, 1)}), flush=True)


if __name__ == "__main__":
    main()
