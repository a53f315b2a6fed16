"""NRC network micro-benchmark (BASELINE.json configs[3] shapes): one frame's worth of inference
(1920x1080 + one query per 8x8 training tile, rounded up to 128: neural_radiance_caching_main.cpp:2304-2316)
and 4 training steps of 16 384 records (:2350-2365).  Prints one JSON line.

    python tools/bench_nrc.py [--steps K] [--hidden 2|5] [--encoding hash|tri]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gfxexp_amd import api  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--hidden", type=int, default=2)
    ap.add_argument("--encoding", default="hash")
    args = ap.parse_args()
    import torch
    ctx = api.Context(0)
    enc = api.NRC_HASH_GRID if args.encoding == "hash" else api.NRC_TRIANGLE_WAVE
    net = api.NeuralRadianceCache(ctx, enc, args.hidden)
    w, h = 1920, 1080
    n_inf = ((w * h + ((w + 7) // 8) * ((h + 7) // 8) + 127) // 128) * 128
    n_train = 16384
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    x = torch.rand((n_inf, 14), generator=g, device="cuda", dtype=torch.float32)
    x[:, 3:8] = x[:, 3:8] * 6 - 3
    y = torch.zeros((n_inf, 3), device="cuda", dtype=torch.float32)
    t = torch.rand((4 * n_train, 3), generator=g, device="cuda", dtype=torch.float32)
    stream = torch.cuda.current_stream().cuda_stream

    def frame():
        net.infer(x.data_ptr(), n_inf, y.data_ptr(), stream)
        for k in range(4):
            net.train(x[k * n_train:].data_ptr(), t[k * n_train:].data_ptr(), n_train, False, stream)

    ctx.timing_enable(True)
    for _ in range(args.warmup):
        frame()
    torch.cuda.synchronize()
    ctx.timing_collect()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        frame()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    timings = {k: v[0] / args.steps for k, v in ctx.timing_collect().items()}
    mats = args.hidden * 4096 + 1024
    flops_inf = 2.0 * mats * n_inf
    ms_inf = timings.get("nrc_infer", float("nan"))
    out = {
        "metric": "NRC network: inference queries/s (fully fused 64-wide MLP, bf16 MFMA)",
        "value": n_inf / (ms_inf * 1e-3) / 1e6, "unit": "Mqueries/s", "n_gpus": 1, "steps": args.steps,
        "ms_per_frame_network": dt * 1e3, "kernels_ms_per_frame": timings,
        "config": {"workload": "configs[3] network shapes", "inference_queries": n_inf, "training": "4 x 16384",
                   "encoding": args.encoding, "hidden_layers": args.hidden},
        "roofline": {"bound": "mfma", "achieved": flops_inf / (ms_inf * 1e-3) / 1e12, "peak": 2500.0, "unit": "TFLOP/s",
                     "frac": flops_inf / (ms_inf * 1e-3) / 1e12 / 2500.0, "flops_per_query": 2 * mats,
                     "note": "the kernel also gathers 128 x 4 B hash-grid corners per query; see profiles/ for MFMA busy cycles"},
        "dtype": "bf16 (fp32 accumulate, fp32 master weights)", "data": "synthetic",
    }
    print(json.dumps(out))


if __name__ == "__main__":
    main()
