#!/usr/bin/env python
"""Where a wave of k_nrc_train spends its clock cycles (GFX_CYC marks, wave-level s_memtime): 4 training steps of 16 384 records,
uniform-random and clustered positions.  Needs the profiling build:
    python gfxexp_amd/build.py --variant laneprof GFX_LANE_PROFILE
    GFX_LIB=$PWD/gfxexp_amd/variants/libgfxexp_laneprof.so python tools/nrc_train_profile.py
One JSON line per record distribution."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gfxexp_amd import api  # noqa: E402

SECTIONS = {0: "inputs + encoding (128 hash-grid gathers per lane, one-blob)", 1: "hidden layers forward (fragments from L2, MFMA, activations to LDS twice)",
            2: "output layer, loss, loss gradient", 3: "backward: dW (MFMA over the batch, partials to HBM) and delta", 4: "hash-grid gradient scatter (128 packed atomics per lane)",
            7: "kernel entry / exit"}


def main():
    import torch
    ctx = api.Context(0)
    net = api.NeuralRadianceCache(ctx, api.NRC_HASH_GRID, 2)
    L = api.lib()
    n = 16384
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    stream = torch.cuda.current_stream().cuda_stream
    for name in ("uniform", "clustered"):
        x = torch.rand((4 * n, 14), generator=g, device="cuda", dtype=torch.float32)
        if name == "clustered":                       # positions on a few surfaces: a thin slab + a small box, as visible geometry is
            x[:, 1] = 0.02 * x[:, 1] + 0.3
            x[: 2 * n, 0] = 0.1 * x[: 2 * n, 0] + 0.45
        x[:, 3:8] = x[:, 3:8] * 6 - 3
        t = torch.rand((4 * n, 3), generator=g, device="cuda", dtype=torch.float32)
        out = (C.c_uint64 * 64)()
        for _ in range(2):
            for k in range(4):
                net.train(x[k * n:].data_ptr(), t[k * n:].data_ptr(), n, False, stream)
        torch.cuda.synchronize()
        assert L.gfx_debug_nrc_profile(out, 1) == 0
        ctx.timing_enable(True)
        ctx.timing_collect()
        reps = 5
        for _ in range(reps):
            for k in range(4):
                net.train(x[k * n:].data_ptr(), t[k * n:].data_ptr(), n, False, stream)
        torch.cuda.synchronize()
        ms = {k: round(v[0] / (4 * reps), 4) for k, v in ctx.timing_collect().items()}
        ctx.timing_enable(False)
        assert L.gfx_debug_nrc_profile(out, 1) == 0
        cyc = {k: int(out[32 + k]) for k in SECTIONS}
        total = sum(cyc.values())
        waves = 4 * reps * (n // 64)
        print(json.dumps({"records": name, "ms_per_step": ms, "cycles_per_wave": round(total / waves),
                          "sections": {SECTIONS[k]: {"share": round(v / max(1, total), 4), "cycles_per_wave": round(v / waves)} for k, v in cyc.items()}}), flush=True)


if __name__ == "__main__":
    main()
