#!/usr/bin/env python
"""Where a wave of k_nrc_infer_staged spends its clock cycles (GFX_CYC marks, wave-level s_memtime; the marks themselves cost ~10 %):
one full-HD frame's inference batch (2.1 M uniform queries).  Needs the profiling build:
    python gfxexp_amd/build.py --variant laneprof GFX_LANE_PROFILE
    GFX_LIB=$PWD/gfxexp_amd/variants/libgfxexp_laneprof.so python tools/nrc_infer_profile.py
One JSON line (profiles/r06_nrc_infer_profile.json)."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gfxexp_amd import api  # noqa: E402

SECTIONS = {0: "positions of the pass's queries (three strided loads per tile)", 1: "barrier: the block's other waves finish the level before",
            2: "level table global -> LDS: DMA issued, landed, second barrier", 3: "hash features of the level (index arithmetic, 8 ds_read_b32, blend, lane exchange)",
            4: "weights into LDS (two barriers)", 5: "tile inputs through LDS, one-blob + identity features, operand assembly", 6: "layers (MFMA, ReLU, pack) + output",
            7: "kernel entry / exit"}


def main():
    import torch
    ctx = api.Context(0)
    net = api.NeuralRadianceCache(ctx, api.NRC_HASH_GRID, 2)
    L = api.lib()
    w, h = 1920, 1080
    n = ((w * h + ((w + 7) // 8) * ((h + 7) // 8) + 127) // 128) * 128
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    x = torch.rand((n, 14), generator=g, device="cuda", dtype=torch.float32)
    x[:, 3:8] = x[:, 3:8] * 6 - 3
    y = torch.zeros((n, 3), device="cuda", dtype=torch.float32)
    stream = torch.cuda.current_stream().cuda_stream
    out = (C.c_uint64 * 64)()
    for _ in range(3):
        net.infer(x.data_ptr(), n, y.data_ptr(), stream)
    torch.cuda.synchronize()
    assert L.gfx_debug_nrc_profile(out, 1) == 0
    ctx.timing_enable(True)
    ctx.timing_collect()
    reps = 10
    for _ in range(reps):
        net.infer(x.data_ptr(), n, y.data_ptr(), stream)
    torch.cuda.synchronize()
    ms = {k: round(v[0] / reps, 4) for k, v in ctx.timing_collect().items()}
    assert L.gfx_debug_nrc_profile(out, 1) == 0
    cyc = {k: int(out[32 + k]) for k in SECTIONS}
    total = sum(cyc.values())
    waves = reps * 256 * 12
    print(json.dumps({"queries": n, "ms_per_launch_with_marks": ms, "waves_per_launch": 256 * 12, "cycles_per_wave": round(total / waves),
                      "sections": {SECTIONS[k]: {"share": round(v / max(1, total), 4), "cycles_per_wave": round(v / waves)} for k, v in cyc.items()}}))


if __name__ == "__main__":
    main()
