"""Streaming-copy rate of the GPU box (SURVEY 8d asks for a measured HBM peak next to the datasheet's 8 TB/s): the library's
gfx_stream_copy (16-byte non-temporal loads / stores per lane, what bench.py's roofline.peak_measured times) next to the torch
byte copy rounds 1-3 used, on buffers far larger than the 256-MiB Infinity Cache; read + write bytes per second."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gfxexp_amd import api  # noqa: E402


def timed(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        fn()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def main():
    ctx = api.Context(0)
    out = {}
    stream = torch.cuda.current_stream().cuda_stream
    for gib in (1, 4):
        n = gib << 30
        a = torch.empty(n, dtype=torch.uint8, device="cuda")
        b = torch.empty(n, dtype=torch.uint8, device="cuda")
        a.fill_(1)
        out[f"gfx_stream_copy_{gib}GiB_GBps"] = round(2 * n / timed(lambda: ctx.stream_copy(b.data_ptr(), a.data_ptr(), n, stream), 10) / 1e9, 1)
        out[f"gfx_stream_read_only_{gib}GiB_GBps"] = round(n / timed(lambda: ctx.stream_copy(0, a.data_ptr(), n, stream), 10) / 1e9, 1)
        out[f"torch_copy_{gib}GiB_GBps"] = round(2 * n / timed(lambda: b.copy_(a), 10) / 1e9, 1)
        del a, b
        torch.cuda.empty_cache()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
