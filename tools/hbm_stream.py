"""Streaming-copy rate of the GPU box (SURVEY 8d asks for a measured HBM peak next to the datasheet's 8 TB/s).
torch copy of a buffer far larger than the 256 MiB Infinity Cache; read + write bytes per second."""
import json
import time

import torch


def main():
    n = 4 << 30
    a = torch.empty(n, dtype=torch.uint8, device="cuda")
    b = torch.empty(n, dtype=torch.uint8, device="cuda")
    a.fill_(1)
    for _ in range(3):
        b.copy_(a)
    torch.cuda.synchronize()
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        b.copy_(a)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    s = torch.empty(n // 4, dtype=torch.float32, device="cuda").fill_(1.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        s.sum()
    torch.cuda.synchronize()
    dr = (time.perf_counter() - t0) / reps
    print(json.dumps({"copy_GBps_read_plus_write": round(2 * n / dt / 1e9, 1), "read_only_sum_GBps": round(n / dr / 1e9, 1), "buffer_GiB": n >> 30}))


if __name__ == "__main__":
    main()
