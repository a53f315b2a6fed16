#!/usr/bin/env python
"""Training-quality check of the NRC kernels' precision contract (bf16 weights / activations / deltas, fp16 hash-grid gradient sums) over a
whole training run: the same records, step by step, through gfx_nrc_train and through the fp32 autograd trainer of oracle/nrc_torch.py
(test infrastructure) from the same initial parameters; the two loss curves, and the error of both EMA networks on held-out queries.

    python tools/nrc_loss_curves.py [--steps 400] [--batch 4096] [--hidden 2] > profiles/r05_nrc_loss_curves.json
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402


def target_field(x):
    """A radiance-like target with detail at several scales (the hash grid's upper levels matter) and a view-dependent term."""
    p = x[:, :3]
    base = 0.5 + 0.5 * np.sin(6.0 * p[:, 0]) * np.cos(5.0 * p[:, 1])
    fine = 0.25 * np.sin(40.0 * p[:, 0] + 17.0 * p[:, 2]) * np.sin(33.0 * p[:, 1])
    edge = (np.floor(8.0 * p[:, 0]) + np.floor(8.0 * p[:, 2])) % 2 * 0.3
    view = 0.2 * np.cos(3.0 * x[:, 3]) * x[:, 8]
    r = np.clip(base + fine + view, 0.0, None)
    g = np.clip(0.6 * base + edge + 0.1 * x[:, 9], 0.0, None)
    b = np.clip(0.3 + 0.4 * fine + 0.3 * edge, 0.0, None)
    return np.stack([r, g, b], 1).astype(np.float32)


def inputs(rng, n):
    x = rng.random((n, 14)).astype(np.float32)
    x[:, 3:8] = x[:, 3:8] * 6 - 3
    return x


def run(steps=400, batch=4096, hidden=2, seed=29, device="cuda"):
    import torch
    from gfxexp_amd import api
    from oracle import nrc_net as N
    from oracle import nrc_torch as T
    rng = np.random.default_rng(seed)
    ctx = api.Context(0)
    net = api.NeuralRadianceCache(ctx, N.POS_HASHGRID, hidden, 1e-2)
    ref = T.Trainer(net.get_params(0), T.HASHGRID, hidden, 1e-2, device=device)
    kernel, fp32 = [], []
    for _ in range(steps):
        x = inputs(rng, batch)
        t = target_field(x)
        dx, dt = torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda()
        kernel.append(float(net.train(dx.data_ptr(), dt.data_ptr(), batch, want_loss=True)))
        fp32.append(float(ref.train(torch.from_numpy(x).to(device), torch.from_numpy(t).to(device))))
    xe = inputs(rng, 16384)
    te = target_field(xe)
    dx = torch.from_numpy(xe).cuda()
    dy = torch.zeros((16384, 3), dtype=torch.float32, device="cuda")
    net.infer(dx.data_ptr(), 16384, dy.data_ptr())
    torch.cuda.synchronize()
    yk = dy.cpu().numpy()
    yr = ref.infer(torch.from_numpy(xe).to(device))
    out = {"steps": steps, "batch": batch, "hidden_layers": hidden, "loss_kernel": kernel, "loss_fp32": fp32,
           "heldout_mse_kernel": float(((yk - te) ** 2).mean()), "heldout_mse_fp32": float(((yr - te) ** 2).mean()),
           "heldout_kernel_vs_fp32_rms": float(np.sqrt(((yk - yr) ** 2).mean())), "target_rms": float(np.sqrt((te ** 2).mean()))}
    net.close()
    ctx.close()
    return out


def smoothed(curve, at, window=20):
    lo = max(0, at - window)
    return float(np.mean(curve[lo:at]))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--hidden", type=int, default=2)
    a = ap.parse_args()
    r = run(a.steps, a.batch, a.hidden)
    marks = [m for m in (20, 50, 100, 200, 400, 800) if m <= a.steps]
    r["smoothed_over_20_steps"] = {str(m): {"kernel": smoothed(r["loss_kernel"], m), "fp32": smoothed(r["loss_fp32"], m)} for m in marks}
    print(json.dumps(r))
