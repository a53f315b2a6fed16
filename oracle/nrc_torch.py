"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Imported by tests/ alone; never by the product (gfxexp_amd/), bench.py or smoke().

A second, independently written model of the neural radiance cache network (the configuration of neural_radiance_caching/
network_interface.cu:48-132, SURVEY 8c: "validate against its own fp32 PyTorch-ROCm model"): plain fp32 PyTorch, every
gradient -- MLP weights AND hash-grid entries -- by autograd.  PARITY UNPINNED like oracle/nrc_net.py (tiny-cuda-nn is an empty
submodule of the reference checkout); what this file adds is a check of the two hand-derived backward passes (the numpy
restatement's and the HIP kernel's) by something that derives nothing by hand.

Written from the published descriptions, not from nrc_net.py:
  * multiresolution hash encoding (Mueller et al. 2022, section 3): L = 16 levels, F = 2 features, T = 2^15 entries, N_min = 16,
    growth factor b = 2.  tiny-cuda-nn's grid convention (its GridEncoding: scale_l = N_min * b^l - 1, resolution
    ceil(scale_l) + 1, the position scaled and shifted by half a cell: x * scale_l + 0.5) is what the reference would run, so it
    is the one used; a level whose dense grid fits its table is indexed densely (x + y R + z R^2), the others through the spatial hash
    (x * 1) xor (y * 2654435761) xor (z * 805459861) mod T in 32-bit arithmetic; d-linear interpolation of the 2^3 corners.
  * one-blob encoding (Mueller et al. 2019 / 2021): k = 4 bins per dimension, the activation of a bin is the integral over the bin of a
    quartic kernel 15/16 (1 - u^2)^2 of radius one bin centred at the input, the input wrapped with period 1.
  * triangle-wave frequency encoding (12 octaves): |frac(x 2^(f-1)) - 1/2| * 4 - 1.
  * network: inputs padded with ones to 64, n hidden layers of 64 ReLU units, linear output layer, no biases.
  * loss: relative L2 with the squared LUMINANCE of the prediction (+ 0.01) as the normaliser, which is treated as a constant
    (Mueller et al. 2021, section 5: "relative L2 loss ... normalised by the luminance").
The parameter vector has the product's layout (gfx_nrc_set_params): W0 [64][64], W1.. [64][64], Wout [16][64] (row = output unit,
column = input feature: position encoding | one-blob 5 x 4 | identity 6 | ones), then the grid [entries][2].
"""
import math

import torch

WIDTH, OUT_ROWS, N_OUT = 64, 16, 3
LEVELS, FEATURES, TABLE = 16, 2, 1 << 15
N_MIN, GROWTH = 16, 2.0
BINS, OCTAVES = 4, 12
PRIME_Y, PRIME_Z = 2654435761, 805459861
HASHGRID, TRIANGLEWAVE = 1, 0


def level_table():
    """(scale, resolution, entries, first entry) of every level; tables are padded to a multiple of 8 entries."""
    rows, first = [], 0
    for level in range(LEVELS):
        scale = float(torch.tensor(N_MIN * GROWTH ** level - 1.0, dtype=torch.float32))
        res = int(math.ceil(scale)) + 1
        entries = min(-(-res ** 3 // 8) * 8, TABLE)
        rows.append((scale, res, entries, first))
        first += entries
    return rows, first


def num_params(pos_enc, hidden_layers):
    n = WIDTH * WIDTH * hidden_layers + OUT_ROWS * WIDTH
    return n + (level_table()[1] * FEATURES if pos_enc == HASHGRID else 0)


class Model:
    def __init__(self, flat_params, pos_enc=HASHGRID, hidden_layers=2, device="cpu"):
        p = torch.as_tensor(flat_params, dtype=torch.float32, device=device).clone()
        assert p.numel() == num_params(pos_enc, hidden_layers)
        self.pos_enc, self.hidden_layers, self.device = pos_enc, hidden_layers, device
        self.flat = p.requires_grad_(True)
        at = 0
        self.weights = []
        for k in range(hidden_layers):
            self.weights.append(self.flat[at:at + WIDTH * WIDTH].view(WIDTH, WIDTH)); at += WIDTH * WIDTH
        self.w_out = self.flat[at:at + OUT_ROWS * WIDTH].view(OUT_ROWS, WIDTH); at += OUT_ROWS * WIDTH
        self.mlp_params = at
        self.grid = self.flat[at:].view(-1, FEATURES) if pos_enc == HASHGRID else None

    # ---- encodings
    def _hash_grid(self, xyz):
        feats = []
        i64 = torch.int64
        for scale, res, entries, first in level_table()[0]:
            pos = xyz * scale + 0.5
            cell = torch.floor(pos)
            t = pos - cell                                       # interpolation weights in [0, 1)
            cell = cell.to(i64)
            acc = torch.zeros(xyz.shape[0], FEATURES, dtype=torch.float32, device=xyz.device)
            for corner in range(8):
                bits = [(corner >> axis) & 1 for axis in range(3)]
                cx, cy, cz = [(cell[:, a] + bits[a]) & 0xFFFFFFFF for a in range(3)]
                if res ** 3 <= entries:
                    index = (cx + cy * res + cz * res * res) % entries
                else:
                    index = ((cx ^ (cy * PRIME_Y) ^ (cz * PRIME_Z)) & 0xFFFFFFFF) % entries
                w = torch.ones(xyz.shape[0], dtype=torch.float32, device=xyz.device)
                for a in range(3):
                    w = w * (t[:, a] if bits[a] else 1.0 - t[:, a])
                acc = acc + w[:, None] * self.grid[first + index]
            feats.append(acc)
        return torch.cat(feats, dim=1)                           # [N, 32]: level-major, feature-minor

    @staticmethod
    def _triangle_wave(xyz):
        cols = []
        for axis in range(3):
            for f in range(OCTAVES):
                v = xyz[:, axis] * (2.0 ** (f - 1))
                cols.append(torch.abs(v - torch.floor(v) - 0.5) * 4.0 - 1.0)
        return torch.stack(cols, dim=1)

    @staticmethod
    def _one_blob(v5):
        def kernel_integral(u):                                  # integral of 15/16 (1 - s^2)^2 from -1 to clamp(u, -1, 1)
            u = torch.clamp(u, -1.0, 1.0)
            return 0.5 + (15.0 / 16.0) * (u - (2.0 / 3.0) * u ** 3 + 0.2 * u ** 5)
        cols = []
        for dim in range(5):
            x = v5[:, dim]
            for b in range(BINS):
                lo, hi = b / BINS, (b + 1) / BINS
                total = 0.0
                for shift in (-1.0, 0.0, 1.0):                    # the input wrapped with period 1
                    centre = x + shift
                    total = total + kernel_integral((hi - centre) * BINS) - kernel_integral((lo - centre) * BINS)
                cols.append(total)
        return torch.stack(cols, dim=1)

    def encode(self, x):
        position = self._hash_grid(x[:, 0:3]) if self.pos_enc == HASHGRID else self._triangle_wave(x[:, 0:3])
        feats = torch.cat([position, self._one_blob(x[:, 3:8]), x[:, 8:14]], dim=1)
        ones = torch.ones(x.shape[0], WIDTH - feats.shape[1], dtype=torch.float32, device=x.device)
        return torch.cat([feats, ones], dim=1)

    def forward(self, x):
        x = torch.as_tensor(x, dtype=torch.float32, device=self.device)
        h = self.encode(x)
        for w in self.weights:
            h = torch.relu(h @ w.t())
        return (h @ self.w_out.t())[:, :N_OUT]

    def normaliser(self, x):
        """luminance(prediction)^2 + 0.01 per record: a constant of the loss (no gradient flows through it)."""
        pred = self.forward(x).detach()
        return (0.299 * pred[:, 0] + 0.587 * pred[:, 1] + 0.114 * pred[:, 2]) ** 2 + 0.01

    def loss(self, x, target, normaliser=None):
        pred = self.forward(x)
        target = torch.as_tensor(target, dtype=torch.float32, device=self.device)
        norm = self.normaliser(x) if normaliser is None else normaliser
        return (((pred - target) ** 2) / norm[:, None]).sum() / (pred.shape[0] * N_OUT)

    def loss_and_gradient(self, x, target):
        """(loss, dLoss/dParams as one flat fp32 vector in the product's layout) by autograd."""
        if self.flat.grad is not None:
            self.flat.grad = None
        value = self.loss(x, target)
        value.backward()
        return float(value.detach()), self.flat.grad.detach().clone()


class Trainer:
    """The training loop of the cache in plain fp32 with autograd gradients: Adam (beta1 0.9, beta2 0.99, L2 regularisation 1e-6 folded into the
    gradient of the MLP weights, hash-grid entries with an exactly zero gradient left alone, epsilon 1e-15 for the hash-grid configuration
    and 1e-8 otherwise -- network_interface.cu:53-64, 91, 118) and the debiased EMA (0.99) of the weights.  No bf16, no fp16 gradient sums,
    no loss scaling: what the kernels' precision contract is measured against over a whole training run (tests/test_gpu_nrc_net.py
    test_loss_curve_against_fp32_training, tools/nrc_loss_curves.py)."""

    def __init__(self, flat_params, pos_enc=HASHGRID, hidden_layers=2, learning_rate=1e-2, device="cpu"):
        self.model = Model(flat_params, pos_enc, hidden_layers, device)
        self.lr, self.beta1, self.beta2, self.l2, self.decay = learning_rate, 0.9, 0.99, 1e-6, 0.99
        self.eps = 1e-15 if pos_enc == HASHGRID else 1e-8
        p = self.model.flat
        self.m, self.v = torch.zeros_like(p), torch.zeros_like(p)
        self.ema = p.detach().clone()
        self.step = 0

    def train(self, x, target):
        """One step; returns the loss the step started from."""
        mdl = self.model
        value, grad = mdl.loss_and_gradient(x, target)
        self.step += 1
        t = self.step
        with torch.no_grad():
            p = mdl.flat
            is_grid = torch.zeros_like(p, dtype=torch.bool)
            is_grid[mdl.mlp_params:] = True
            active = ~(is_grid & (grad == 0))
            grad = torch.where(is_grid, grad, grad + self.l2 * p)
            self.m = torch.where(active, self.beta1 * self.m + (1 - self.beta1) * grad, self.m)
            self.v = torch.where(active, self.beta2 * self.v + (1 - self.beta2) * grad * grad, self.v)
            lr_t = self.lr * math.sqrt(1.0 - self.beta2 ** t) / (1.0 - self.beta1 ** t)
            p -= torch.where(active, lr_t * self.m / (torch.sqrt(self.v) + self.eps), torch.zeros_like(p))
            self.ema = ((1 - self.decay) * p + self.decay * (1.0 - self.decay ** (t - 1)) * self.ema) / (1.0 - self.decay ** t)
        return value

    def infer(self, x):
        """Predictions with the EMA weights (what gfx_nrc_infer uses)."""
        with torch.no_grad():
            return Model(self.ema, self.model.pos_enc, self.model.hidden_layers, self.model.device).forward(x).cpu().numpy()
