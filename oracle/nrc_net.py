"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; never by the product (gfxexp_amd/).

numpy restatement of the neural radiance cache network that neural_radiance_caching/
network_interface.cu:48-157 configures in tiny-cuda-nn:

    NetworkWithInputEncoding(14 -> 3)
      encoding  Composite{ HashGrid(3 dims: 16 levels x 2 features, 2^15 entries, base 16, scale 2)
                           | TriangleWave(3 dims, 12 frequencies),
                           OneBlob(5 dims, 4 bins), Identity(6 dims) }  -> padded with ones to 64
      network   FullyFusedMLP(64 neurons, n hidden layers, ReLU, no output activation, no biases)
      loss      RelativeL2Luminance           optimizer  EMA(0.99) o Adam(lr, 0.9, 0.99, eps, l2_reg 1e-6)

PARITY UNPINNED: tiny-cuda-nn (NVlabs/tiny-cuda-nn, git submodule ext/tiny-cuda-nn) is not vendored
in the reference tree and its pinned commit is not recoverable (.gitmodules carries no SHA), and no
test in the reference exercises the network.  The arithmetic below restates the published
algorithms (Mueller et al. 2022 "Instant neural graphics primitives": grid scale b^l * N_min - 1,
resolution ceil(scale) + 1, x * scale + 0.5, trilinear weights, spatial hash
x ^ y * 2654435761 ^ z * 805459861, dense indexing while the level fits; Mueller et al. 2021
"Real-time neural radiance caching": one-blob encoding with a quartic kernel wrapped with period 1,
relative L2 loss normalised by the squared prediction luminance + 0.01) and is anchored on the
reference's call sites (network_interface.cu:141-157: column-major fp32 [14, N] in / [3, N] out).
Parameter initialisation is this build's own (PCG32 stream, see init_params); the product takes the
same arrays through gfx_nrc_set_params so both sides start identically.

Numerical contract shared with the HIP kernels: parameters, encoded inputs and hidden activations
are rounded to bf16 (round-to-nearest-even) where the product feeds the MFMA units; products are
accumulated in fp32.  Tolerances are stated in tests/test_gpu_nrc_net.py.
"""
import numpy as np

N_IN, N_OUT, WIDTH, OUT_PAD = 14, 3, 64, 16
HASH_LEVELS, HASH_FEATURES, LOG2_HASHMAP, BASE_RES, LEVEL_SCALE = 16, 2, 15, 16, 2.0
TRI_FREQS = 12
ONEBLOB_BINS = 4
POS_HASHGRID, POS_TRIANGLEWAVE = 1, 0     # PositionEncoding enum order of network_interface.h:5-8


def bf16_round(x):
    """fp32 -> bf16 (round to nearest even) -> fp32."""
    x = np.ascontiguousarray(x, np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    rounded = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
    out = rounded.view(np.float32).copy()
    nan = np.isnan(x)
    out[nan] = x[nan]
    return out.reshape(x.shape)


def grid_levels():
    """[(scale, resolution, numEntries, offset)] per level; entries padded to a multiple of 8."""
    out, offset = [], 0
    for l in range(HASH_LEVELS):
        scale = np.float32(np.exp2(np.float32(l) * np.float32(np.log2(LEVEL_SCALE))) * BASE_RES - 1.0)
        res = int(np.ceil(scale)) + 1
        n = min(((res ** 3 + 7) // 8) * 8, 1 << LOG2_HASHMAP)
        out.append((scale, res, n, offset))
        offset += n
    return out, offset


def layout(pos_enc, num_hidden_layers):
    """Offsets (in fp32 elements) of the parameter blob: W0 [64][64], hidden [64][64] x (n-1),
    Wout [16][64], then the hash grid (entries x 2).  W[out][in], `in` in canonical feature order."""
    mats = [("W0", WIDTH, WIDTH)] + [(f"W{i}", WIDTH, WIDTH) for i in range(1, num_hidden_layers)] + [("Wout", OUT_PAD, WIDTH)]
    off, table = 0, {}
    for name, rows, cols in mats:
        table[name] = (off, rows, cols)
        off += rows * cols
    grid_off = off
    if pos_enc == POS_HASHGRID:
        _, total = grid_levels()
        off += total * HASH_FEATURES
    return table, grid_off, off


class Pcg32:
    def __init__(self, state):
        self.state = np.uint64(state)

    def uniform(self, n):
        out = np.empty(n, np.float32)
        s = int(self.state)
        for i in range(n):
            old = s
            s = (old * 6364136223846793005 + 1) & 0xFFFFFFFFFFFFFFFF
            xorshifted = (((old >> 18) ^ old) >> 27) & 0xFFFFFFFF
            rot = old >> 59
            v = ((xorshifted >> rot) | (xorshifted << ((-rot) & 31))) & 0xFFFFFFFF
            out[i] = np.uint32((v >> 9) | 0x3F800000).view(np.float32) - np.float32(1.0)
        self.state = np.uint64(s)
        return out


def init_params(pos_enc, num_hidden_layers, seed=1337):
    """Xavier-uniform MLP weights, U(-1e-4, 1e-4) grid features (the distributions tiny-cuda-nn uses;
    the stream itself -- PCG32 state `seed`, increment 1 -- is this build's choice)."""
    table, grid_off, total = layout(pos_enc, num_hidden_layers)
    rng = Pcg32(seed)
    p = np.zeros(total, np.float32)
    for name, (off, rows, cols) in table.items():
        fan_in = 64 if name != "W0" else 64
        bound = np.float32(np.sqrt(6.0 / (fan_in + rows)))
        p[off:off + rows * cols] = (rng.uniform(rows * cols) * np.float32(2) - np.float32(1)) * bound
    if total > grid_off:
        n = total - grid_off
        p[grid_off:] = (rng.uniform(n) * np.float32(2) - np.float32(1)) * np.float32(1e-4)
    return p


def _quartic_cdf(x, inv_radius):
    u = (x * np.float32(inv_radius)).astype(np.float32)
    u2 = u * u
    u4 = u2 * u2
    v = np.float32(15.0 / 16.0) * u * (np.float32(1) - np.float32(2.0 / 3.0) * u2 + np.float32(1.0 / 5.0) * u4) + np.float32(0.5)
    return np.clip(v, np.float32(0), np.float32(1)).astype(np.float32)


def encode_oneblob(x5):
    """x5: [N, 5] -> [N, 20]; feature 4*d + bin.  Kernel radius = 1 bin, wrapped with period 1."""
    n = x5.shape[0]
    out = np.zeros((n, 5 * ONEBLOB_BINS), np.float32)
    for d in range(5):
        x = x5[:, d].astype(np.float32)
        cdf = []
        for b in range(ONEBLOB_BINS + 1):
            left = np.float32(b / ONEBLOB_BINS)
            c = _quartic_cdf(left - x, ONEBLOB_BINS) + _quartic_cdf(left - x - np.float32(1), ONEBLOB_BINS) + \
                _quartic_cdf(left - x + np.float32(1), ONEBLOB_BINS)
            cdf.append(c.astype(np.float32))
        for b in range(ONEBLOB_BINS):
            out[:, ONEBLOB_BINS * d + b] = cdf[b + 1] - cdf[b]
    return out


def encode_trianglewave(x3):
    """[N, 3] -> [N, 36]; feature 12*d + f: |frac(x * 2^(f-1)) - 0.5| * 4 - 1."""
    n = x3.shape[0]
    out = np.zeros((n, 3 * TRI_FREQS), np.float32)
    for d in range(3):
        for f in range(TRI_FREQS):
            x = np.ldexp(x3[:, d].astype(np.float32), f - 1).astype(np.float32)
            out[:, TRI_FREQS * d + f] = np.abs(x - np.floor(x) - np.float32(0.5)) * np.float32(4) - np.float32(1)
    return out


def _grid_index(ix, iy, iz, res, n_entries):
    ix, iy, iz = ix.astype(np.uint64), iy.astype(np.uint64), iz.astype(np.uint64)
    if res ** 3 <= n_entries:          # dense level
        return ((ix + iy * res + iz * res * res) % n_entries).astype(np.int64)
    h = (ix * np.uint64(1)) ^ (iy * np.uint64(2654435761)) ^ (iz * np.uint64(805459861))
    return ((h & np.uint64(0xFFFFFFFF)) % np.uint64(n_entries)).astype(np.int64)


def hash_corners(x3):
    """Per level: (corner indices [N, 8] into the level's table, trilinear weights [N, 8])."""
    levels, _ = grid_levels()
    out = []
    for scale, res, n_entries, offset in levels:
        pos = x3.astype(np.float32) * np.float32(scale) + np.float32(0.5)
        base = np.floor(pos).astype(np.float32)
        frac = (pos - base).astype(np.float32)
        b = base.astype(np.int64)
        idx = np.zeros((x3.shape[0], 8), np.int64)
        w = np.ones((x3.shape[0], 8), np.float32)
        for c in range(8):
            o = [(c >> k) & 1 for k in range(3)]
            cx, cy, cz = (b[:, 0] + o[0]) & 0xFFFFFFFF, (b[:, 1] + o[1]) & 0xFFFFFFFF, (b[:, 2] + o[2]) & 0xFFFFFFFF
            idx[:, c] = _grid_index(cx, cy, cz, res, n_entries) + offset
            wc = np.ones(x3.shape[0], np.float32)
            for k in range(3):
                wc = wc * (frac[:, k] if o[k] else (np.float32(1) - frac[:, k]))
            w[:, c] = wc
        out.append((idx, w))
    return out


def encode_hashgrid(x3, grid_bf16):
    """[N, 3] -> [N, 32]; feature 2*level + f.  grid: [entries, 2] (already bf16-rounded)."""
    n = x3.shape[0]
    out = np.zeros((n, HASH_LEVELS * HASH_FEATURES), np.float32)
    for l, (idx, w) in enumerate(hash_corners(x3)):
        acc = np.zeros((n, HASH_FEATURES), np.float32)
        for c in range(8):
            acc = acc + w[:, c:c + 1] * grid_bf16[idx[:, c]]
        out[:, 2 * l:2 * l + 2] = acc
    return out


class NrcNet:
    """fp32 master parameters + Adam/EMA state; forward/backward with the bf16 rounding contract."""

    def __init__(self, pos_enc=POS_HASHGRID, num_hidden_layers=2, learning_rate=1e-2, params=None, bf16=True, grid_grad_f16=True):
        # grid_grad_f16: the hash-grid gradient is summed as fp16 pairs (tiny-cuda-nn scatters __half2; nrc.hip k_nrc_grid_scatter
        # sums a level table per chunk of records in LDS with ds_pk_add_f16): contributions are rounded to fp16, and so is the
        # per-entry sum of a chunk.  The order of the additions inside a chunk is not defined on the device, so the restatement
        # sums a chunk in fp32 and rounds once; the chunks are added in fp32 (gradients()).
        self.grid_grad_f16 = grid_grad_f16 and bf16
        self.pos_enc, self.n_hidden, self.lr = pos_enc, num_hidden_layers, np.float32(learning_rate)
        self.rnd = bf16_round if bf16 else (lambda a: np.ascontiguousarray(a, np.float32))   # bf16=False: plain fp32 (gradient checks)
        self.table, self.grid_off, self.total = layout(pos_enc, num_hidden_layers)
        self.params = init_params(pos_enc, num_hidden_layers) if params is None else np.array(params, np.float32)
        assert self.params.size == self.total
        self.m = np.zeros(self.total, np.float32)
        self.v = np.zeros(self.total, np.float32)
        self.ema = self.params.copy()          # inference parameters (EMA of the trained weights)
        self.step = 0
        self.eps = np.float32(1e-15 if pos_enc == POS_HASHGRID else 1e-8)
        self.beta1, self.beta2, self.l2_reg, self.ema_decay = np.float32(0.9), np.float32(0.99), np.float32(1e-6), np.float32(0.99)

    def _mats(self, p):
        names = ["W0"] + [f"W{i}" for i in range(1, self.n_hidden)] + ["Wout"]
        return [self.rnd(p[self.table[k][0]:self.table[k][0] + self.table[k][1] * self.table[k][2]]).reshape(self.table[k][1], self.table[k][2])
                for k in names]

    def encode(self, x, p):
        """x: [N, 14] -> canonical encoded features [N, 64] (bf16-rounded)."""
        x = np.ascontiguousarray(x, np.float32)
        if self.pos_enc == POS_HASHGRID:
            grid = self.rnd(p[self.grid_off:]).reshape(-1, HASH_FEATURES)
            pos = encode_hashgrid(x[:, 0:3], grid)
        else:
            pos = encode_trianglewave(x[:, 0:3])
        ob = encode_oneblob(x[:, 3:8])
        ident = x[:, 8:14]
        feat = np.concatenate([pos, ob, ident], axis=1)
        pad = np.ones((x.shape[0], WIDTH - feat.shape[1]), np.float32)
        return self.rnd(np.concatenate([feat, pad], axis=1))

    def forward(self, x, p, keep=False):
        mats = self._mats(p)
        h = self.encode(x, p)
        acts = [h]
        for W in mats[:-1]:
            h = self.rnd(np.maximum(h @ W.T, np.float32(0)))
            acts.append(h)
        y = (h @ mats[-1].T).astype(np.float32)
        return (y[:, :N_OUT], acts, mats) if keep else y[:, :N_OUT]

    def infer(self, x):
        return self.forward(x, self.ema)

    @staticmethod
    def loss_and_grad(pred, target, loss_scale=1.0):
        lum = np.float32(0.299) * pred[:, 0] + np.float32(0.587) * pred[:, 1] + np.float32(0.114) * pred[:, 2]
        denom = (lum * lum + np.float32(0.01)).astype(np.float32)[:, None]
        diff = (pred - target).astype(np.float32)
        n_total = np.float32(pred.shape[0] * N_OUT)
        loss = (diff * diff / denom) / n_total
        grad = np.float32(loss_scale) * np.float32(2) * diff / denom / n_total
        return loss.astype(np.float32), grad.astype(np.float32)

    def gradients(self, x, target, loss_scale=128.0):
        """(mean loss, dL/dparams * loss_scale) for the TRAINING parameters."""
        p = self.params
        pred, acts, mats = self.forward(x, p, keep=True)
        loss, dy = self.loss_and_grad(pred, np.ascontiguousarray(target, np.float32), loss_scale)
        g = np.zeros(self.total, np.float32)
        names = ["W0"] + [f"W{i}" for i in range(1, self.n_hidden)] + ["Wout"]
        delta = np.zeros((x.shape[0], OUT_PAD), np.float32)
        delta[:, :N_OUT] = dy
        delta = self.rnd(delta)
        for li in range(len(mats) - 1, -1, -1):
            a = acts[li]
            off, rows, cols = self.table[names[li]]
            g[off:off + rows * cols] = (delta.T @ a).reshape(-1)
            if li > 0 or self.pos_enc == POS_HASHGRID:
                back = (delta @ mats[li]).astype(np.float32)
                if li > 0:
                    back = back * (a > 0)
                delta = self.rnd(back) if li > 0 else back    # the grid scatter consumes fp32
        if self.pos_enc == POS_HASHGRID:
            n_rec = x.shape[0]
            entries = (self.total - self.grid_off) // HASH_FEATURES
            # nrc.hip k_nrc_grid_scatter: the records are cut into chunks (sixteen for a 16 384-record step, at least 256 records each);
            # a chunk's contributions are summed in fp16 (LDS atomics, order undefined -- restated as an fp32 sum rounded once), the
            # chunks' sums are added in fp32 in chunk order
            chunk = max(256, ((n_rec + 15) // 16 + 63) // 64 * 64) if self.grid_grad_f16 else n_rec
            corners = hash_corners(np.ascontiguousarray(x[:, 0:3], np.float32))
            total = np.zeros((entries, HASH_FEATURES), np.float32)
            for begin in range(0, n_rec, chunk):
                sl = slice(begin, min(begin + chunk, n_rec))
                gg = np.zeros((entries, HASH_FEATURES), np.float32)
                for l, (idx, w) in enumerate(corners):
                    d = delta[sl, 2 * l:2 * l + 2]
                    for c in range(8):
                        contrib = (w[sl, c:c + 1] * d).astype(np.float32)
                        if self.grid_grad_f16:
                            contrib = np.clip(contrib, -65504.0, 65504.0).astype(np.float16).astype(np.float32)
                        np.add.at(gg, idx[sl, c], contrib)
                if self.grid_grad_f16:
                    gg = gg.astype(np.float16).astype(np.float32)
                total = (total + gg).astype(np.float32)
            g[self.grid_off:] = total.reshape(-1)
        return np.float32(loss.sum()), g

    def optimizer_step(self, g, loss_scale=128.0):
        """Adam (with L2 regularisation folded into the gradient for the MLP weights; hash-grid entries
        with an exactly zero gradient are skipped) followed by the debiased EMA of the weights."""
        self.step += 1
        t = self.step
        grad = (g / np.float32(loss_scale)).astype(np.float32)
        is_grid = np.zeros(self.total, bool)
        is_grid[self.grid_off:] = True
        active = ~(is_grid & (grad == 0))
        grad = np.where(is_grid, grad, grad + self.l2_reg * self.params).astype(np.float32)
        m = np.where(active, self.beta1 * self.m + (np.float32(1) - self.beta1) * grad, self.m).astype(np.float32)
        v = np.where(active, self.beta2 * self.v + (np.float32(1) - self.beta2) * grad * grad, self.v).astype(np.float32)
        lr_t = np.float32(float(self.lr) * np.sqrt(1.0 - float(self.beta2) ** t) / (1.0 - float(self.beta1) ** t))
        upd = (lr_t * m / (np.sqrt(v) + self.eps)).astype(np.float32)
        self.params = np.where(active, self.params - upd, self.params).astype(np.float32)
        self.m, self.v = m, v
        d = np.float32(self.ema_decay)
        debias_old = np.float32(1.0 - float(d) ** (t - 1))
        debias_new = np.float32(1.0 / (1.0 - float(d) ** t))
        self.ema = (((np.float32(1) - d) * self.params + d * debias_old * self.ema) * debias_new).astype(np.float32)

    def train(self, x, target, loss_scale=128.0):
        loss, g = self.gradients(x, target, loss_scale)
        self.optimizer_step(g, loss_scale)
        return loss
