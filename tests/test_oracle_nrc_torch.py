"""The two restatements of the NRC network against each other on the CPU: oracle/nrc_net.py (numpy, hand-derived backward pass) in its
plain-fp32 mode and oracle/nrc_torch.py (PyTorch fp32, every gradient by autograd, written from the published descriptions).  What
agrees here: the forward definitions (hash-grid indexing and interpolation, one-blob, triangle wave, padding, layer order, the
parameter layout) and -- the point of the second model -- the backward pass through the MLP AND into the hash-grid entries.
The GPU kernels are held against the autograd model in tests/test_gpu_nrc_net.py."""
import numpy as np
import pytest

from oracle import nrc_net as N
from oracle import nrc_torch as T


def _inputs(rng, n):
    x = rng.random((n, 14)).astype(np.float32)
    x[:, 3:8] = x[:, 3:8] * 6 - 3                       # one-blob inputs outside [0, 1): the wrap-around
    x[:4, :3] = [[0, 0, 0], [1, 1, 1], [0.999999, 0.5, 0.25], [0.5, 0.0, 1.0]]
    return x


def _targets(x):
    return np.stack([np.sin(6 * x[:, 0]) * 0.5 + 0.5, x[:, 1] * x[:, 8], 0.3 + 0.2 * np.cos(9 * x[:, 2])], 1).astype(np.float32)


def _params(rng, pos_enc, hidden):
    p = N.init_params(pos_enc, hidden)
    _, grid_off, _ = N.layout(pos_enc, hidden)
    p[grid_off:] = rng.random(p.size - grid_off).astype(np.float32) * 2 - 1
    return p


def test_the_two_layouts_agree():
    for pos_enc in (N.POS_HASHGRID, N.POS_TRIANGLEWAVE):
        for hidden in (2, 5):
            assert T.num_params(pos_enc, hidden) == N.layout(pos_enc, hidden)[2]
    rows, total = T.level_table()
    levels, total_n = N.grid_levels()
    assert total == total_n and [(r[1], r[2], r[3]) for r in rows] == [(l[1], l[2], l[3]) for l in levels]
    assert all(abs(r[0] - float(l[0])) == 0.0 for r, l in zip(rows, levels))
    # levels 0-1 are dense (17^3, 32^3 <= 2^15), the others hashed: both index paths are exercised
    assert rows[0][1] ** 3 <= rows[0][2] and rows[2][1] ** 3 > rows[2][2]


@pytest.mark.parametrize("pos_enc,hidden", [(N.POS_HASHGRID, 2), (N.POS_HASHGRID, 5), (N.POS_TRIANGLEWAVE, 2)])
def test_forward_passes_agree_in_fp32(pos_enc, hidden):
    rng = np.random.default_rng(21)
    p = _params(rng, pos_enc, hidden)
    x = _inputs(rng, 1500)
    ref = N.NrcNet(pos_enc, hidden, params=p, bf16=False)
    model = T.Model(p, pos_enc, hidden)
    enc_n = ref.encode(x, p)
    enc_t = model.encode(__import__("torch").as_tensor(x)).detach().numpy()
    assert np.abs(enc_n - enc_t).max() <= 2e-6, np.abs(enc_n - enc_t).max()
    y_n = ref.forward(x, p)
    y_t = model.forward(x).detach().numpy()
    assert np.abs(y_n - y_t).max() <= 2e-5 * max(1.0, np.abs(y_n).max()), np.abs(y_n - y_t).max()


@pytest.mark.parametrize("pos_enc,hidden", [(N.POS_HASHGRID, 2), (N.POS_TRIANGLEWAVE, 5)])
def test_hand_derived_gradients_agree_with_autograd(pos_enc, hidden):
    """Loss and dLoss/dParams of one batch: numpy's hand-written backward pass (fp32 mode, fp32 grid scatter) against autograd --
    MLP weights and every touched hash-grid entry."""
    rng = np.random.default_rng(22)
    p = _params(rng, pos_enc, hidden)
    x = _inputs(rng, 1024)
    t = _targets(x)
    ref = N.NrcNet(pos_enc, hidden, params=p, bf16=False, grid_grad_f16=False)
    loss_n, g_n = ref.gradients(x, t, loss_scale=1.0)
    loss_t, g_t = T.Model(p, pos_enc, hidden).loss_and_gradient(x, t)
    g_t = g_t.numpy()
    assert abs(loss_n - loss_t) <= 1e-5 * abs(loss_t), (loss_n, loss_t)
    grid_off = ref.grid_off
    for name, sl in (("mlp", slice(0, grid_off)), ("grid", slice(grid_off, None))):
        if g_t[sl].size == 0:
            continue
        rel = np.linalg.norm(g_n[sl] - g_t[sl]) / np.linalg.norm(g_t[sl])
        assert rel <= 2e-5, (name, rel)
        assert np.array_equal(g_n[sl] != 0, g_t[sl] != 0) or np.mean((g_n[sl] != 0) == (g_t[sl] != 0)) > 0.9999, name
    # the unused output rows (3..15 of Wout) receive no gradient on either side
    off, rows, cols = ref.table["Wout"]
    assert not g_t[off + 3 * cols: off + rows * cols].any() and not g_n[off + 3 * cols: off + rows * cols].any()


def test_what_the_bf16_contract_costs_in_the_gradient():
    """The restatement in the kernels' precision contract (bf16 operands, bf16-rounded deltas, fp16 grid scatter) against fp32
    autograd on the batch of tests/test_gpu_nrc_net.py::test_kernels_against_the_autograd_model: 0.3 % over the MLP weights, 5.9 %
    over the hash-grid entries (the end of the backward chain) -- the figures that test's tolerances are set by."""
    rng = np.random.default_rng(31)
    p = _params(rng, N.POS_HASHGRID, 2)
    x = rng.random((128 * 24, 14)).astype(np.float32)
    x[:, 3:8] = x[:, 3:8] * 6 - 3
    t = _targets(x)
    _, g_t = T.Model(p, N.POS_HASHGRID, 2).loss_and_gradient(x, t)
    g_t = g_t.numpy()
    _, g_n = N.NrcNet(N.POS_HASHGRID, 2, params=p, bf16=True, grid_grad_f16=True).gradients(x, t, loss_scale=128.0)
    g_n = g_n / np.float32(128.0)
    grid_off = N.layout(N.POS_HASHGRID, 2)[1]
    rel_mlp = np.linalg.norm(g_n[:grid_off] - g_t[:grid_off]) / np.linalg.norm(g_t[:grid_off])
    rel_grid = np.linalg.norm(g_n[grid_off:] - g_t[grid_off:]) / np.linalg.norm(g_t[grid_off:])
    assert rel_mlp <= 1e-2 and 1e-2 <= rel_grid <= 8e-2, (rel_mlp, rel_grid)


def test_autograd_gradient_is_a_descent_direction():
    """Independent of either backward pass: with the loss's normaliser held at its value at p (it is a constant of the loss by
    definition), a small step against the autograd gradient lowers the loss by step * |g|^2 to first order."""
    rng = np.random.default_rng(23)
    p = _params(rng, N.POS_HASHGRID, 2)
    x = _inputs(rng, 512)
    t = _targets(x)
    m0 = T.Model(p, N.POS_HASHGRID, 2)
    norm = m0.normaliser(__import__("torch").as_tensor(x))
    loss0, g = m0.loss_and_gradient(x, t)
    g = g.numpy().astype(np.float64)
    step = 1e-4 / np.linalg.norm(g)
    m1 = T.Model((p - step * g).astype(np.float32), N.POS_HASHGRID, 2)
    loss1 = float(m1.loss(x, t, normaliser=norm).detach())
    predicted = step * float(np.dot(g, g))
    assert loss1 < loss0 and abs((loss0 - loss1) - predicted) <= 0.15 * predicted, (loss0, loss1, predicted)
