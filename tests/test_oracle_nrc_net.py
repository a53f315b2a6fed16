"""CPU checks of the NRC network restatement (oracle/nrc_net.py): encoding invariants, analytic
gradients against finite differences (fp32 mode), and that training fits a smooth target."""
import numpy as np

from oracle import nrc_net as N


def _inputs(rng, n):
    x = rng.random((n, 14)).astype(np.float32)
    x[:, 3:8] = x[:, 3:8] * 6 - 3          # raw angles / roughness as the reference feeds them
    return x


def test_level_table_matches_the_published_grid_rule():
    levels, total = N.grid_levels()
    assert [lv[1] for lv in levels[:4]] == [16, 32, 64, 128]
    assert levels[0][2] == 4096 and all(lv[2] == 1 << 15 for lv in levels[1:])
    assert total == 4096 + 15 * 32768
    table, grid_off, n_params = N.layout(N.POS_HASHGRID, 2)
    assert grid_off == 2 * 4096 + 16 * 64 and n_params == grid_off + 2 * total
    assert N.layout(N.POS_TRIANGLEWAVE, 5)[2] == 5 * 4096 + 1024


def test_oneblob_is_a_partition_of_unity_on_the_unit_interval():
    rng = np.random.default_rng(3)
    x = rng.random((256, 5)).astype(np.float32)
    e = N.encode_oneblob(x)
    assert np.allclose(e.reshape(256, 5, 4).sum(axis=2), 1.0, atol=2e-6)
    assert np.all(e >= -1e-6)
    # the kernel wraps once (+-1), so the two boundary points of the interval encode alike
    a = N.encode_oneblob(np.full((1, 5), 1e-4, np.float32))
    b = N.encode_oneblob(np.full((1, 5), 1 - 1e-4, np.float32))
    assert np.allclose(a, b, atol=2e-3)
    # raw angles outside [0, 1) (what the reference feeds) stay finite and bounded
    wild = N.encode_oneblob((rng.random((64, 5)) * 8 - 4).astype(np.float32))
    assert np.all(np.isfinite(wild)) and wild.min() >= -1e-6 and wild.max() <= 1 + 1e-6


def test_hash_grid_trilinear_weights_and_dense_levels():
    rng = np.random.default_rng(4)
    x = rng.random((512, 3)).astype(np.float32)
    corners = N.hash_corners(x)
    levels, total = N.grid_levels()
    for (idx, w), (scale, res, n, off) in zip(corners, levels):
        assert np.allclose(w.sum(axis=1), 1.0, atol=1e-5)
        assert idx.min() >= off and idx.max() < off + n
    # a grid that stores its own x coordinate at the dense level 0 is reproduced exactly by interpolation
    grid = np.zeros((total, 2), np.float32)
    res = levels[0][1]
    ii = np.arange(res ** 3)
    grid[:res ** 3, 0] = (ii % res).astype(np.float32)
    enc = N.encode_hashgrid(x, grid)
    inside = x[:, 0] < 0.96                      # the +1 corner of the last cell wraps into the next row
    assert np.allclose(enc[inside, 0], x[inside, 0] * 15.0 + 0.5, rtol=1e-5, atol=1e-5)


def test_triangle_wave_range_and_frequency_doubling():
    x = np.linspace(0, 1, 97, dtype=np.float32)[:, None].repeat(3, 1)
    e = N.encode_trianglewave(x)
    assert e.min() >= -1 - 1e-6 and e.max() <= 1 + 1e-6
    assert np.allclose(e[:, 1], N.encode_trianglewave(x * 2)[:, 0], atol=1e-5)


def test_analytic_gradients_match_finite_differences_in_fp32_mode():
    rng = np.random.default_rng(5)
    net = N.NrcNet(N.POS_HASHGRID, 2, bf16=False)
    net.params[net.grid_off:] *= 1000.0          # make the grid matter
    x, t = _inputs(rng, 256), rng.random((256, 3)).astype(np.float32)
    loss0, g = net.gradients(x, t, loss_scale=1.0)

    def loss_at(p):
        pred = net.forward(x, p)
        # the loss treats the normaliser as a constant (its gradient ignores d denom / d pred)
        lum = 0.299 * pred0[:, 0] + 0.587 * pred0[:, 1] + 0.114 * pred0[:, 2]
        return float((((pred - t) ** 2) / (lum * lum + 0.01)[:, None]).sum() / (256 * 3))

    pred0 = net.forward(x, net.params).astype(np.float64)
    t = t.astype(np.float64)
    checked = 0
    touched = np.nonzero(g[net.grid_off:])[0]
    for k in list(rng.integers(0, net.grid_off, 12)) + list(net.grid_off + rng.choice(touched, 6)):
        eps = 2e-2 if k < net.grid_off else 5e-2
        p1, p2 = net.params.copy(), net.params.copy()
        p1[k] += eps; p2[k] -= eps
        fd = (loss_at(p1) - loss_at(p2)) / (2 * eps)
        if abs(fd) < 1e-4 and abs(g[k]) < 1e-4:
            continue
        assert abs(fd - g[k]) <= 0.08 * max(abs(fd), abs(g[k])) + 2e-4, (k, fd, g[k])
        checked += 1
    assert checked >= 8


def test_training_fits_a_smooth_target():
    rng = np.random.default_rng(6)
    net = N.NrcNet(N.POS_HASHGRID, 2)

    def batch(n):
        x = _inputs(rng, n)
        t = np.stack([np.sin(6 * x[:, 0]) * 0.5 + 0.5, x[:, 1] * x[:, 8], 0.3 + 0.2 * np.cos(9 * x[:, 2])], 1).astype(np.float32)
        return x, t
    losses = [net.train(*batch(2048)) for _ in range(40)]
    assert losses[-1] < 0.05 * losses[0]
    x, t = batch(2048)
    assert np.abs(net.infer(x) - t).mean() < 0.3
    assert not np.array_equal(net.ema, net.params)
