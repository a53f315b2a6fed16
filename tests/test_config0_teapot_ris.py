"""BASELINE configs[0] part (ii), CPU only: the teapot emitter set -- all 15 704 teapot
triangles emit RGB(1, 1, 1), importance = area (compute_light_probs.cu:33-43) -- sampled by streaming RIS with
M = 32 candidates per stream (optix_restir_di_kernels.cu:57-122: candidates from sampleLight, target = mean RGB of
the unshadowed contribution, Reservoir::update, W = sum w / (M * p_hat)) from one Lambert shading point
p = (0, 80, 0), rho = 0.5, against the exhaustive expectation: the integral of f * Le * G over every triangle by
quadrature.  (SURVEY names n = (0, 1, 0); the teapot spans y = 0 .. 79, so that normal faces away from every emitter
and the expectation is identically zero -- the harness uses n = (0, -1, 0), the point looking down at the lid.)  Pins the light-sampling plumbing (three-level distribution, area density,
reservoir law, 1/M estimator) end to end with the oracle; part (i), the notebook replay, is test_oracle_golden.py."""
import numpy as np

from oracle import oracle as O
from tests import util

P = np.array([0.0, 80.0, 0.0])
N = np.array([0.0, -1.0, 0.0])
RHO = 0.5
M = 32


def _contribution(pos, nrm, emittance):
    """f * Le * G (performDirectLighting, restir_di_shared.h:518-557, unshadowed) for Lambert at P."""
    d = pos - P
    d2 = np.sum(d * d, -1)
    w = d / np.sqrt(d2)[..., None]
    lp_cos = -np.sum(w * nrm, -1)
    sp_cos = np.sum(w * N, -1)
    g = np.where(lp_cos > 0, lp_cos * np.abs(sp_cos) / d2, 0.0)
    f = np.where(sp_cos > 0, RHO / np.pi, 0.0)          # Lambert evaluate: vGiven.z * vSampled.z > 0 (vOut = +n side)
    return (f * g)[..., None] * (emittance / np.pi)


def test_teapot_emitter_set_importance_is_area_and_ris_matches_exhaustive_expectation():
    hs = util.teapot_scene(emissive=True)
    assert hs.counts()["triangles"] == 15704
    osc = util.feed_oracle(hs, brute_force=True)
    tris = osc.world_triangles().astype(np.float64)
    nrm = np.cross(tris[:, 1] - tris[:, 0], tris[:, 2] - tris[:, 0])
    area = 0.5 * np.linalg.norm(nrm, axis=1)

    # importance = area: every level-2 distribution holds its triangles' areas (luminance of RGB(1,1,1) is 1)
    ids = osc.tri_ids()
    got = np.zeros(len(tris))
    for gi in np.unique(ids["geomInstSlot"]):
        w, cdf, integral = osc.lights_read(2, int(gi))
        sel = ids["geomInstSlot"] == gi
        assert len(w) == sel.sum()
        got[np.nonzero(sel)[0][np.argsort(ids["primIndex"][sel])]] = w
        assert abs(integral - w.astype(np.float64).sum()) <= 1e-5 * integral
    np.testing.assert_allclose(got, area, rtol=2e-5)

    # exhaustive expectation: every triangle cut into S x S congruent sub-triangles, integrand at their centroids
    # (S = 24 for the triangles close to the shading point, where 1 / d^2 varies quickly, S = 3 elsewhere), with the
    # interpolated and normalised vertex normals sampleLight uses (restir_di_shared.h:496-502)
    vn = _vertex_normals(hs, ids)
    dist = np.linalg.norm(tris.mean(1) - P, axis=1)
    size = np.sqrt(area)
    near = dist < 12 * size
    exact = np.zeros(3)
    for sel, S in ((near, 24), (~near, 3)):
        t, n3, ar = tris[sel], vn[sel], area[sel]
        for i in range(S):
            for j in range(S - i):
                for up in (0, 1):
                    if up and j >= S - i - 1:
                        continue
                    # centroid of the sub-triangle in barycentric coordinates
                    b1, b2 = ((i + 1 / 3 + up / 3) / S, (j + 1 / 3 + up / 3) / S)
                    b0 = 1 - b1 - b2
                    pos = b0 * t[:, 0] + b1 * t[:, 1] + b2 * t[:, 2]
                    n = b0 * n3[:, 0] + b1 * n3[:, 1] + b2 * n3[:, 2]
                    n /= np.linalg.norm(n, axis=1, keepdims=True)
                    exact += np.sum(_contribution(pos, n, np.ones(3)) * (ar / (S * S))[:, None], 0)
    assert exact.mean() > 0

    # streaming RIS, K streams of M candidates through the oracle's sampleLight and Reservoir::update
    K = 60000
    rng = np.random.default_rng(7)
    u = rng.random((M * K, 3)).astype(np.float32)
    ls, pd = osc.sample_light(P.astype(np.float32), u)
    cont = _contribution(ls[:, 3:6].astype(np.float64), ls[:, 6:9].astype(np.float64), ls[:, 0:3].astype(np.float64))
    target = cont.mean(-1)                                   # convertToWeight: (r + g + b) / 3
    weight = np.where(pd > 0, target / np.maximum(pd, 1e-30), 0.0).astype(np.float32).reshape(M, K)
    sel, sum_w, length = O.reservoir_stream(weight, rng.random((M, K)).astype(np.float32))
    assert np.all(length == M)
    valid = sel >= 0
    idx = np.where(valid, sel, 0) * K + np.arange(K)
    w_est = np.where(valid, sum_w / (M * np.maximum(target[idx], 1e-300)), 0.0)    # recPDFEstimate, :116-122
    per_stream = (cont[idx] * w_est[:, None]).mean(-1)
    estimate, est_err = per_stream.mean(), per_stream.std() / np.sqrt(K)
    # plain Monte Carlo over the same candidates (M = 1): the same expectation, far more variance (the point sits
    # three units above the lid, 1 / d^2 is heavy-tailed)
    per_sample = np.where(pd > 0, target / np.maximum(pd, 1e-30), 0.0)
    plain, plain_err = per_sample.mean(), per_sample.std() / np.sqrt(len(per_sample))
    assert abs(plain - exact.mean()) < 4 * plain_err + 0.01 * exact.mean(), (plain, plain_err, exact.mean())
    assert abs(estimate - exact.mean()) < 4 * est_err + 0.01 * exact.mean(), (estimate, est_err, exact.mean())
    assert est_err < 0.05 * exact.mean()          # 60 000 streams of 32 candidates pin the mean to a few per cent
    # RIS with 32 candidates: the per-stream variance is far below the per-sample variance of one candidate
    assert per_stream.var() < 0.2 * per_sample.var()


def _vertex_normals(hs, ids):
    """Object-space vertex normals per flattened triangle (identity instance transform)."""
    geoms = hs.geoms()
    out = np.zeros((len(ids), 3, 3))
    for gi, (v, t, _) in enumerate(geoms):
        sel = np.nonzero(ids["geomInstSlot"] == gi)[0]
        out[sel] = v["normal"][t[ids["primIndex"][sel]]]
    return out
