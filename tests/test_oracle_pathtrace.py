"""CPU checks of the path-tracer restatement (oracle/orc_pathtrace.h).

The reference holds no golden image for the path tracer, so the restatement is pinned through
estimator identities against the ReSTIR restatement, which is itself pinned against the
reference's notebook vectors (test_oracle_golden.py):
  * maxPathLength = 2 is a direct-lighting estimator (emission + MIS-combined NEE / BSDF hit), so
    its expectation equals the expectation of the unbiased ReSTIR renderer;
  * longer paths only add non-negative energy and converge (maxPathLength 5 vs 8 nearly equal).
"""
import numpy as np

from tests import util
from oracle import oracle as O
from gfxexp_amd import api


def _camera(width, height):
    return api.make_camera(width, height, pos=(1.5, 5.0, 14.0), pitch=12.0, yaw=186.0)


def _pt_mean_image(osc, width, height, max_len, frames, env=None, seed=util.PIXEL_RNG_SEED):
    pb = util.PixelBuffers(width, height, seed=seed)
    if env is not None:
        pb.set_env(*env)
    s = pb.host_static_params()
    cam = util.copy_struct(O.GfxCamera, _camera(width, height))
    for frame in range(frames):
        f = util.frame_params(O.GfxRestirFrameParams, O.GfxCamera, width, height, cam, frameIndex=frame,
                              bufferIndex=frame % 2, resetFlowBuffer=int(frame == 0), numAccumFrames=frame,
                              enableJittering=0, enableEnvLight=int(env is not None))
        osc.pt_launch(s, f, 0, max_len)
        osc.pt_launch(s, f, 1, max_len)
    return pb.beauty[:, :3].copy(), pb.gb0[(frames - 1) % 2]["instSlot"].copy(), pb


def _restir_mean_image(osc, width, height, frames):
    pb = util.PixelBuffers(width, height)
    s = pb.host_static_params()
    cam = util.copy_struct(O.GfxCamera, _camera(width, height))
    for frame in range(frames):
        f = util.frame_params(O.GfxRestirFrameParams, O.GfxCamera, width, height, cam, frameIndex=frame,
                              bufferIndex=frame % 2, resetFlowBuffer=1, numAccumFrames=frame,
                              enableTemporalReuse=0, enableSpatialReuse=0, useUnbiasedEstimator=1, reuseVisibility=1)
        osc.restir_launch(s, f, frame % 2, 0, api.PASS_SETUP_GBUFFERS)
        osc.restir_launch(s, f, frame % 2, 0, api.PASS_INITIAL_RIS)
        osc.restir_launch(s, f, frame % 2, 0, api.PASS_SHADING)
    return pb.beauty[:, :3].copy()


def test_direct_only_path_tracer_matches_restir_expectation():
    hs = util.bunny_scene()
    osc = util.feed_oracle(hs)
    w, h = 48, 32
    pt, inst, _ = _pt_mean_image(osc, w, h, 2, 96)
    rs = _restir_mean_image(osc, w, h, 96)
    surf = inst != 0xFFFFFFFF
    assert surf.sum() > 0.5 * w * h
    a, b = pt[surf].mean(axis=0), rs[surf].mean(axis=0)
    assert np.all(np.isfinite(pt)) and np.all(pt >= 0)
    assert np.allclose(a, b, rtol=0.04), (a, b)


def test_longer_paths_add_energy_and_converge():
    hs = util.bunny_scene()
    osc = util.feed_oracle(hs)
    w, h = 48, 32
    m2, inst, _ = _pt_mean_image(osc, w, h, 2, 48)
    m5, _, _ = _pt_mean_image(osc, w, h, 5, 48)
    m8, _, _ = _pt_mean_image(osc, w, h, 8, 48)
    surf = inst != 0xFFFFFFFF
    e2, e5, e8 = m2[surf].mean(), m5[surf].mean(), m8[surf].mean()
    assert e5 > e2 * 1.02
    assert abs(e8 - e5) < 0.1 * e5


def test_rng_draw_budget_per_path():
    """Each pixel draws 5 numbers at the first hit and at most 6 per further vertex (RR + NEE 3 + BSDF 2);
    background pixels draw none -- guards the draw order the GPU wavefront has to reproduce."""
    hs = util.bunny_scene()
    osc = util.feed_oracle(hs)
    w, h = 32, 24
    _, inst, pb = _pt_mean_image(osc, w, h, 4, 1)
    start = O.seed_rngs(w * h, util.PIXEL_RNG_SEED)
    bg = inst == 0xFFFFFFFF
    assert np.array_equal(pb.rng[bg], start[bg])
    # advance every start state by k draws and find k for surface pixels
    states = start.copy()
    steps = np.full(w * h, -1, np.int64)
    for k in range(0, 5 + 6 * 3 + 1):
        hit = (states == pb.rng) & (steps < 0)
        steps[hit] = k
        states = states * np.uint64(6364136223846793005) + np.uint64(1)
    s = steps[~bg]
    assert np.all(s >= 5)
    allowed = {5} | {5 + 1 + 6 * j for j in range(0, 3)} | {5 + 6 * j for j in range(0, 3)} | {5 + 6 * 2 + 1}
    assert set(np.unique(s)).issubset(allowed), np.unique(s)


def test_solid_angle_sampling_has_the_same_expectation_and_less_noise_near_lights():
    """sampleLight<true> (restir_di_shared.h:417-483) draws the point uniformly in the solid angle of the selected
    triangle; with the matching hypothetical density in computeSurfacePoint (path_tracing_shared.h:550-568) the MIS
    estimator keeps its expectation.  Direct lighting only (maxPathLength 2), many frames, both modes."""
    hs = util.bunny_scene()
    osc = util.feed_oracle(hs)
    w, h = 48, 32
    area, inst, _ = _pt_mean_image(osc, w, h, 2, 128)
    with util.frame_overrides(useSolidAngleSampling=1):
        solid, _, _ = _pt_mean_image(osc, w, h, 2, 128)
    surf = inst != 0xFFFFFFFF
    assert np.all(np.isfinite(solid)) and np.all(solid >= 0)
    assert not np.array_equal(solid, area)
    a, b = area[surf].mean(axis=0), solid[surf].mean(axis=0)
    assert np.allclose(a, b, rtol=0.04), (a, b)
