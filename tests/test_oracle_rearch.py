"""CPU checks of the rearchitected-ReSTIR restatement (oracle/orc_restir_rearch.h).

No reference vectors exist for these kernels, so the restatement is pinned through estimator
identities against passes that are pinned (test_oracle_golden.py, test_oracle_pathtrace.py):
the unbiased rearchitected renderer, with and without temporal/spatial reuse, must have the same
expectation as the original unbiased renderer without reuse."""
import numpy as np

from gfxexp_amd import api
from oracle import oracle as O
from tests import util


def _camera(width, height):
    return api.make_camera(width, height, pos=(1.5, 5.0, 14.0), pitch=12.0, yaw=186.0)


def rearch_frames(osc, width, height, frames, temporal, spatial, unbiased, accumulate=True, low_discrepancy=True):
    pb = util.PixelBuffers(width, height)
    s = pb.host_static_params()
    cam = util.copy_struct(O.GfxCamera, _camera(width, height))
    last_res, last_base = 1, 0
    for frame in range(frames):
        f = util.frame_params(O.GfxRestirFrameParams, O.GfxCamera, width, height, cam, frameIndex=frame,
                              bufferIndex=frame % 2, resetFlowBuffer=int(frame == 0),
                              numAccumFrames=frame if accumulate else 0, numSpatialNeighbors=1,
                              enableTemporalReuse=int(temporal), enableSpatialReuse=int(spatial),
                              useUnbiasedEstimator=int(unbiased), useLowDiscrepancyNeighbors=int(low_discrepancy))
        cur = (last_res + 1) % 2
        trace_pass, shade_pass = api.rearch_passes(temporal, spatial, unbiased, frame == 0)
        for pass_id in (api.PASS_SETUP_GBUFFERS, api.PASS_LIGHT_PRESAMPLING, api.PASS_PER_PIXEL_RIS, trace_pass, shade_pass):
            osc.restir_launch(s, f, cur, last_base, pass_id)
        last_base += 1
        last_res = cur
    return pb


def _reference_mean(osc, width, height, frames):
    pb = util.PixelBuffers(width, height)
    s = pb.host_static_params()
    cam = util.copy_struct(O.GfxCamera, _camera(width, height))
    for frame in range(frames):
        f = util.frame_params(O.GfxRestirFrameParams, O.GfxCamera, width, height, cam, frameIndex=frame,
                              bufferIndex=frame % 2, resetFlowBuffer=1, numAccumFrames=frame,
                              enableTemporalReuse=0, enableSpatialReuse=0, useUnbiasedEstimator=1)
        for pass_id in (api.PASS_SETUP_GBUFFERS, api.PASS_INITIAL_RIS, api.PASS_SHADING):
            osc.restir_launch(s, f, frame % 2, 0, pass_id)
    return pb


def test_presampled_subsets_and_rng_budget():
    hs = util.bunny_scene()
    osc = util.feed_oracle(hs)
    pb = rearch_frames(osc, 32, 24, 1, False, False, False)
    # every pre-sampled light drew exactly 3 numbers
    start = O.seed_rngs(util.PRESAMPLED_LIGHTS, 894213312210)
    adv = start.copy()
    for _ in range(3):
        adv = adv * np.uint64(6364136223846793005) + np.uint64(1)
    assert np.array_equal(adv, pb.presample_rngs)
    ls = pb.presampled
    assert np.all(ls[:, 10] > 0) and np.all(ls[:, 11] == 0)          # area densities; pad word
    assert np.all(np.isfinite(ls)) and np.all(ls[:, :3].sum(axis=1) > 0)   # emitters only
    n = np.linalg.norm(ls[:, 6:9], axis=1)
    assert np.allclose(n, 1.0, atol=1e-5)


def test_unbiased_rearchitected_matches_reference_expectation():
    hs = util.bunny_scene()
    osc = util.feed_oracle(hs)
    w, h, frames = 48, 32, 64
    ref = _reference_mean(osc, w, h, frames)
    surf = ref.gb0[(frames - 1) % 2]["instSlot"] != 0xFFFFFFFF
    want = ref.beauty[surf, :3].mean(axis=0)
    for temporal, spatial in ((False, False), (True, False), (False, True), (True, True)):
        pb = rearch_frames(osc, w, h, frames, temporal, spatial, True)
        got = pb.beauty[surf, :3].mean(axis=0)
        assert np.all(np.isfinite(pb.beauty))
        assert np.allclose(got, want, rtol=0.05), (temporal, spatial, got, want)


def test_biased_rearchitected_is_darker_but_close_and_reuse_grows_stream_length():
    hs = util.bunny_scene()
    osc = util.feed_oracle(hs)
    w, h, frames = 48, 32, 24
    ref = _reference_mean(osc, w, h, frames)
    surf = ref.gb0[(frames - 1) % 2]["instSlot"] != 0xFFFFFFFF
    want = ref.beauty[surf, :3].mean()
    pb = rearch_frames(osc, w, h, frames, True, True, False, low_discrepancy=False)
    got = pb.beauty[surf, :3].mean()
    assert 0.75 * want < got < 1.05 * want, (got, want)
    # final reservoirs: M grows with reuse but stays under the 20x cap per merged neighbour
    cur = (1 + frames) % 2
    m = pb.res[cur][2][:, 3].view(np.uint32)[surf]
    assert m.max() > 32 and m.max() <= 32 * (1 + 20 + 20)
    vis = pb.vis[(frames - 1) % 2][surf]
    assert np.all(vis < (1 << 12))
