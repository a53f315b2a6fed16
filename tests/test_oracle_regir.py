"""CPU checks of the ReGIR restatement (oracle/orc_regir.h + the REGIR branch of orc_pathtrace.h)."""
import numpy as np

from gfxexp_amd import api
from oracle import oracle as O
from tests import util


def _camera(width, height):
    return api.make_camera(width, height, pos=(1.5, 5.0, 14.0), pitch=12.0, yaw=186.0)


def regir_frames(osc, hs, width, height, frames, max_len, temporal=True, accumulate=True, dims=(8, 4, 8)):
    pb = util.PixelBuffers(width, height)
    rb = util.RegirBuffers(hs.bounds(), dims=dims)
    s = pb.host_static_params()
    osc.regir_set_params(rb.host_params())
    cam = util.copy_struct(O.GfxCamera, _camera(width, height))
    for frame in range(frames):
        f = util.frame_params(O.GfxRestirFrameParams, O.GfxCamera, width, height, cam, frameIndex=frame,
                              bufferIndex=frame % 2, resetFlowBuffer=int(frame == 0),
                              numAccumFrames=frame if accumulate else 0)
        build = api.PT_REGIR_BUILD_CELLS_TEMPORAL if (temporal and frame > 0) else api.PT_REGIR_BUILD_CELLS
        for pass_id in (api.PT_SETUP_GBUFFERS, build, api.PT_PATH_TRACE_REGIR, api.PT_REGIR_UPDATE_LAST_ACCESS):
            osc.pt_launch(s, f, pass_id, max_len)
    return pb, rb


def test_grid_bookkeeping_and_slot_rng_budget():
    hs = util.bunny_scene()
    osc = util.feed_oracle(hs)
    w, h = 32, 24
    pb, rb = regir_frames(osc, hs, w, h, 1, 3)
    # frame 0: every cell is active (0 - 0xFFFFFFFF = 1 <= 8); each slot draws 4 numbers per candidate
    adv = O.seed_rngs(rb.slots, util.PIXEL_RNG_SEED)
    for _ in range(8 * 4):
        adv = adv * np.uint64(6364136223846793005) + np.uint64(1)
    assert np.array_equal(adv, rb.rngs)
    m = rb.res[0][2][:, 3].view(np.uint32)
    assert np.all(m == 8)
    touched = rb.accesses > 0
    assert 0 < touched.sum() < rb.cells
    assert np.array_equal(rb.last_access[touched], np.zeros(touched.sum(), np.uint32))
    assert np.all(rb.last_access[~touched] == 0xFFFFFFFF)
    assert rb.active[0][0] == touched.sum()
    # every shaded vertex touches exactly one cell: accesses >= surface pixels
    surf = pb.gb0[0]["instSlot"] != 0xFFFFFFFF
    assert rb.accesses.sum() >= surf.sum()


def test_cells_go_inactive_after_eight_untouched_frames_and_temporal_growth():
    hs = util.bunny_scene()
    osc = util.feed_oracle(hs)
    pb, rb = regir_frames(osc, hs, 32, 24, 10, 2)
    cur = (10 - 1) % 2
    m = rb.res[cur][2][:, 3].view(np.uint32).reshape(rb.cells, 512)
    touched = rb.last_access == 9
    never = rb.last_access == 0xFFFFFFFF
    assert touched.any() and never.any()
    # touched cells accumulate M = 8 + min(prev, 160) per built frame (80 after 10 frames); a cell first
    # touched at frame 8 skipped one build and merges the stale frame-6 reservoirs (8 + 56)
    assert m[touched].max() == 80 and m[touched].min() >= 64
    assert m[never].max() == 64      # built for frames 0..7 only
    rec = rb.info[cur][:, 0]
    assert np.all(np.isfinite(rec)) and np.all(rec >= 0)


def test_regir_direct_lighting_close_to_baseline_direct_lighting():
    """maxPathLength = 2 with ReGIR is emission + one ReGIR NEE per pixel: a direct-lighting estimator
    whose mean must be close to the baseline path tracer's (ReGIR's cell target ignores visibility and
    BSDF, which costs variance, not energy; grid-induced bias stays within a few percent here)."""
    hs = util.bunny_scene()
    osc = util.feed_oracle(hs)
    w, h, frames = 48, 32, 64
    pb, rb = regir_frames(osc, hs, w, h, frames, 2)
    surf = pb.gb0[(frames - 1) % 2]["instSlot"] != 0xFFFFFFFF
    got = pb.beauty[surf, :3].mean()
    pb2 = util.PixelBuffers(w, h)
    s = pb2.host_static_params()
    cam = util.copy_struct(O.GfxCamera, _camera(w, h))
    for frame in range(frames):
        f = util.frame_params(O.GfxRestirFrameParams, O.GfxCamera, w, h, cam, frameIndex=frame, bufferIndex=frame % 2,
                              resetFlowBuffer=int(frame == 0), numAccumFrames=frame)
        osc.pt_launch(s, f, api.PT_SETUP_GBUFFERS, 2)
        osc.pt_launch(s, f, api.PT_PATH_TRACE_BASELINE, 2)
    want = pb2.beauty[surf, :3].mean()
    assert np.all(np.isfinite(pb.beauty))
    assert abs(got - want) < 0.12 * want, (got, want)
