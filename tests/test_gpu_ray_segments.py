"""-m gpu: the scheduling forms of k_trace that small launches use -- ray segments (SEG lanes per ray, trace.hip) and the
occluder hint of slot-addressed shadow rays -- change no result: every form against the oracle and against the plain form."""
import numpy as np
import pytest

from gfxexp_amd import api
from tests import util
from tests.test_gpu_trace import _gpu_trace, _tri_ids, _compare_closest
from tests.test_gpu_restir import run_sequence_both

pytestmark = pytest.mark.gpu


def _awkward_rays(hs, n, seed):
    """Rays that start inside, outside and far outside the scene's box, with finite / huge / empty intervals and tmin > 0."""
    rng = np.random.default_rng(seed)
    b = hs.bounds()
    ext = b[3:] - b[:3]
    lo, hi = b[:3] - 0.6 * ext, b[3:] + 0.6 * ext
    p0 = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
    p1 = rng.uniform(b[:3], b[3:], (n, 3)).astype(np.float32)
    d = p1 - p0
    dist = np.linalg.norm(d, axis=1).astype(np.float32)
    org = np.zeros((n, 4), np.float32); org[:, :3] = p0
    dirs = np.zeros((n, 4), np.float32); dirs[:, :3] = d / dist[:, None]; dirs[:, 3] = dist * np.float32(0.9999)
    dirs[::3, 3] = 1e10                          # shadow rays to the environment / primary rays
    dirs[1::13, 3] = np.float32(np.inf)
    org[::7, 3] = 0.25 * dist[::7]               # tmin > 0
    dirs[::11, 3] = 0.0                          # empty intervals
    dirs[5::17, :3] *= -1.0                      # pointing away from the scene
    axis = np.arange(n) % 29 == 0                # axis-parallel directions (zero components)
    dirs[axis, 0] = 0.0; dirs[axis, 1] = 0.0; dirs[axis, 2] = -1.0
    return org, dirs


@pytest.mark.parametrize("segments", [1, 2, 4, 8])
def test_every_segment_count_returns_the_oracles_hits(built_lib, segments):
    hs = util.small_street()
    ctx = api.Context(0)
    ctx.tunable_set("trace_segments", segments)
    hs.upload(ctx)
    accel = ctx.accel_build()
    osc = util.feed_oracle(hs)
    org, dirs = util.pinhole_rays(256, 160, (2.0, 5.0, 26.0), (0.0, 3.0, 0.0), fov_y_deg=60.0)
    gpu = _gpu_trace(ctx, accel, api.TRACE_CLOSEST, org, dirs)
    _compare_closest(gpu, _tri_ids(ctx, accel), osc.trace(0, org, dirs), osc.tri_ids(), f"street, {segments} segments")
    org, dirs = _awkward_rays(hs, 60000, 11)
    gpu = _gpu_trace(ctx, accel, api.TRACE_CLOSEST, org, dirs)
    _compare_closest(gpu, _tri_ids(ctx, accel), osc.trace(0, org, dirs), osc.tri_ids(), f"awkward rays, {segments} segments")
    occ = _gpu_trace(ctx, accel, api.TRACE_ANY, org, dirs)
    ref = osc.trace(1, org, dirs)
    assert np.array_equal(occ != 0, ref != 0), f"any-hit differs on {np.count_nonzero((occ != 0) != (ref != 0))} rays"
    assert 0.05 < (occ != 0).mean() < 0.95
    # ragged sizes: fewer rays than one group, one more than a ticket batch
    for n in (1, 3, 65, 4097):
        g = _gpu_trace(ctx, accel, api.TRACE_CLOSEST, org[:n], dirs[:n])
        _compare_closest(g, _tri_ids(ctx, accel), osc.trace(0, org[:n], dirs[:n]), osc.tri_ids(), f"{n} rays, {segments} segments")
    # the counters of a segmented launch: every ray counted once, at least as many node fetches as the plain launch
    _, cnt = _gpu_trace(ctx, accel, api.TRACE_CLOSEST, org, dirs, counters=True)
    live = (dirs[:, 3] > org[:, 3])
    assert cnt[2] == np.count_nonzero(live), (cnt, np.count_nonzero(live))


@pytest.mark.parametrize("segments", [2, 8])
def test_empty_scene_and_single_triangle_with_segments(built_lib, segments):
    ctx = api.Context(0)
    ctx.tunable_set("trace_segments", segments)
    accel = ctx.accel_build()                     # empty scene: every ray misses and reports its own tmax
    org, dirs = util.pinhole_rays(8, 8, (0, 0, 5), (0, 0, 0))
    hits = _gpu_trace(ctx, accel, api.TRACE_CLOSEST, org, dirs)
    assert np.all(hits["triIndex"] == api.GFX_INVALID_SLOT)
    util.assert_same_bits("empty scene dist", hits["dist"], dirs[:, 3])
    assert np.all(_gpu_trace(ctx, accel, api.TRACE_ANY, org, dirs) == 0)
    hs = api.HostScene()
    mat = hs.add_material_traditional((0.5, 0.5, 0.5), (0, 0, 0), 0.1)
    v = np.zeros(3, api.VERTEX_DTYPE)
    v["position"] = [(-1, -1, 0), (1, -1, 0), (0, 1, 0)]
    v["normal"] = (0, 0, 1); v["texCoord0Dir"] = (1, 0, 0)
    hs.add_instance(hs.add_group([hs.add_geom(v, [(0, 1, 2)], mat)]), api.make_transform())
    ctx2 = api.Context(0)
    ctx2.tunable_set("trace_segments", segments)
    hs.upload(ctx2)
    accel2 = ctx2.accel_build()
    org, dirs = util.pinhole_rays(32, 32, (0, 0, 4), (0, 0, 0))
    osc = util.feed_oracle(hs, brute_force=True)
    _compare_closest(_gpu_trace(ctx2, accel2, api.TRACE_CLOSEST, org, dirs), _tri_ids(ctx2, accel2),
                     osc.trace(2, org, dirs), osc.tri_ids(), "single triangle")


@pytest.mark.parametrize("segments,hints", [(1, 0), (1, 2), (2, 1), (8, 2), (0, 1)])
def test_restir_frames_under_every_trace_form(built_lib, monkeypatch, segments, hints):
    """Three frames of original ReSTIR (temporal + spatial reuse) against the oracle, every buffer after every pass, with the
    trace launches cut into `segments` pieces per ray and the occluder hints off / clearing / keeping."""
    monkeypatch.setenv("GFX_TRACE_SEGMENTS", str(segments))
    monkeypatch.setenv("GFX_ANY_HINTS", str(hints))
    diffs = run_sequence_both(util.small_street(), 192, 108, frames=3, renderer=api.RENDERER_BIASED, scene_kind="street")
    assert not diffs, "\n".join(diffs)
    diffs = run_sequence_both(util.bunny_scene(), 150, 91, frames=3, renderer=api.RENDERER_UNBIASED)
    assert not diffs, "\n".join(diffs)


def test_occluder_hints_save_work_and_change_nothing(built_lib, monkeypatch):
    """Static scene, static camera: from the second frame on a shadow ray that was occluded one frame ago starts at its old
    occluder.  Same frames bit for bit with the hints off and on; fewer any-hit item fetches with them on."""
    import torch
    monkeypatch.setenv("GFX_SERIAL_FRAMES", "1")
    out = {}
    for hints in (0, 1, 2):
        ctx = api.Context(0)
        ctx.tunable_set("any_hints", hints)
        ctx.tunable_set("trace_segments", 1)
        hs = util.small_street()
        hs.upload(ctx)
        cfg = api.RestirRenderer.default_config(320, 180, api.RENDERER_BIASED)
        cfg.camera = api.make_camera(320, 180, pos=(2.0, 5.0, 26.0), pitch=4.0, yaw=180.0)
        r = api.RestirRenderer(ctx, cfg)
        ctx.counters_enable(True)
        per_frame = []
        for _ in range(4):
            ctx.counters_read(reset=True)
            r.render_frame()
            torch.cuda.synchronize()
            c = ctx.counters_read(reset=True)["any"]
            per_frame.append(c["nodeFetches"] + c["triFetches"])
        out[hints] = per_frame
        out[("beauty", hints)] = ctx.read_device(r.beauty_ptr(), 320 * 180 * 16).copy()
        r.close()
    assert out[0][0] == out[1][0] == out[2][0], out                # nothing to hint at in the first frame
    assert (out[("beauty", 0)] == out[("beauty", 1)]).all() and (out[("beauty", 0)] == out[("beauty", 2)]).all()
    print("any-hit items per frame, hints off / clearing / keeping:", out[0], out[1], out[2])
    assert out[1][3] < out[0][3], out
