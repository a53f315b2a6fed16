"""-m gpu: wave compaction of the traversal kernel (trace.hip: once the ray queue is dry, a wave left with few live rays hands them --
state and LDS stack column -- to the other waves of its block) changes no result: every threshold, with persistent grids small
enough that every block's waves all hold rays, against the oracle and against the kernel with compaction off."""
import numpy as np
import pytest

from gfxexp_amd import api
from tests import util
from tests.test_gpu_trace import _gpu_trace, _tri_ids, _compare_closest
from tests.test_gpu_restir import run_sequence_both

pytestmark = pytest.mark.gpu


def _mixed_rays(hs, n, seed):
    """Rays of very different length next to each other in the queue (so the lanes of a wave finish far apart): through the
    whole scene, short stubs, empty intervals, rays that leave the scene at once."""
    rng = np.random.default_rng(seed)
    b = hs.bounds()
    ext = b[3:] - b[:3]
    p0 = rng.uniform(b[:3] - 0.2 * ext, b[3:] + 0.2 * ext, (n, 3)).astype(np.float32)
    p1 = rng.uniform(b[:3], b[3:], (n, 3)).astype(np.float32)
    d = p1 - p0
    dist = np.linalg.norm(d, axis=1).astype(np.float32)
    org = np.zeros((n, 4), np.float32); org[:, :3] = p0
    dirs = np.zeros((n, 4), np.float32); dirs[:, :3] = d / dist[:, None]; dirs[:, 3] = 1e10
    stub = rng.random(n) < 0.5
    dirs[stub, 3] = (0.05 * dist[stub]).astype(np.float32)
    dirs[::11, 3] = 0.0
    org[::7, 3] = 0.1 * dist[::7]
    return org, dirs


@pytest.mark.parametrize("compact,blocks_per_cu,refill", [(0, 4, 8), (1, 1, 8), (16, 1, 8), (63, 1, 1), (32, 2, 16), (63, 4, 64)])
def test_every_compaction_threshold_returns_the_oracles_hits(built_lib, compact, blocks_per_cu, refill):
    hs = util.small_street(cluttered=True)
    ctx = api.Context(0)
    ctx.tunable_set("trace_compact", compact)
    ctx.tunable_set("trace_blocks_per_cu", blocks_per_cu)
    ctx.tunable_set("trace_refill", refill)
    hs.upload(ctx)
    accel = ctx.accel_build()
    osc = util.feed_oracle(hs)
    ids = _tri_ids(ctx, accel)
    for n, seed in ((150000, 3), (70001, 4), (4097, 5), (65, 6), (3, 7)):
        org, dirs = _mixed_rays(hs, n, seed)
        gpu = _gpu_trace(ctx, accel, api.TRACE_CLOSEST, org, dirs)
        _compare_closest(gpu, ids, osc.trace(0, org, dirs), osc.tri_ids(), f"{n} rays, compaction below {compact}")
        occ = _gpu_trace(ctx, accel, api.TRACE_ANY, org, dirs)
        ref = osc.trace(1, org, dirs)
        assert np.array_equal(occ != 0, ref != 0), f"any-hit differs on {np.count_nonzero((occ != 0) != (ref != 0))} of {n} rays"
    # every ray is counted once, wherever it finished
    org, dirs = _mixed_rays(hs, 150000, 3)
    _, cnt = _gpu_trace(ctx, accel, api.TRACE_CLOSEST, org, dirs, counters=True)
    assert cnt[2] == np.count_nonzero(dirs[:, 3] > org[:, 3]), cnt


def test_compaction_moves_rays_and_saves_iterations(built_lib):
    """The scheduling diagnostics of the counting launches: with compaction the same rays fetch the same items in fewer wave
    iterations (the drain phase runs at a higher lane occupancy)."""
    import torch
    hs = util.small_street(cluttered=True)
    org, dirs = _mixed_rays(hs, 400000, 9)
    got = {}
    for compact in (0, 32):
        ctx = api.Context(0)
        ctx.tunable_set("trace_compact", compact)
        ctx.tunable_set("trace_blocks_per_cu", 1)
        hs.upload(ctx)
        accel = ctx.accel_build()
        ctx.counters_enable(True)
        ctx.trace_diag_read(reset=True)
        hits, cnt = _gpu_trace(ctx, accel, api.TRACE_CLOSEST, org, dirs, counters=True)
        d = ctx.trace_diag_read(reset=True)
        got[compact] = (hits.copy(), cnt.copy(), d)
    assert np.array_equal(got[0][0].view(np.uint8), got[32][0].view(np.uint8))
    assert np.array_equal(got[0][1][:3], got[32][1][:3])          # same node fetches, triangle fetches, rays
    assert got[0][2]["itemLanes"] == got[32][2]["itemLanes"]
    print("wave iterations without / with compaction:", got[0][2]["iterations"], got[32][2]["iterations"])
    assert got[32][2]["iterations"] < got[0][2]["iterations"]


@pytest.mark.parametrize("compact", [0, 63])
def test_restir_frames_with_and_without_compaction(built_lib, monkeypatch, compact):
    monkeypatch.setenv("GFX_TRACE_COMPACT", str(compact))
    monkeypatch.setenv("GFX_TRACE_BLOCKS_PER_CU", "1")
    diffs = run_sequence_both(util.small_street(), 192, 108, frames=3, renderer=api.RENDERER_BIASED, scene_kind="street")
    assert not diffs, "\n".join(diffs)
    diffs = run_sequence_both(util.bunny_scene(), 150, 91, frames=2, renderer=api.RENDERER_UNBIASED)
    assert not diffs, "\n".join(diffs)
