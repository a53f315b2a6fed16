"""-m gpu: the dual-item form of the traversal kernel (trace.hip DUAL: a lane tests the last pending leaf triangle of a node and
fetches the next node of its walk in the same wave iteration) against the oracle and against the one-item form: same hits bit
for bit, same triangle / ray counts, fewer wave iterations."""
import numpy as np
import pytest

from gfxexp_amd import api
from tests import util
from tests.test_gpu_trace import _gpu_trace, _tri_ids, _compare_closest
from tests.test_gpu_restir import run_sequence_both

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dual", [0, 1])
def test_both_forms_return_the_oracles_hits(built_lib, dual):
    hs = util.small_street(cluttered=True)
    ctx = api.Context(0)
    ctx.tunable_set("trace_dual", dual)
    hs.upload(ctx)
    accel = ctx.accel_build()
    osc = util.feed_oracle(hs)
    ids = _tri_ids(ctx, accel)
    org, dirs = util.pinhole_rays(320, 200, (2.0, 3.0, 28.0), (0.0, 3.0, 0.0), fov_y_deg=60.0)
    gpu = _gpu_trace(ctx, accel, api.TRACE_CLOSEST, org, dirs)
    _compare_closest(gpu, ids, osc.trace(0, org, dirs), osc.tri_ids(), f"cluttered street, dual {dual}")
    rng = np.random.default_rng(4)
    b = hs.bounds()
    n = 80000
    p0 = rng.uniform(b[:3], b[3:], (n, 3)).astype(np.float32)
    p1 = rng.uniform(b[:3], b[3:], (n, 3)).astype(np.float32)
    d = p1 - p0
    dist = np.linalg.norm(d, axis=1).astype(np.float32)
    o = np.zeros((n, 4), np.float32); o[:, :3] = p0
    dd = np.zeros((n, 4), np.float32); dd[:, :3] = d / dist[:, None]; dd[:, 3] = dist * np.float32(0.9999)
    o[::7, 3] = 0.3; dd[::11, 3] = 0.0
    occ = _gpu_trace(ctx, accel, api.TRACE_ANY, o, dd)
    ref = osc.trace(1, o, dd)
    assert np.array_equal(occ != 0, ref != 0), f"any-hit differs on {np.count_nonzero((occ != 0) != (ref != 0))} rays"
    gpu = _gpu_trace(ctx, accel, api.TRACE_CLOSEST, o, dd)
    _compare_closest(gpu, ids, osc.trace(0, o, dd), osc.tri_ids(), f"segments, dual {dual}")


def test_dual_iterations_save_wave_iterations_and_change_no_hit(built_lib):
    import torch
    hs = util.small_street(cluttered=True)
    org, dirs = util.pinhole_rays(640, 400, (2.0, 3.0, 28.0), (0.0, 3.0, 0.0), fov_y_deg=60.0)
    got = {}
    for dual in (0, 1):
        ctx = api.Context(0)
        ctx.tunable_set("trace_dual", dual)
        hs.upload(ctx)
        accel = ctx.accel_build()
        ctx.counters_enable(True)
        ctx.trace_diag_read(reset=True)
        hits, cnt = _gpu_trace(ctx, accel, api.TRACE_CLOSEST, org, dirs, counters=True)
        got[dual] = (hits.copy(), cnt.copy(), ctx.trace_diag_read(reset=True))
        n = len(org)
        d_org, d_dir = torch.from_numpy(org).cuda(), torch.from_numpy(dirs).cuda()
        d_out = torch.zeros(n * 4, dtype=torch.int32, device="cuda")
        d_cnt = torch.zeros(4, dtype=torch.int64, device="cuda")
        d_items = torch.full((n,), -1, dtype=torch.int32, device="cuda")
        ctx.trace(accel, api.TRACE_CLOSEST, d_org.data_ptr(), d_dir.data_ptr(), n, d_out.data_ptr(), d_cnt.data_ptr(), d_per_ray_items=d_items.data_ptr())
        torch.cuda.synchronize()
        items, c2 = d_items.cpu().numpy(), d_cnt.cpu().numpy()
        assert items.min() >= 1 and items.sum() == c2[0] + c2[1] and c2[2] == n
    assert np.array_equal(got[0][0].view(np.uint8), got[1][0].view(np.uint8))
    assert got[0][1][2] == got[1][1][2] == len(org)
    print("wave iterations one-item / dual:", got[0][2]["iterations"], got[1][2]["iterations"], "node fetches:", got[0][1][0], got[1][1][0])
    assert got[1][2]["iterations"] < 0.95 * got[0][2]["iterations"]
    assert got[1][1][0] <= 1.05 * got[0][1][0]            # the node chosen before the triangle result costs few extra visits


@pytest.mark.parametrize("dual", [0, 1])
def test_restir_frames_under_both_forms(built_lib, monkeypatch, dual):
    monkeypatch.setenv("GFX_TRACE_DUAL", str(dual))
    diffs = run_sequence_both(util.small_street(), 192, 108, frames=3, renderer=api.RENDERER_BIASED, scene_kind="street")
    assert not diffs, "\n".join(diffs)
    diffs = run_sequence_both(util.bunny_scene(), 150, 91, frames=2, renderer=api.RENDERER_UNBIASED)
    assert not diffs, "\n".join(diffs)
