"""-m gpu: HIP LBVH->BVH8 build + wavefront traversal against the CPU oracle.

Bar: (instSlot, geomInstSlot, primIndex) and the hit distance / barycentrics are BIT-EXACT against
the oracle's closest hit (reference traversal + canonical tie-break) and against brute force.
"""
import numpy as np
import pytest

from gfxexp_amd import api
from tests import util

pytestmark = pytest.mark.gpu


def _gpu_trace(ctx, accel, mode, org, dirs, counters=False):
    import torch
    n = len(org)
    d_org = torch.from_numpy(org).cuda()
    d_dir = torch.from_numpy(dirs).cuda()
    if mode == api.TRACE_ANY:
        d_out = torch.zeros(n, dtype=torch.int32, device="cuda")
    else:
        d_out = torch.zeros(n * 4, dtype=torch.int32, device="cuda")
    d_cnt = torch.zeros(4, dtype=torch.int64, device="cuda") if counters else None
    ctx.trace(accel, mode, d_org.data_ptr(), d_dir.data_ptr(), n, d_out.data_ptr(), d_cnt.data_ptr() if counters else 0,
              stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    out = d_out.cpu().numpy()
    res = out.view(np.uint32) if mode == api.TRACE_ANY else out.view(api.HIT_DTYPE).reshape(n)
    if counters:
        return res, d_cnt.cpu().numpy()
    return res


def _tri_ids(ctx, accel):
    ptr, n = ctx.accel_tri_ids_ptr(accel)
    return ctx.read_device(ptr, n * 12).view(api.TRI_IDS_DTYPE).reshape(n)


def _compare_closest(gpu_hits, gpu_ids, orc_hits, orc_ids, what):
    g_hit = gpu_hits["triIndex"] != api.GFX_INVALID_SLOT
    o_hit = orc_hits["triIndex"] != api.GFX_INVALID_SLOT
    assert np.array_equal(g_hit, o_hit), f"{what}: hit/miss differs on {np.count_nonzero(g_hit != o_hit)} rays"
    gi = gpu_ids[gpu_hits["triIndex"][g_hit]]
    oi = orc_ids[orc_hits["triIndex"][o_hit]]
    for f in ("instSlot", "geomInstSlot", "primIndex"):
        assert np.array_equal(gi[f], oi[f]), f"{what}: {f} differs on {np.count_nonzero(gi[f] != oi[f])} rays"
    for f in ("dist", "bcB", "bcC"):
        util.assert_same_bits(f"{what}.{f}", gpu_hits[f][g_hit], orc_hits[f][o_hit])
    # misses report tmax
    util.assert_same_bits(f"{what}.miss dist", gpu_hits["dist"][~g_hit], orc_hits["dist"][~o_hit])


@pytest.mark.parametrize("scene_name,res", [("bunny", 512), ("teapot", 384)])
def test_closest_hit_matches_oracle(built_lib, scene_name, res):
    hs = util.bunny_scene(with_light=True) if scene_name == "bunny" else util.teapot_scene()
    ctx = api.Context(0)
    hs.upload(ctx)
    accel = ctx.accel_build()
    stats = ctx.accel_stats(accel)
    assert stats["triangles"] == hs.counts()["triangles"] == stats["triRecords"]
    osc = util.feed_oracle(hs)
    b = hs.bounds()
    centre = 0.5 * (b[:3] + b[3:])
    ext = np.linalg.norm(b[3:] - b[:3])
    org, dirs = util.pinhole_rays(res, res, centre + np.array([0.35, 0.3, 0.9]) * ext, centre)
    gpu_hits = _gpu_trace(ctx, accel, api.TRACE_CLOSEST, org, dirs)
    gpu_ids = _tri_ids(ctx, accel)
    orc_hits = osc.trace(0, org, dirs)
    orc_ids = osc.tri_ids()
    _compare_closest(gpu_hits, gpu_ids, orc_hits, orc_ids, scene_name)
    assert np.count_nonzero(gpu_hits["triIndex"] != api.GFX_INVALID_SLOT) > res * res // 20
    if scene_name == "bunny":
        brute = osc.trace(2, org, dirs)
        _compare_closest(gpu_hits, gpu_ids, brute, orc_ids, "bunny vs brute force")


def test_any_hit_and_intervals(built_lib):
    hs = util.bunny_scene(with_light=True)
    ctx = api.Context(0)
    hs.upload(ctx)
    accel = ctx.accel_build()
    osc = util.feed_oracle(hs)
    rng = np.random.default_rng(3)
    n = 100000
    b = hs.bounds()
    lo, hi = np.array([-8, 0.01, -8], np.float32), np.array([8, 12, 8], np.float32)
    p0 = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
    p1 = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
    d = p1 - p0
    dist = np.linalg.norm(d, axis=1).astype(np.float32)
    org = np.zeros((n, 4), np.float32); org[:, :3] = p0
    dirs = np.zeros((n, 4), np.float32); dirs[:, :3] = d / dist[:, None]; dirs[:, 3] = dist * np.float32(0.9999)
    org[::7, 3] = 0.5           # non-zero tmin on some rays
    dirs[::11, 3] = 0.0         # empty intervals
    gpu = _gpu_trace(ctx, accel, api.TRACE_ANY, org, dirs)
    ref = osc.trace(1, org, dirs)
    assert np.array_equal(gpu, ref), f"occlusion differs on {np.count_nonzero(gpu != ref)} of {n} rays"
    assert 0.02 < gpu.mean() < 0.98
    # the closest-hit kernel agrees with the any-hit kernel about "something in the interval"
    closest = _gpu_trace(ctx, accel, api.TRACE_CLOSEST, org, dirs)
    assert np.array_equal((closest["triIndex"] != api.GFX_INVALID_SLOT).astype(np.uint32), gpu)


def test_street_scene_and_counters(built_lib):
    hs = util.small_street()
    ctx = api.Context(0)
    hs.upload(ctx)
    accel = ctx.accel_build()
    osc = util.feed_oracle(hs)
    org, dirs = util.pinhole_rays(320, 200, (2.0, 6.0, 28.0), (0.0, 2.0, 0.0), fov_y_deg=60.0)
    (gpu_hits, counters) = _gpu_trace(ctx, accel, api.TRACE_CLOSEST, org, dirs, counters=True)
    gpu_ids = _tri_ids(ctx, accel)
    _compare_closest(gpu_hits, gpu_ids, osc.trace(0, org, dirs), osc.tri_ids(), "street")
    assert counters[2] == len(org)
    assert counters[0] >= len(org) and counters[1] > 0
    # leaf-size knob: a different tree, the same answers
    ctx.accel_set_max_leaf(2)
    accel2 = ctx.accel_build()
    h2 = _gpu_trace(ctx, accel2, api.TRACE_CLOSEST, org, dirs)
    _compare_closest(h2, _tri_ids(ctx, accel2), gpu_hits, gpu_ids, "street maxLeaf=2 vs 4")


def test_launches_on_alternating_streams_hand_the_ticket_areas_over(built_lib):
    """k_trace zeroes the ticket counters of the NEXT launch (no memset per launch): that hand-over is stream order, so a launch on
    another stream waits for the last one and starts over with zeroed areas.  Six launches alternating between two streams with no
    host synchronisation in between return what one launch returns."""
    import torch
    hs = util.small_street()
    ctx = api.Context(0)
    hs.upload(ctx)
    accel = ctx.accel_build()
    org, dirs = util.pinhole_rays(320, 200, (2.0, 5.0, 26.0), (0.0, 3.0, 0.0), fov_y_deg=60.0)
    want = _gpu_trace(ctx, accel, api.TRACE_CLOSEST, org, dirs)
    n = len(org)
    d_org, d_dir = torch.from_numpy(org).cuda(), torch.from_numpy(dirs).cuda()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = [torch.zeros(n * 4, dtype=torch.int32, device="cuda") for _ in range(6)]
    torch.cuda.synchronize()
    for k, out in enumerate(outs):
        ctx.trace(accel, api.TRACE_CLOSEST, d_org.data_ptr(), d_dir.data_ptr(), n, out.data_ptr(), 0, stream=streams[k % 2].cuda_stream)
    torch.cuda.synchronize()
    for k, out in enumerate(outs):
        got = out.cpu().numpy().view(api.HIT_DTYPE).reshape(n)
        assert np.array_equal(got.view(np.uint8), want.view(np.uint8)), f"launch {k}"


def test_degenerate_inputs(built_lib):
    ctx = api.Context(0)
    # empty scene: every ray misses
    accel = ctx.accel_build()
    org, dirs = util.pinhole_rays(8, 8, (0, 0, 5), (0, 0, 0))
    hits = _gpu_trace(ctx, accel, api.TRACE_CLOSEST, org, dirs)
    assert np.all(hits["triIndex"] == api.GFX_INVALID_SLOT)
    assert np.all(_gpu_trace(ctx, accel, api.TRACE_ANY, org, dirs) == 0)
    # one triangle
    hs = api.HostScene()
    mat = hs.add_material_traditional((0.5, 0.5, 0.5), (0, 0, 0), 0.1)
    v = np.zeros(3, api.VERTEX_DTYPE)
    v["position"] = [(-1, -1, 0), (1, -1, 0), (0, 1, 0)]
    v["normal"] = (0, 0, 1); v["texCoord0Dir"] = (1, 0, 0)
    hs.add_instance(hs.add_group([hs.add_geom(v, [(0, 1, 2)], mat)]), api.make_transform())
    ctx2 = api.Context(0)
    hs.upload(ctx2)
    accel2 = ctx2.accel_build()
    assert ctx2.accel_stats(accel2)["nodes"] == 1
    org, dirs = util.pinhole_rays(32, 32, (0, 0, 4), (0, 0, 0))
    osc = util.feed_oracle(hs, brute_force=True)
    _compare_closest(_gpu_trace(ctx2, accel2, api.TRACE_CLOSEST, org, dirs), _tri_ids(ctx2, accel2),
                     osc.trace(2, org, dirs), osc.tri_ids(), "single triangle")


def test_split_tree_static_plus_animated_subtree(built_lib):
    """A declared-animated rectangle light (2 triangles: single-kernel subtree) and a declared-animated teapot
    (general builder) next to a static street: after every transform update the in-place rebuild touches only the
    animated subtree, and closest / any hits still equal the oracle's (rebuilt from scratch each time)."""
    hs = util.small_street()
    light = hs.add_rectangle(2.0, 1.0, (30, 30, 30))
    light_slot = hs.add_instance(light, api.make_transform(pos=(0.0, 6.0, 0.0)))
    pot = hs.load_obj(__import__("os").path.join(util.ASSETS, "teapot.obj"))
    pot_slot = hs.add_instance(pot, api.make_transform(scale=0.4, pos=(3.0, 0.0, 2.0)))
    ctx = api.Context(0)
    hs.upload(ctx)
    ctx.instance_set_dynamic(light_slot)
    accel = ctx.accel_build()
    nodes_before = ctx.accel_stats(accel)["nodes"]
    osc = util.feed_oracle(hs)
    b = hs.bounds()
    centre = 0.5 * (b[:3] + b[3:])
    org, dirs = util.pinhole_rays(192, 128, centre + np.array([0.1, 0.25, 0.7]) * np.linalg.norm(b[3:] - b[:3]), centre)
    shadow_o = org.copy(); shadow_d = dirs.copy()
    shadow_d[:, 3] = 40.0                                   # finite tmax: any-hit within 40 units
    for step in range(4):
        t = step / 3.0
        moves = [(light_slot, api.make_transform(pos=(-4.0 + 8.0 * t, 6.0 - 2.0 * t, 1.0), roll=35.0 * t))]
        if step >= 1:                                        # the teapot joins the animated set at step 1 (one full rebuild)
            moves.append((pot_slot, api.make_transform(scale=0.4, yaw=90.0 * t, pos=(3.0 - 2.0 * t, 0.5 * t, 2.0))))
        for slot, xfm in moves:
            ctx.instance_set_transform(slot, xfm)
            osc.set_instance_transform(slot, xfm)
        assert ctx.accel_build(handle=accel) == accel
        osc.commit()
        gpu = _gpu_trace(ctx, accel, api.TRACE_CLOSEST, org, dirs)
        _compare_closest(gpu, _tri_ids(ctx, accel), osc.trace(0, org, dirs), osc.tri_ids(), f"step {step}")
        occ = _gpu_trace(ctx, accel, api.TRACE_ANY, shadow_o, shadow_d)
        assert np.array_equal(occ != 0, osc.trace(1, shadow_o, shadow_d) != 0), f"step {step}: any-hit"
    assert ctx.accel_stats(accel)["triRecords"] == hs.counts()["triangles"]
    assert nodes_before > 0


def test_cluttered_street_and_per_ray_item_counts(built_lib):
    """The depth-complexity variant of the stand-in (leaf cards, cables, railings: thin and tiny triangles whose boxes a ray
    grazes) against the oracle, and gfx_trace_counted's per-ray item counts: they add up to the launch counters, every ray
    that entered the tree fetched at least the root, and the hits are the same as the plain launch's."""
    import torch
    from gfxexp_amd import scenes
    hs = scenes.small_street(cluttered=True)
    plain = scenes.small_street()
    assert hs.counts()["triangles"] > plain.counts()["triangles"] + 5000
    ctx = api.Context(0)
    hs.upload(ctx)
    accel = ctx.accel_build()
    osc = util.feed_oracle(hs)
    org, dirs = util.pinhole_rays(320, 200, (2.0, 3.0, 28.0), (0.0, 3.0, 0.0), fov_y_deg=60.0)
    n = len(org)
    gpu_hits = _gpu_trace(ctx, accel, api.TRACE_CLOSEST, org, dirs)
    _compare_closest(gpu_hits, _tri_ids(ctx, accel), osc.trace(0, org, dirs), osc.tri_ids(), "cluttered street")
    d_org, d_dir = torch.from_numpy(org).cuda(), torch.from_numpy(dirs).cuda()
    d_out = torch.zeros(n * 4, dtype=torch.int32, device="cuda")
    d_cnt = torch.zeros(4, dtype=torch.int64, device="cuda")
    d_items = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    ctx.trace(accel, api.TRACE_CLOSEST, d_org.data_ptr(), d_dir.data_ptr(), n, d_out.data_ptr(), d_cnt.data_ptr(), d_per_ray_items=d_items.data_ptr())
    torch.cuda.synchronize()
    items, cnt = d_items.cpu().numpy(), d_cnt.cpu().numpy()
    assert np.array_equal(d_out.cpu().numpy().view(api.HIT_DTYPE).reshape(n), gpu_hits)
    assert items.min() >= 1 and items.sum() == cnt[0] + cnt[1] and cnt[2] == n
    assert items.max() > 4 * np.median(items)          # the tail the band split runs into (profiles/r02_band_notes.txt)
    with pytest.raises(api.GfxError):
        ctx.trace(accel, api.TRACE_CLOSEST, d_org.data_ptr(), d_dir.data_ptr(), n, d_out.data_ptr(), 0, d_per_ray_items=d_items.data_ptr())


@pytest.mark.parametrize("num_bytes", [16, 16 * 255, 16 * 1024 * 4 + 16 * 3, (1 << 24) + 16 * 77])
def test_stream_copy_moves_every_byte(built_lib, num_bytes):
    """gfx_stream_copy (the measurement utility behind roofline.peak_measured) is a copy: every 16-byte word arrives, nothing
    outside the range is written; the read-only form writes nothing; sizes that are not multiples of 16 are refused."""
    import torch
    ctx = api.Context(0)
    rng = np.random.default_rng(num_bytes)
    src = torch.from_numpy(rng.integers(0, 256, num_bytes + 64, dtype=np.uint8)).cuda()
    dst = torch.full((num_bytes + 64,), 0xA5, dtype=torch.uint8, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    ctx.stream_copy(dst.data_ptr() + 32, src.data_ptr() + 16, num_bytes, stream=s)
    ctx.stream_copy(0, src.data_ptr(), num_bytes, stream=s)          # read-only pass
    torch.cuda.synchronize()
    got, want = dst.cpu().numpy(), src.cpu().numpy()
    assert np.array_equal(got[32:32 + num_bytes], want[16:16 + num_bytes])
    assert np.all(got[:32] == 0xA5) and np.all(got[32 + num_bytes:] == 0xA5)
    with pytest.raises(api.GfxError, match="multiples of 16"):
        ctx.stream_copy(dst.data_ptr(), src.data_ptr(), num_bytes - 8, stream=s)
    with pytest.raises(api.GfxError, match="multiples of 16"):
        ctx.stream_copy(dst.data_ptr() + 4, src.data_ptr(), 16, stream=s)
    ctx.close()
