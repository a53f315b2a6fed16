"""CPU, world_size 2 / 3 / 4, gloo: the row-band split reproduces the single-process frame BIT FOR BIT.

World 3 and 4 run on frames whose bands are UNEQUAL (56 rows / 3 ranks = 24 + 16 + 16, 72 rows / 4 = 24 + 16 + 16 + 16; 1080
rows / 8 ranks are 7 x 136 + 128): interior ranks exchange strips with two neighbours, and the all-gathers pad every slab
to the tallest band.  The smallest band (16 rows) still holds the tallest strip of the cases (radius 5 + 8 motion rows).

Two generations of the split are covered:
  * strip exchange (gfxh_restir_set_exchange): every pass runs on the band only and the rows the next pass reads
    across the seams are exchanged between passes.  The ranks execute gfxh_restir_frame_program -- the list of steps the
    GPU driver executes -- with the CPU oracle as the kernels (tests/bandprog.py) and the production communication code
    (gfxexp_amd/tilesplit.StripExchange) on the gloo backend: original ReSTIR biased / unbiased (static and moving
    camera), rearchitected ReSTIR, the ReGIR path tracer (all-reduce of the cell access counters), baseline path tracer.
  * halo recompute (gfxh_band_plan + HaloExchange + BandGather), the round-1 scheme that a band renderer WITHOUT an
    exchange callback still uses."""
import os
import socket
import sys
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

W, H, FRAMES, RADIUS, PASSES, NB = 64, 48, 3, 5.0, 2, 3


def _sequence(osc, pb, plan, cam, frames, exchange=None, gather=None):
    from gfxexp_amd import api
    from oracle import oracle as O
    from tests import util
    s = pb.host_static_params()
    ocam = util.copy_struct(O.GfxCamera, cam)
    last_res, last_base = 1, 0
    for frame in range(frames):
        kw = dict(frameIndex=frame, bufferIndex=frame % 2, resetFlowBuffer=int(frame == 0), numAccumFrames=0,
                  numSpatialNeighbors=NB, spatialNeighborRadius=RADIUS)
        f = util.frame_params(O.GfxRestirFrameParams, O.GfxCamera, W, H, ocam, **kw)
        cur = (last_res + 1) % 2

        def rows(r):
            return (0, int(r[0]), W, int(r[1]))
        osc.restir_launch(s, f, cur, last_base, api.PASS_SETUP_GBUFFERS, rect=rows(plan.gbufferRows))
        osc.restir_launch(s, f, cur, last_base, api.PASS_INITIAL_RIS if frame == 0 else api.PASS_INITIAL_TEMPORAL_BIASED,
                          rect=rows(plan.initialRows))
        for i in range(PASSES):
            osc.restir_launch(s, f, cur, last_base + NB * i, api.PASS_SPATIAL_BIASED, rect=rows(plan.spatialRows[i]))
            cur = (cur + 1) % 2
        last_base += NB * PASSES
        osc.restir_launch(s, f, cur, last_base, api.PASS_SHADING, rect=rows(plan.shadingRows))
        last_res = cur
        if exchange is not None:
            exchange.exchange(last_res)
        if gather is not None:
            gather.all_gather()
    return last_res


def _worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist
    from gfxexp_amd import api, tilesplit
    from tests import util
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    hs = util.bunny_scene()
    osc = util.feed_oracle(hs, threads=2)
    pb = util.PixelBuffers(W, H)
    b, e = tilesplit.band_for_rank(H, world, rank)
    plan = api.band_plan(H, b, e, int(np.ceil(RADIUS)), PASSES)
    assert e - b >= plan.haloRows
    state = {"rng": torch.from_numpy(pb.rng.view(np.int64)), "beauty": torch.from_numpy(pb.beauty.reshape(-1))}
    for i in range(2):
        state[f"info{i}"] = torch.from_numpy(pb.info[i].reshape(-1))
        state[f"res{i}"] = torch.from_numpy(pb.res[i].reshape(-1))
    exchange = tilesplit.HaloExchange(state, plan, W, H, rank, world, dist)
    gather = tilesplit.BandGather(state["beauty"], W, H, world, rank, dist)
    cam = api.make_camera(W, H, pos=(1.5, 5.0, 14.0), pitch=12.0, yaw=186.0)
    last = _sequence(osc, pb, plan, cam, FRAMES, exchange, gather)
    gather.finish()
    np.save(os.path.join(out_dir, f"beauty_{rank}.npy"), pb.beauty)
    np.save(os.path.join(out_dir, f"res_{rank}.npy"), pb.res[last])
    np.save(os.path.join(out_dir, f"band_{rank}.npy"), np.array([b, e]))
    dist.barrier()
    dist.destroy_process_group()


def test_band_rows_are_tile_aligned_and_cover_the_frame():
    from gfxexp_amd import tilesplit
    for h in (1080, 48, 720, 1081):
        for world in (1, 2, 4, 8):
            bands = tilesplit.band_rows(h, world)
            assert bands[0][0] == 0 and bands[-1][1] == h
            for (b0, e0), (b1, e1) in zip(bands, bands[1:]):
                assert e0 == b1 and b0 % 8 == 0 and b1 % 8 == 0
    assert tilesplit.band_rows(1080, 8)[0] == (0, 136) and tilesplit.band_rows(1080, 8)[-1] == (952, 1080)


def test_band_plan_ranges(built_lib):
    from gfxexp_amd import api
    p = api.band_plan(1080, 136, 272, 20, 2)
    assert p.haloRows == 40
    assert list(p.gbufferRows) == [96, 312] and list(p.initialRows) == [96, 312]
    assert list(p.spatialRows[0]) == [116, 292] and list(p.spatialRows[1]) == [136, 272]
    assert list(p.shadingRows) == [136, 272]
    assert list(p.recvAbove) == [96, 136] and list(p.sendAbove) == [136, 176]
    assert list(p.recvBelow) == [272, 312] and list(p.sendBelow) == [232, 272]
    top = api.band_plan(1080, 0, 136, 20, 2)
    assert list(top.gbufferRows) == [0, 176] and list(top.recvAbove) == [0, 0] and top.sendAbove[0] == top.sendAbove[1]
    bottom = api.band_plan(1080, 952, 1080, 20, 1)
    assert list(bottom.gbufferRows) == [932, 1080] and bottom.recvBelow[0] == bottom.recvBelow[1] == 1080


def test_two_rank_band_split_is_bit_exact(built_lib):
    import torch.multiprocessing as mp
    from gfxexp_amd import api
    from tests import util
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    with tempfile.TemporaryDirectory() as out_dir:
        mp.spawn(_worker, args=(2, port, out_dir), nprocs=2, join=True)
        gathered = [np.load(os.path.join(out_dir, f"beauty_{r}.npy")) for r in range(2)]
        res = [np.load(os.path.join(out_dir, f"res_{r}.npy")) for r in range(2)]
        bands = [np.load(os.path.join(out_dir, f"band_{r}.npy")) for r in range(2)]
    # single process, whole frame
    hs = util.bunny_scene()
    osc = util.feed_oracle(hs, threads=4)
    pb = util.PixelBuffers(W, H)
    plan = api.band_plan(H, 0, H, int(np.ceil(RADIUS)), PASSES)
    cam = api.make_camera(W, H, pos=(1.5, 5.0, 14.0), pitch=12.0, yaw=186.0)
    last = _sequence(osc, pb, plan, cam, FRAMES)
    assert np.abs(pb.beauty[:, :3]).sum() > 0
    for r in range(2):
        util.assert_same_bits(f"rank {r} gathered HDR frame", gathered[r], pb.beauty)
        b, e = bands[r]
        want = pb.res[last].reshape(3, H, W, 4)[:, b:e]
        util.assert_same_bits(f"rank {r} final reservoirs of its band", res[r].reshape(3, H, W, 4)[:, b:e], want)


# ---------------------------------------------------------------------------------------------------------------------
# strip exchange: the driver's frame program, run with the oracle
# ---------------------------------------------------------------------------------------------------------------------
MOTION_ROWS = 8
#            name              renderer attr                frames  moving  regir
STRIP_CASES = [("biased", "RENDERER_BIASED", 3, False, False),
               ("unbiased", "RENDERER_UNBIASED", 3, False, False),
               ("biased_moving", "RENDERER_BIASED", 3, True, False),
               ("unbiased_moving", "RENDERER_UNBIASED", 3, True, False),
               ("rearch_biased", "RENDERER_REARCH_BIASED", 3, False, False),
               ("rearch_unbiased_moving", "RENDERER_REARCH_UNBIASED", 3, True, False),
               ("regir", "RENDERER_PATH_TRACE_REGIR", 3, False, True),
               ("path_trace", "RENDERER_PATH_TRACE", 2, False, False)]


# frame heights by world size: equal bands for 2 ranks, unequal ones (24 rows, then 16s) for 3 and 4
HEIGHTS = {2: 48, 3: 56, 4: 72}


def _case_camera(frame, moving, height=H):
    from gfxexp_amd import api
    dy = 1.0 * frame if moving else 0.0
    return api.make_camera(W, height, pos=(1.5 + 0.5 * dy, 5.0 + dy, 14.0), pitch=12.0, yaw=186.0)


def _strip_mode(case):
    """gfxh_restir_frame_program's stripMode per case: 3 (the intermediate spatial pass recomputed on its halo, ONE reservoir exchange per frame --
    what the driver runs) for most, 1 (an exchange before every spatial pass) for "unbiased", so that both stay under the oracle."""
    return 1 if case[0] == "unbiased" else 3


def _run_case(case, band, exchange, threads, height=H):
    """One sequence of a STRIP_CASES entry on `band` ((0, 0) = whole frame); returns the renderer (state in .pb / .regir)."""
    from gfxexp_amd import api
    from tests import bandprog, util
    name, renderer, frames, moving, with_regir = case
    hs = util.bunny_scene()
    osc = util.feed_oracle(hs, threads=threads)
    cfg = bandprog.small_config(W, height, getattr(api, renderer), band=band, radius=RADIUS, passes=PASSES, neighbors=NB)
    cfg.maxPathLength = 3
    regir = util.RegirBuffers(hs.bounds(), dims=(8, 4, 8)) if with_regir else None
    r = bandprog.OracleBandRenderer(osc, cfg, regir=regir)
    r.strip_mode = _strip_mode(case)
    if exchange is not None:
        r.set_exchange(exchange, MOTION_ROWS if moving else 0)
    for frame in range(frames):
        r.render_frame(_case_camera(frame, moving, height))
    return r


def _state(r):
    out = {"beauty": r.pb.beauty, "res": r.pb.res[r.last_res], "info": r.pb.info[r.last_res], "rng": r.pb.rng,
           "motion": r.pb.gb1[(r.frame_index - 1) % 2], "surface": r.pb.gb0[(r.frame_index - 1) % 2]["instSlot"] != 0xFFFFFFFF}
    if r.regir is not None:
        out.update(regir_accesses=r.regir.accesses, regir_last_access=r.regir.last_access, regir_rngs=r.regir.rngs,
                   regir_res0=r.regir.res[0], regir_res1=r.regir.res[1])
    return out


def _strip_worker(rank, world, port, out_dir, case_names, custom_bands=None, host_staged=False):
    import torch.distributed as dist
    from gfxexp_amd import tilesplit
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    if host_staged:        # the adapter bench.py's GFX_BENCH_ONE_GPU mode puts between StripExchange and gloo
        dist = tilesplit.HostStaged(dist)
    height = HEIGHTS[world]
    band = custom_bands[rank] if custom_bands else tilesplit.band_for_rank(height, world, rank)
    for case in STRIP_CASES:
        if case[0] not in case_names:
            continue
        # the asynchronous gather (bench.py's mode) on the odd cases, the synchronous default on the others
        ex = tilesplit.StripExchange(dist, rank, world, height, tilesplit.host_view, async_gather=(STRIP_CASES.index(case) % 2 == 1),
                                     bands=custom_bands)
        r = _run_case(case, band, ex, threads=2, height=height)
        ex.finish()
        np.savez(os.path.join(out_dir, f"{case[0]}_{rank}.npz"), band=np.array(band), bytes_moved=ex.bytes_moved,
                 log=np.array([(f, op, rows, bufs) for f, op, rows, bufs in r.log], np.int64).reshape(-1, 4), **_state(r))
    dist.barrier()
    dist.destroy_process_group()


def _spawn_strip_runs(world, case_names, custom_bands=None, host_staged=False):
    import torch.multiprocessing as mp
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    with tempfile.TemporaryDirectory() as out_dir:
        mp.spawn(_strip_worker, args=(world, port, out_dir, case_names, custom_bands, host_staged), nprocs=world, join=True)
        return {(c, r): dict(np.load(os.path.join(out_dir, f"{c}_{r}.npz"))) for c in case_names for r in range(world)}


@pytest.fixture(scope="module")
def strip_runs(built_lib):
    return _spawn_strip_runs(2, [c[0] for c in STRIP_CASES])


# three ranks: every case; four ranks: the cases whose exchanges differ in kind (strips with motion, the rearchitected
# once-per-frame strip, the counter all-reduce)
WORLD3_CASES = [c[0] for c in STRIP_CASES]
WORLD4_CASES = ["biased_moving", "rearch_unbiased_moving", "regir"]


@pytest.fixture(scope="module")
def strip_runs_world3(built_lib):
    return _spawn_strip_runs(3, WORLD3_CASES)


@pytest.fixture(scope="module")
def strip_runs_world4(built_lib):
    return _spawn_strip_runs(4, WORLD4_CASES)


@pytest.mark.parametrize("case", STRIP_CASES, ids=[c[0] for c in STRIP_CASES])
def test_two_rank_strip_exchange_is_bit_exact(strip_runs, case):
    _check_strip_runs(strip_runs, case, 2)


@pytest.mark.parametrize("case", STRIP_CASES, ids=[c[0] for c in STRIP_CASES])
def test_three_rank_strip_exchange_with_unequal_bands_is_bit_exact(strip_runs_world3, case):
    """Rank 1 has two neighbours; the bands are 24 + 16 + 16 rows."""
    _check_strip_runs(strip_runs_world3, case, 3)


# a partition that is not the default one (what gfxh_balance_bands hands bench.py): the short band first
CUSTOM_BANDS_WORLD3 = [(0, 16), (16, 40), (40, 56)]
CUSTOM_CASES = ["biased_moving", "rearch_unbiased_moving"]


@pytest.fixture(scope="module")
def strip_runs_host_staged(built_lib):
    return _spawn_strip_runs(3, WORLD4_CASES, host_staged=True)


@pytest.mark.parametrize("case", [c for c in STRIP_CASES if c[0] in WORLD4_CASES], ids=WORLD4_CASES)
def test_three_rank_strip_exchange_through_the_host_staged_adapter(strip_runs_host_staged, case):
    """tilesplit.HostStaged (the transport of `GFX_BENCH_ONE_GPU=1 bench.py --gpus N`, N ranks on one device) between StripExchange
    and gloo: the same frames as the direct transport -- strips, the counter all-reduce, synchronous and asynchronous band gathers."""
    _check_strip_runs(strip_runs_host_staged, case, 3)


@pytest.fixture(scope="module")
def strip_runs_custom(built_lib):
    return _spawn_strip_runs(3, CUSTOM_CASES, CUSTOM_BANDS_WORLD3)


@pytest.mark.parametrize("case", [c for c in STRIP_CASES if c[0] in CUSTOM_CASES], ids=CUSTOM_CASES)
def test_three_rank_strip_exchange_with_an_explicit_partition_is_bit_exact(strip_runs_custom, case):
    """Cost-balanced bands: StripExchange(bands=...) and band renderers cut at 16 | 24 | 16 rows instead of 24 | 16 | 16."""
    from gfxexp_amd import tilesplit
    assert CUSTOM_BANDS_WORLD3 != tilesplit.band_rows(HEIGHTS[3], 3)
    _check_strip_runs(strip_runs_custom, case, 3, CUSTOM_BANDS_WORLD3)


def test_balance_bands_and_check_bands(built_lib):
    """gfxh_balance_bands / gfxh_restir_check_bands (include/gfxexp_host.h): tile-aligned cuts that cover the frame, honour the
    minimum band height, equalise the predicted band times -- and do nothing to a partition whose bands already take equally long."""
    from gfxexp_amd import api, tilesplit
    H, world = 1080, 8
    equal = tilesplit.band_rows(H, world)
    ms = [0.6072, 0.7118, 0.7705, 0.7858, 0.7631, 0.7489, 0.7401, 0.5912]      # tools/bench_band.py, eight bands of the bench frame
    cut = api.balance_bands(H, equal, ms, min_rows=24)
    assert cut[0][0] == 0 and cut[-1][1] == H and all(a[1] == b[0] for a, b in zip(cut, cut[1:]))
    assert all(b % 8 == 0 for b, _ in cut) and all(e - b >= 24 for b, e in cut)
    cost = np.zeros(H)
    for (b, e), t in zip(equal, ms):
        cost[b:e] = t / (e - b)
    predicted = [cost[b:e].sum() for b, e in cut]
    assert max(predicted) < 0.95 * max(ms) and cut[0][1] > equal[0][1] and cut[-1][0] < equal[-1][0]   # the sky and ground bands grow
    same = api.balance_bands(H, equal, [1.0 * (e - b) for b, e in equal], min_rows=24)    # uniform cost: 135 rows each, to the nearest tile
    assert all(abs((e - b) - H / world) <= 8 for b, e in same) and all(abs(b - k * H / world) <= 4 for k, (b, _) in enumerate(same))
    with pytest.raises(api.GfxError):
        api.balance_bands(H, equal, ms, min_rows=200)                       # eight bands of 200 rows do not fit 1080
    # a lopsided profile cannot squeeze a band below the minimum
    tight = api.balance_bands(H, equal, [10.0] + [0.1] * 7, min_rows=64)
    assert all(e - b >= 64 for b, e in tight) and tight[0][1] - tight[0][0] == 64
    # a height off the tile grid (1084 = 135 tiles + 4 rows): the last band ends in the partial tile and still gets its minimum in ROWS
    ragged = api.balance_bands(1084, tilesplit.band_rows(1084, world), [0.1] * 7 + [10.0], min_rows=24)
    assert ragged[-1][1] == 1084 and all(e - b >= 24 for b, e in ragged) and all(b % 8 == 0 for b, _ in ragged), ragged
    ragged = api.balance_bands(1084, tilesplit.band_rows(1084, world), [10.0] + [0.1] * 7, min_rows=24)
    assert all(e - b >= 24 for b, e in ragged), ragged
    with pytest.raises(api.GfxError):
        api.balance_bands(H, [(0, 100)] + [(100 + 140 * k, 240 + 140 * k) for k in range(6)] + [(940, H)], ms, min_rows=24)   # a boundary off the tile grid
    cfg = api.RestirRenderer.default_config(1920, H, api.RENDERER_BIASED)
    api.check_bands(cfg, cut)                                              # radius-20 strips fit every band of `cut`
    with pytest.raises(api.GfxError):
        api.check_bands(cfg, [(0, 8), (8, H)])                             # a 20-row strip out of an 8-row band
    with pytest.raises(api.GfxError):
        api.check_bands(cfg, [(0, 100), (100, H)])                         # not on a tile boundary


@pytest.mark.parametrize("case", [c for c in STRIP_CASES if c[0] in WORLD4_CASES], ids=WORLD4_CASES)
def test_four_rank_strip_exchange_with_unequal_bands_is_bit_exact(strip_runs_world4, case):
    """Two interior ranks; the bands are 24 + 16 + 16 + 16 rows."""
    _check_strip_runs(strip_runs_world4, case, 4)


def _check_strip_runs(strip_runs, case, world, custom_bands=None):
    from gfxexp_amd import api, tilesplit
    from tests import util
    H = HEIGHTS[world]
    bands = custom_bands if custom_bands else tilesplit.band_rows(H, world)
    if world > 2:
        assert len(set(e - b for b, e in bands)) > 1, "the bands of this test are meant to be unequal"
    whole = _run_case(case, (0, 0), None, threads=4, height=H)
    want = _state(whole)
    assert np.abs(want["beauty"][:, :3]).sum() > 0
    name, _, frames, moving, with_regir = case
    if moving:   # the sequence really crosses the seam, and stays inside the strip the ranks exchange
        my = np.abs(want["motion"][want["surface"], 1])
        assert 3.0 < my.max() <= MOTION_ROWS - 1
    for rank in range(world):
        got = strip_runs[(name, rank)]
        b, e = got["band"]
        assert (int(b), int(e)) == bands[rank]
        util.assert_same_bits(f"{name}: rank {rank} gathered HDR frame", got["beauty"], want["beauty"])
        rows = lambda a, per: np.ascontiguousarray(a).reshape(-1, H, W, per)[:, b:e]
        util.assert_same_bits(f"{name}: rank {rank} final reservoirs", rows(got["res"], 4), rows(want["res"], 4))
        util.assert_same_bits(f"{name}: rank {rank} final reservoir infos", rows(got["info"], 2), rows(want["info"], 2))
        util.assert_same_bits(f"{name}: rank {rank} pixel RNGs", got["rng"].reshape(H, W)[b:e], want["rng"].reshape(H, W)[b:e])
        if with_regir:   # the world-space grid is replicated: identical on every rank, and to the single process
            for k in ("regir_accesses", "regir_last_access", "regir_rngs", "regir_res0", "regir_res1"):
                util.assert_same_bits(f"{name}: rank {rank} {k}", got[k], want[k])
            assert want["regir_accesses"].sum() > 0
        # what moved: per frame exactly the exchange points gfxexp_host.h documents
        ops = [int(op) for f, op, _, _ in got["log"] if f == frames - 1]
        spatial_exchanges = PASSES if _strip_mode(case) == 1 else 1       # mode 3: one exchange of radius x passes rows (+ RNG states) behind the candidate pass
        if name in ("biased", "unbiased"):
            assert ops == [api.STEP_EXCHANGE_STRIPS] * (1 + spatial_exchanges) + [api.STEP_GATHER_BANDS]
            tall = [int(rows) for f, op, rows, _ in got["log"] if f == frames - 1 and op == api.STEP_EXCHANGE_STRIPS]
            assert tall == ([int(np.ceil(RADIUS))] * (1 + PASSES) if _strip_mode(case) == 1 else [int(np.ceil(RADIUS)) * PASSES] * 2)
        elif name.endswith("_moving") and not name.startswith("rearch"):
            assert ops == [api.STEP_EXCHANGE_STRIPS] * (2 + spatial_exchanges) + [api.STEP_GATHER_BANDS]
        elif name.startswith("rearch"):
            assert ops == [api.STEP_EXCHANGE_STRIPS, api.STEP_GATHER_BANDS]
        elif name == "regir":
            assert ops == [api.STEP_ALLREDUCE_CELL_ACCESSES, api.STEP_GATHER_BANDS]
        else:
            assert ops == [api.STEP_GATHER_BANDS]


def test_strip_rows_and_too_tall_strips(built_lib):
    from gfxexp_amd import api
    d, rc = api.strip_rows(1080, 136, 272, 20)
    assert rc == 0
    assert list(d.sendAbove) == [136, 156] and list(d.recvAbove) == [116, 136]
    assert list(d.sendBelow) == [252, 272] and list(d.recvBelow) == [272, 292]
    top, rc = api.strip_rows(1080, 0, 136, 20)
    assert rc == 0 and top.sendAbove[0] == top.sendAbove[1] and top.recvAbove[0] == top.recvAbove[1]
    bottom, rc = api.strip_rows(1080, 952, 1080, 20)
    assert rc == 0 and bottom.sendBelow[0] == bottom.sendBelow[1] and bottom.recvBelow[0] == bottom.recvBelow[1]
    _, rc = api.strip_rows(1080, 136, 272, 137)           # would need rows of a rank two bands away
    assert rc == 1
    cfg = api.RestirRenderer.default_config(1920, 1080, api.RENDERER_BIASED)
    cfg.rowBegin, cfg.rowEnd = 136, 272
    with pytest.raises(api.GfxError):
        api.frame_program(cfg, True, 200, False, 1, 0, False)


def test_partition_check_is_the_same_verdict_on_every_rank(built_lib):
    """gfxh_restir_check_partition compares the tallest strip with the SMALLEST band, so all ranks agree.  The per-frame test
    of the frame program only sees the calling rank's band: 1080 rows over 8 ranks are 7 x 136 + 128, and a 130-row strip
    passes there on ranks 0-6 and fails on rank 7 -- whose neighbours would already be waiting in the exchange."""
    from gfxexp_amd import api, tilesplit
    for h, world in ((1080, 8), (48, 2), (56, 3), (1081, 5)):
        assert [api.band_rows(h, world, r) for r in range(world)] == tilesplit.band_rows(h, world)
    cfg = api.RestirRenderer.default_config(1920, 1080, api.RENDERER_BIASED)
    api.check_partition(cfg, 8, 0)                      # radius 20 against 128 rows
    api.check_partition(cfg, 8, 128)
    with pytest.raises(api.GfxError, match="130 rows.*smallest band .128"):
        api.check_partition(cfg, 8, 130)
    verdicts = []
    for rank in range(8):                               # what each rank would have decided on its own
        cfg.rowBegin, cfg.rowEnd = api.band_rows(1080, 8, rank)
        try:
            api.frame_program(cfg, True, 130, False, 1, 0, False)
            verdicts.append(True)
        except api.GfxError:
            verdicts.append(False)
    assert verdicts == [True] * 7 + [False]
    # the rearchitected renderer moves radius + motion rows at once; 40 rows over 3 ranks leave an 8-row band
    small = api.RestirRenderer.default_config(64, 40, api.RENDERER_REARCH_UNBIASED)
    small.spatialNeighborRadius = 5.0
    api.check_partition(small, 3, 3)
    with pytest.raises(api.GfxError):
        api.check_partition(small, 3, 8)
    with pytest.raises(api.GfxError):
        api.check_partition(small, 6, 0)                # more ranks than 8-row tiles


# ---------------------------------------------------------------------------------------------------------------------
# NRC band renderers: records gathered in rank order, every rank trains its own copy (or: rank 0 trains, inference parameters broadcast)
# ---------------------------------------------------------------------------------------------------------------------
NRC_FRAMES = 2


def _nrc_run(band, rank, exchange, threads, H=H, train_on_rank0=False):
    from gfxexp_amd import api
    from tests import bandprog, util
    hs = util.bunny_scene()
    osc = util.feed_oracle(hs, threads=threads)
    r = bandprog.OracleNrcBandRenderer(osc, hs, W, H, band=band, rank=rank, exchange=exchange, train_on_rank0=train_on_rank0)
    cam = api.make_camera(W, H, pos=(1.5, 5.0, 14.0), pitch=12.0, yaw=186.0)
    for _ in range(NRC_FRAMES):
        r.render_frame(cam)
    b = (NRC_FRAMES - 1) % 2
    n = int(r.nb.a[f"nrc_num_{b}"][0])
    return {"beauty": r.pb.beauty, "ema": r.net.ema, "num": np.array([n]), "tile": r.nb.a[f"nrc_tile_{b}"],
            "trainq": r.nb.a["nrc_trainq_0"][:n], "traint": r.nb.a["nrc_traint_0"][:n],
            "batchq": r.nb.a["nrc_trainq_1"][:1 << 16], "batcht": r.nb.a["nrc_traint_1"][:1 << 16], "rng": r.pb.rng}


def _nrc_worker(rank, world, port, out_dir, train_on_rank0):
    import torch.distributed as dist
    from gfxexp_amd import tilesplit
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    H = HEIGHTS[world]
    band = tilesplit.band_for_rank(H, world, rank)
    ex = tilesplit.StripExchange(dist, rank, world, H, tilesplit.host_view)
    out = _nrc_run(band, rank, ex, threads=2, H=H, train_on_rank0=train_on_rank0)
    ex.finish()
    np.savez(os.path.join(out_dir, f"nrc_{rank}.npz"), band=np.array(band), **out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,train_on_rank0", [(2, False), (3, False), (2, True)])
def test_nrc_band_split_is_bit_exact(built_lib, world, train_on_rank0):
    """The CPU oracle allocates training records in pixel order, so the bands' records concatenated in rank order ARE the
    single-process records: the gathered batch, the trained parameters and both frames are bit-identical -- with every rank training
    its own copy of the network on the gathered batch (the default) and with rank 0 training and broadcasting (train_on_rank0).  Three
    ranks: unequal bands (24 + 16 + 16 rows), so the record gather pads to the largest count of three different ones."""
    import torch.multiprocessing as mp
    from tests import util
    H = HEIGHTS[world]
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    with tempfile.TemporaryDirectory() as out_dir:
        mp.spawn(_nrc_worker, args=(world, port, out_dir, train_on_rank0), nprocs=world, join=True)
        got = [dict(np.load(os.path.join(out_dir, f"nrc_{r}.npz"))) for r in range(world)]
    want = _nrc_run((0, 0), 0, None, threads=4, H=H)
    assert want["num"][0] > 100 and np.abs(want["beauty"][:, :3]).sum() > 0
    assert not np.array_equal(want["ema"], __import__("oracle.nrc_net", fromlist=["x"]).NrcNet(0, 2, 1e-2).ema)   # it trained
    for rank in range(world):
        g = got[rank]
        b, e = g["band"]
        for k in ("num", "tile", "trainq", "traint", "batchq", "batcht", "ema"):
            util.assert_same_bits(f"nrc rank {rank} {k}", g[k], want[k])
        util.assert_same_bits(f"nrc rank {rank} gathered HDR frame", g["beauty"], want["beauty"])
        util.assert_same_bits(f"nrc rank {rank} pixel RNGs", g["rng"].reshape(H, W)[b:e], want["rng"].reshape(H, W)[b:e])
