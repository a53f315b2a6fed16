"""-m gpu: the ReSTIR DI passes (G-buffer, initial + temporal RIS, spatial RIS, shading), pass by
pass, against the CPU oracle on identical scenes, seeds and parameters.

Bar (north_star): reservoir sample selection bit-exact -- here EVERY per-pixel buffer is compared
bit for bit (RNG state, G-buffers, reservoirs, ReservoirInfo, beauty/albedo/normal).
"""
import ctypes as C

import numpy as np
import pytest

from gfxexp_amd import api
from oracle import oracle as O
from tests import util


def _cam_for(hs, width, height):
    b = hs.bounds()
    centre = 0.5 * (b[:3] + b[3:])
    return centre


def default_camera(scene_kind, width, height):
    if scene_kind == "bunny":
        return api.make_camera(width, height, pos=(1.5, 5.0, 14.0), pitch=12.0, yaw=186.0)
    return api.make_camera(width, height, pos=(2.0, 5.0, 26.0), pitch=6.0, yaw=184.0)


def run_sequence_both(hs, width, height, frames=2, renderer=api.RENDERER_BIASED, scene_kind="bunny",
                      low_discrepancy=True, reuse_visibility=True, camera=None, stop_after=None, threads=None,
                      env=None, env_power=1.0, env_rotation=0.0, animate=None, tunables=None, env_tables="interleaved"):
    """Run `frames` frames with the sequencing of restir_di_main.cpp:2311-2493 on the GPU (through
    the C ABI) and in the oracle, comparing all buffers after every pass.  Returns a list of
    mismatch descriptions (empty = bit-identical)."""
    import torch
    ctx = api.Context(0)
    for name, value in (tunables or {}).items():
        ctx.tunable_set(name, value)
    hs.upload(ctx)
    accel = ctx.accel_build()
    ctx.lights_build_static()
    osc = util.feed_oracle(hs, threads=threads)
    cam = camera if camera is not None else default_camera(scene_kind, width, height)
    ocam = util.copy_struct(O.GfxCamera, cam)

    pb_gpu_init = util.PixelBuffers(width, height)
    pb_cpu = util.PixelBuffers(width, height)
    if env is not None:
        pb_gpu_init.set_env(*env)
        pb_cpu.set_env(*env, oracle_side=True)
        # what the GPU searches: the interleaved rows (gfx_restir_static_params::envRowTable), the separate arrays with guide tables, or the plain arrays
        pb_gpu_init.use_env_row_table = env_tables in ("interleaved", "sketch")
        pb_gpu_init.use_env_row_sketch = env_tables == "sketch"
        pb_gpu_init.use_env_guides = env_tables != "plain"
    dev = util.DeviceBuffers(pb_gpu_init)
    s_gpu = dev.static_params()
    s_cpu = pb_cpu.host_static_params()

    unbiased = renderer == api.RENDERER_UNBIASED
    num_passes = 1 if unbiased else 2
    num_nb = 3 if unbiased else 5
    last_res, last_base = 1, 0
    diffs = []
    stream = torch.cuda.current_stream().cuda_stream

    def compare(tag):
        got = dev.download()
        want = pb_cpu.arrays()
        for k in want:
            a = np.ascontiguousarray(got[k]).view(np.uint8).reshape(-1)
            b = np.ascontiguousarray(want[k]).view(np.uint8).reshape(-1)
            if not np.array_equal(a, b):
                item = want[k].dtype.itemsize
                nbad = len(np.unique(np.nonzero(a != b)[0] // item))
                diffs.append(f"{tag}: {k}: {nbad} of {want[k].size} elements differ")

    for frame in range(frames):
        buffer_index = frame % 2
        new_sequence = frame == 0
        if animate is not None:
            # InstanceController::update for the animated instances, then updateASs (restir_di_main.cpp:2258-2264)
            moves = animate(frame)
            for inst_slot, xfm, nm in moves:
                ctx.instance_set_transform(inst_slot, xfm, nm)
                osc.set_instance_transform(inst_slot, xfm, nm)
            if moves:
                assert ctx.accel_build(handle=accel) == accel
                osc.commit()
        kw = dict(frameIndex=frame, bufferIndex=buffer_index, resetFlowBuffer=int(new_sequence), numAccumFrames=0,
                  numSpatialNeighbors=num_nb, useUnbiasedEstimator=int(unbiased),
                  useLowDiscrepancyNeighbors=int(low_discrepancy), reuseVisibility=int(reuse_visibility),
                  enableEnvLight=int(env is not None), envLightPowerCoeff=env_power, envLightRotation=env_rotation)
        f_gpu = util.frame_params(api.GfxRestirFrameParams, api.GfxCamera, width, height, cam, travHandle=accel, **kw)
        f_cpu = util.frame_params(O.GfxRestirFrameParams, O.GfxCamera, width, height, ocam, travHandle=0, **kw)
        ctx.lights_build_instances(stream)
        cur = (last_res + 1) % 2

        def both(pass_id, cur_res, base, tag):
            ctx.restir_set_params(s_gpu, f_gpu, cur_res, base, stream)
            ctx.restir_launch(pass_id, width, height, stream)
            osc.restir_launch(s_cpu, f_cpu, cur_res, base, pass_id)
            compare(f"frame {frame} {tag}")

        both(api.PASS_SETUP_GBUFFERS, cur, last_base, "gbuffer")
        if stop_after == "gbuffer":
            break
        entry = api.PASS_INITIAL_RIS
        if not new_sequence:
            entry = api.PASS_INITIAL_TEMPORAL_UNBIASED if unbiased else api.PASS_INITIAL_TEMPORAL_BIASED
        both(entry, cur, last_base, "initial/temporal")
        for i in range(num_passes):
            both(api.PASS_SPATIAL_UNBIASED if unbiased else api.PASS_SPATIAL_BIASED, cur, last_base + num_nb * i, f"spatial {i}")
            cur = (cur + 1) % 2
        last_base += num_nb * num_passes
        both(api.PASS_SHADING, cur, last_base, "shading")
        last_res = cur
    run_sequence_both.last_beauty = pb_cpu.beauty.copy()
    run_sequence_both.last_gb0 = pb_cpu.gb0[(frames - 1) % 2].copy()
    run_sequence_both.last_motion = np.nan_to_num(np.asarray(pb_cpu.gb1[(frames - 1) % 2], np.float32).copy())
    return diffs


@pytest.mark.gpu
@pytest.mark.parametrize("renderer", [api.RENDERER_BIASED, api.RENDERER_UNBIASED])
def test_bunny_sequence_bit_exact(built_lib, renderer):
    diffs = run_sequence_both(util.bunny_scene(), 160, 96, frames=3, renderer=renderer)
    assert not diffs, "\n".join(diffs)
    beauty = run_sequence_both.last_beauty
    surf = run_sequence_both.last_gb0["instSlot"] != 0xFFFFFFFF
    assert surf.mean() > 0.3
    assert np.isfinite(beauty).all() and beauty[surf, :3].mean() > 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("split,fuse", [(1, 1), (2, 1), (4, 1), (1, 2), (4, 2)])
def test_lanes_per_pixel_of_the_candidate_pass_change_nothing(built_lib, split, fuse):
    """k_initial_candidates with 1, 2 or 4 lanes per pixel (the library picks by launch size; "candidate_split" forces it), as a kernel
    of its own between k_trace launches (fuse_passes 1) or inside k_initial_fused (2): same reservoirs, same RNG streams, same
    G-buffers and beauty -- street scene (many emitters, textured ones among them), two frames."""
    diffs = run_sequence_both(util.small_street(), 192, 108, frames=2, renderer=api.RENDERER_BIASED, scene_kind="street",
                              tunables={"candidate_split": split, "fuse_passes": fuse})
    assert not diffs, "\n".join(diffs)
    ctx = api.Context(0)
    with pytest.raises(api.GfxError, match="candidate_split"):
        ctx.tunable_set("candidate_split", 3)
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("fuse", [1, 2])
@pytest.mark.parametrize("low_discrepancy", [True, False])
def test_unbiased_estimator_with_the_ray_passes_fused_and_not(built_lib, fuse, low_discrepancy):
    """The estimator of BASELINE configs[4] (unbiased: MIS-weighted temporal pass, spatial pass with up to four MIS rays per pixel) with the
    G-buffer / candidate / shading passes as three kernels each (fuse_passes 1: what a full-HD frame runs) or one (2: what a rank's band
    runs; the unbiased spatial pass keeps its select / k_trace / finish form at every size).  Street scene with an environment map, three
    frames, Halton-disk and random neighbours (the MIS terms then draw their positions from the pixel's stream, in term order): every
    buffer after every pass."""
    sky = api.env_make_sky(64, 32)
    diffs = run_sequence_both(util.small_street(), 160, 96, frames=3, renderer=api.RENDERER_UNBIASED, scene_kind="street",
                              env=(sky, 64, 32), env_rotation=0.4, low_discrepancy=low_discrepancy, tunables={"fuse_passes": fuse})
    assert not diffs, "\n".join(diffs[:12])


@pytest.mark.gpu
def test_street_sequence_bit_exact(built_lib):
    diffs = run_sequence_both(util.small_street(), 192, 108, frames=2, renderer=api.RENDERER_BIASED, scene_kind="street")
    assert not diffs, "\n".join(diffs)


@pytest.mark.gpu
def test_random_neighbours_and_no_visibility_reuse(built_lib):
    diffs = run_sequence_both(util.bunny_scene(), 128, 80, frames=2, renderer=api.RENDERER_UNBIASED,
                              low_discrepancy=False, reuse_visibility=False)
    assert not diffs, "\n".join(diffs)


@pytest.mark.gpu
def test_headless_driver_matches_pass_by_pass(built_lib):
    """gfxh_restir_render_frame (the C++ frame loop) produces the same beauty buffer as the
    oracle sequenced by the test harness."""
    import torch
    width, height, frames = 128, 72, 3
    hs = util.bunny_scene()
    diffs = run_sequence_both(hs, width, height, frames=frames, renderer=api.RENDERER_BIASED)
    assert not diffs, "\n".join(diffs)
    want = run_sequence_both.last_beauty
    ctx = api.Context(0)
    hs.upload(ctx)
    cfg = api.RestirRenderer.default_config(width, height, api.RENDERER_BIASED)
    cfg.camera = default_camera("bunny", width, height)
    r = api.RestirRenderer(ctx, cfg)
    for _ in range(frames):
        r.render_frame()
    torch.cuda.synchronize()
    out = ctx.read_device(r.beauty_ptr(), width * height * 16).view(np.float32).reshape(-1, 4)
    util.assert_same_bits("driver beauty", out, want)


@pytest.mark.gpu
@pytest.mark.parametrize("renderer,with_env", [(api.RENDERER_BIASED, False), (api.RENDERER_UNBIASED, True)])
def test_row_band_renderers_match_full_frame(built_lib, renderer, with_env):
    """Two band-limited renderers on one GPU (bands 0 and 1 of a 2-way split), halo rows refreshed by
    device copies exactly as HaloExchange would send them, reproduce the full-frame renderer bit for bit."""
    import torch
    from gfxexp_amd import tilesplit
    width, height, frames = 128, 96, 3
    hs = util.bunny_scene()
    ctx = api.Context(0)
    hs.upload(ctx)
    cam = default_camera("bunny", width, height)

    sky = api.env_make_sky(64, 32) if with_env else None     # BASELINE configs[4]: unbiased + environment map, row bands

    def make(band):
        cfg = api.RestirRenderer.default_config(width, height, renderer)
        cfg.camera = cam
        cfg.spatialNeighborRadius = 6.0
        cfg.rowBegin, cfg.rowEnd = band
        r = api.RestirRenderer(ctx, cfg)
        if sky is not None:
            r.set_env(sky.copy(), 64, 32, 0.7, 0.6)
        return r

    full = make((0, 0))
    bands = tilesplit.band_rows(height, 2)
    parts = [make(b) for b in bands]
    views = [tilesplit.renderer_state_views(r, width, height) for r in parts]
    plans = [r.band_plan() for r in parts]
    halo = 12 if renderer == api.RENDERER_BIASED else 6      # radius 6 x spatial passes (2 / 1)
    assert plans[0].haloRows == halo and list(plans[0].recvBelow) == [48, 48 + halo] and list(plans[1].recvAbove) == [48 - halo, 48]
    n = width * height

    def rows(state, key, r, planes=1, comps=1):
        b, e = int(r[0]), int(r[1])
        return [state[key][pl * n * comps + b * width * comps: pl * n * comps + e * width * comps] for pl in range(planes)]

    for _ in range(frames):
        full.render_frame()
        for r in parts:
            r.render_frame()
        torch.cuda.synchronize()
        last = parts[0].params()[2]
        assert last == parts[1].params()[2] == full.params()[2]
        # rank 0 -> rank 1 (rank 1's recvAbove = rank 0's sendBelow) and back
        for src, dst, send, recv in ((0, 1, plans[0].sendBelow, plans[1].recvAbove), (1, 0, plans[1].sendAbove, plans[0].recvBelow)):
            assert int(send[1]) - int(send[0]) == int(recv[1]) - int(recv[0])
            for key, planes, comps in (("rng", 1, 1), (f"info{last}", 1, 2), (f"res{last}", 3, 4)):
                for s_t, d_t in zip(rows(views[src], key, send, planes, comps), rows(views[dst], key, recv, planes, comps)):
                    d_t.copy_(s_t)
        torch.cuda.synchronize()
    want = ctx.read_device(full.beauty_ptr(), n * 16).view(np.float32).reshape(height, width, 4)
    for r, (b, e) in zip(parts, bands):
        got = ctx.read_device(r.beauty_ptr(), n * 16).view(np.float32).reshape(height, width, 4)
        util.assert_same_bits(f"band {b}:{e}", got[b:e], want[b:e])
    assert np.abs(want[..., :3]).sum() > 0


@pytest.mark.gpu
def test_halo_band_renderer_refuses_a_frame_after_the_camera_moved(built_lib):
    """A halo-recompute band renderer (no strip exchange) refreshes radius x passes rows of final state per frame and no motion rows:
    after a camera move its temporal pass would reproject into rows nobody refreshed.  The driver refuses that frame (before touching
    any renderer state) instead of rendering it silently different from the whole-frame renderer; a new sequence is fine."""
    from gfxexp_amd import tilesplit
    width, height = 128, 96
    ctx = api.Context(0)
    util.bunny_scene().upload(ctx)
    cfg = api.RestirRenderer.default_config(width, height, api.RENDERER_BIASED)
    cfg.camera = default_camera("bunny", width, height)
    cfg.spatialNeighborRadius = 6.0
    cfg.rowBegin, cfg.rowEnd = tilesplit.band_rows(height, 2)[0]
    r = api.RestirRenderer(ctx, cfg)
    r.render_frame()
    r.render_frame()
    before = r.params()[2:]
    r.set_camera(api.make_camera(width, height, pos=(1.7, 5.1, 14.0), pitch=12.0, yaw=186.0))
    with pytest.raises(api.GfxError, match="halo-recompute"):
        r.render_frame()
    assert r.params()[2:] == before               # nothing advanced: the caller can fix the configuration and render the same frame
    r.reset()                                     # ... for instance by starting a new sequence, which reads no previous frame
    r.render_frame()
    full_cfg = api.RestirRenderer.default_config(width, height, api.RENDERER_BIASED)
    full_cfg.camera = cfg.camera
    full = api.RestirRenderer(ctx, full_cfg)      # the whole-frame renderer takes the same move without complaint
    full.render_frame()
    full.set_camera(api.make_camera(width, height, pos=(1.7, 5.1, 14.0), pitch=12.0, yaw=186.0))
    full.render_frame()


@pytest.mark.gpu
@pytest.mark.parametrize("env_tables", ["sketch", "interleaved", "guided", "plain"])
@pytest.mark.parametrize("renderer", [api.RENDERER_BIASED, api.RENDERER_UNBIASED])
def test_environment_light_sequence_bit_exact(built_lib, renderer, env_tables):
    """BASELINE config 5 ingredients: environment light (importance-sampled lat-long map, 25 % of the
    candidates) + area lights, unbiased and biased estimators -- with the map's rows interleaved into 32-byte records (envRowTable:
    CDF, PDF, guide and texel of a column side by side), with the rows' inverse-CDF sketches on top (envRowSketch: the column predicted
    to within one, no guide read), as separate arrays with guide tables, and as the plain arrays with the reference's binary searches:
    the same samples bit for bit."""
    w, h = 64, 32
    sky = api.env_make_sky(w, h)
    diffs = run_sequence_both(util.bunny_scene(), 128, 80, frames=2, renderer=renderer, env=(sky, w, h),
                              env_power=0.7, env_rotation=0.6, env_tables=env_tables)
    assert not diffs, "\n".join(diffs)
    beauty = run_sequence_both.last_beauty
    bg = run_sequence_both.last_gb0["instSlot"] == 0xFFFFFFFF
    assert bg.any() and beauty[bg, :3].min() > 0.02      # background shows the sky, not the 0.01 grey


@pytest.mark.gpu
def test_environment_light_only(built_lib):
    """No emissive geometry at all: every candidate comes from the environment map."""
    w, h = 48, 24
    sky = api.env_make_sky(w, h, sun_elevation=50.0)
    diffs = run_sequence_both(util.bunny_scene(with_light=False), 96, 64, frames=2, renderer=api.RENDERER_BIASED, env=(sky, w, h))
    assert not diffs, "\n".join(diffs)


def _moving_scene_animation(frame):
    """Frame-by-frame transforms of the two lights (slots 2, 3) and the bunny (slot 0) of util.bunny_scene."""
    t = 0.5 - 0.5 * np.cos(2 * np.pi * frame / 5.0)
    light = api.make_transform(pos=(-3.0 + 6.0 * t, 12.0 - 2.0 * t, 2.0), yaw=40.0 * t)
    light2 = api.make_transform(pitch=-60.0 + 25.0 * t, pos=(-6.0, 6.0 + t, 6.0), scale=1.0 + 0.5 * t)
    bunny = api.make_transform(scale=0.1, yaw=70.0 * t, pos=(1.5 * t, 0.0, 0.0))
    moves = [(2, light, None), (0, bunny, None)]
    if frame % 2 == 1:                       # this one only moves every other frame (stale curToPrev in between)
        # with the controller's own normal matrix: rotation / scale
        s = 1.0 + 0.5 * t
        rot = light2.reshape(3, 4)[:, :3] / np.float32(s)
        moves.append((3, light2, (rot / np.float32(s)).astype(np.float32).reshape(9)))
    return moves


@pytest.mark.gpu
@pytest.mark.parametrize("renderer", [api.RENDERER_BIASED, api.RENDERER_UNBIASED])
def test_animated_instances_motion_vectors_and_temporal_reuse(built_lib, renderer):
    """Moving emitters and a rotating mesh: per-frame gfx_instance_set_transform + rebuild, motion vectors from
    curToPrevTransform = prev * invert(cur), temporal reuse across the motion; every buffer bit for bit."""
    hs = util.bunny_scene(with_light=True)
    diffs = run_sequence_both(hs, 96, 64, frames=4, renderer=renderer, scene_kind="bunny", animate=_moving_scene_animation)
    assert not diffs, "\n".join(diffs[:12])
    # the motion vectors are not all zero: the G-buffer saw the instances move
    assert np.abs(run_sequence_both.last_motion).max() > 0.5


@pytest.mark.gpu
def test_outputs_consumed_orders_the_pipelined_gbuffer_pass(built_lib):
    """gfxh_restir_outputs_consumed / gfxh_nrc_outputs_consumed: a caller that reads the albedo accumulator on its own stream after every frame
    and says so gets, frame by frame, the albedo buffer of THAT frame (equal to a serial run's), and the frames themselves do not change."""
    import torch
    from bench import _device_view
    width, height, frames = 192, 108, 5
    n = width * height

    def run(kind, serial, monkeypatch_env):
        import os
        os.environ["GFX_SERIAL_FRAMES"] = "1" if serial else "0"
        try:
            ctx = api.Context(0)
            hs = util.small_street()
            hs.upload(ctx)
            if kind == "restir":
                cfg = api.RestirRenderer.default_config(width, height, api.RENDERER_BIASED)
                cfg.camera = default_camera("street", width, height)
                r = api.RestirRenderer(ctx, cfg)
                consumed = lambda st: api.lib().gfxh_restir_outputs_consumed(r.h, C.c_void_p(st))
                s, f, cur, base, _ = r.params()
                albedo_ptr = s.albedoAccumBuffer
            else:
                cfg = api.NrcRenderer.default_config(width, height, hs.bounds())
                cfg.camera = default_camera("street", width, height)
                r = api.NrcRenderer(ctx, cfg)
                consumed = lambda st: api.lib().gfxh_nrc_outputs_consumed(r.h, C.c_void_p(st))
                albedo_ptr = None
        finally:
            del os.environ["GFX_SERIAL_FRAMES"]
        stream = torch.cuda.Stream()
        snaps = []
        for _ in range(frames):
            r.render_frame(stream.cuda_stream)
            if albedo_ptr:
                with torch.cuda.stream(stream):
                    big = _device_view(albedo_ptr, 4 * n).clone()      # the caller's read of the accumulator, on its own stream
                    for _ in range(20):
                        big = big * 1.0                                # ... and some more work behind it
                    snaps.append(big)
            assert consumed(stream.cuda_stream) == 0
        if kind == "nrc":
            r.network()
        torch.cuda.synchronize()
        beauty = ctx.read_device(r.beauty_ptr(), 16 * n).copy()
        out = [t.cpu().numpy().copy() for t in snaps]
        r.close()
        ctx.close()
        return beauty, out

    b_serial, a_serial = run("restir", True, None)
    b_piped, a_piped = run("restir", False, None)
    assert np.array_equal(b_serial, b_piped), "beauty differs between the pipelined and the serial loop"
    for k, (x, y) in enumerate(zip(a_serial, a_piped)):
        assert np.array_equal(x.view(np.uint32), y.view(np.uint32)), f"frame {k}: the caller read an albedo buffer the next G-buffer pass had already rewritten"
    b_nrc, _ = run("nrc", False, None)                  # (training order noise: no bit comparison between two NRC runs)
    assert np.isfinite(np.frombuffer(b_nrc, np.float32)).all()


@pytest.mark.gpu
def test_headless_driver_with_animated_instances(built_lib):
    """The C++ frame loop (pipelined G-buffer pass included) with per-frame instance updates and in-place BVH
    rebuilds ends on the same beauty buffer as the oracle sequenced pass by pass."""
    import torch
    width, height, frames = 96, 64, 4
    diffs = run_sequence_both(util.bunny_scene(), width, height, frames=frames, renderer=api.RENDERER_BIASED,
                              animate=_moving_scene_animation)
    assert not diffs, "\n".join(diffs[:12])
    want = run_sequence_both.last_beauty
    ctx = api.Context(0)
    util.bunny_scene().upload(ctx)
    cfg = api.RestirRenderer.default_config(width, height, api.RENDERER_BIASED)
    cfg.camera = default_camera("bunny", width, height)
    r = api.RestirRenderer(ctx, cfg)
    for frame in range(frames):
        for inst_slot, xfm, nm in _moving_scene_animation(frame):
            ctx.instance_set_transform(inst_slot, xfm, nm)
        r.rebuild_accel()
        r.render_frame()
    torch.cuda.synchronize()
    out = ctx.read_device(r.beauty_ptr(), width * height * 16).view(np.float32).reshape(-1, 4)
    util.assert_same_bits("driver beauty (animated)", out, want)


@pytest.mark.gpu
@pytest.mark.parametrize("fuse", [1, 2])
def test_spatial_and_shading_as_one_pass(built_lib, fuse):
    """GFX_RESTIR_SPATIAL_BIASED_AND_SHADING = the last biased spatial pass followed by the shading of what it wrote, as two
    launches (fuse_passes 1) or one kernel (2): beauty, reservoirs, RNG states equal to the oracle's two passes, bit for bit."""
    import torch
    hs = util.small_street()
    width, height = 192, 108
    ctx = api.Context(0)
    ctx.tunable_set("fuse_passes", fuse)
    hs.upload(ctx)
    accel = ctx.accel_build()
    ctx.lights_build_static()
    osc = util.feed_oracle(hs)
    cam = default_camera("street", width, height)
    ocam = util.copy_struct(O.GfxCamera, cam)
    pb_cpu = util.PixelBuffers(width, height)
    dev = util.DeviceBuffers(util.PixelBuffers(width, height))
    s_gpu, s_cpu = dev.static_params(), pb_cpu.host_static_params()
    stream = torch.cuda.current_stream().cuda_stream
    last_res, last_base, nb = 1, 0, 5
    for frame in range(2):
        kw = dict(frameIndex=frame, bufferIndex=frame % 2, resetFlowBuffer=int(frame == 0), numAccumFrames=0, numSpatialNeighbors=nb)
        f_gpu = util.frame_params(api.GfxRestirFrameParams, api.GfxCamera, width, height, cam, travHandle=accel, **kw)
        f_cpu = util.frame_params(O.GfxRestirFrameParams, O.GfxCamera, width, height, ocam, travHandle=0, **kw)
        ctx.lights_build_instances(stream)
        cur = (last_res + 1) % 2
        entry = api.PASS_INITIAL_RIS if frame == 0 else api.PASS_INITIAL_TEMPORAL_BIASED
        for pass_id, cur_res, base in ((api.PASS_SETUP_GBUFFERS, cur, last_base), (entry, cur, last_base), (api.PASS_SPATIAL_BIASED, cur, last_base),
                                       (api.PASS_SPATIAL_BIASED_AND_SHADING, (cur + 1) % 2, last_base + nb)):
            ctx.restir_set_params(s_gpu, f_gpu, cur_res, base, stream)
            ctx.restir_launch(pass_id, width, height, stream)
            if pass_id == api.PASS_SPATIAL_BIASED_AND_SHADING:      # the oracle: the two passes it stands for
                osc.restir_launch(s_cpu, f_cpu, cur_res, base, api.PASS_SPATIAL_BIASED)
                osc.restir_launch(s_cpu, f_cpu, (cur_res + 1) % 2, base + nb, api.PASS_SHADING)
            else:
                osc.restir_launch(s_cpu, f_cpu, cur_res, base, pass_id)
        last_base += 2 * nb
        last_res = cur                                                # two flips
        got, want = dev.download(), pb_cpu.arrays()
        for k in want:
            assert np.array_equal(np.ascontiguousarray(got[k]).view(np.uint8), np.ascontiguousarray(want[k]).view(np.uint8)), f"frame {frame}: {k}"
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("band", [(136, 272), (128, 416)])
def test_block_start_order_of_the_fused_kernels_changes_nothing(built_lib, band):
    """A band of a 1920-wide frame through the headless driver, four frames: 136 rows (four lanes per pixel: 4 080 blocks of
    k_initial_fused start in cost order) and 288 rows (one lane per pixel: 2 160 blocks of k_initial_fused and of k_shading_fused do).
    Index order ("block_order" 0) and cost order (1, in effect from the second frame on) leave the same beauty, reservoirs, reservoir
    infos and RNG states, bit for bit -- and so does the three-kernel form."""
    import torch
    from gfxexp_amd import scenes
    W, H = 1920, 544
    hs = scenes.bench_street(textured=True)
    cam = api.make_camera(W, H, pos=(1.5, 2.2, 52.0), pitch=4.0, yaw=181.5)

    def render(tunables):
        ctx = api.Context(0)
        for k, v in tunables.items():
            ctx.tunable_set(k, v)
        hs.upload(ctx)
        cfg = api.RestirRenderer.default_config(W, H, api.RENDERER_BIASED)
        cfg.camera = cam
        cfg.enableBumpMapping = 1
        cfg.rowBegin, cfg.rowEnd = band
        r = api.RestirRenderer(ctx, cfg)
        r.set_exchange(lambda stream, d: None, 0)
        for _ in range(4):
            r.render_frame()
        torch.cuda.synchronize()
        s, f, cur, base, _ = r.params()
        n = W * H
        out = {"beauty": ctx.read_device(r.beauty_ptr(), 16 * n), "rng": ctx.read_device(s.rngBuffer, 8 * n)}
        for i in range(2):
            out[f"res{i}"] = ctx.read_device(s.reservoirBuffer[i], 48 * n)
            out[f"info{i}"] = ctx.read_device(s.reservoirInfoBuffer[i], 8 * n)
        rows = slice(band[0] * W, band[1] * W)
        out = {k: (np.frombuffer(v, np.uint8).reshape(3, n, 16)[:, rows] if k.startswith("res") else np.frombuffer(v, np.uint8).reshape(n, -1)[rows]).copy()
               for k, v in out.items()}
        r.close()
        ctx.close()
        return out

    ordered = render({"block_order": 1})
    for name, other in (("index order", render({"block_order": 0})), ("three kernels per ray pass", render({"fuse_passes": 1}))):
        for k in ordered:
            assert np.array_equal(ordered[k], other[k]), f"{name}: {k} differs"
