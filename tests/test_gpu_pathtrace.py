"""-m gpu: the baseline path tracer (BASELINE.json config 2) through the C ABI against the CPU
oracle: RNG states, G-buffers and the accumulated beauty buffer are compared bit for bit after
every frame."""
import numpy as np
import pytest

from gfxexp_amd import api
from oracle import oracle as O
from tests import util


def run_pt_both(hs, width, height, frames=2, max_len=5, camera=None, env=None, jitter=0, env_rotation=0.0, rows=None, fuse=None, regen=None):
    import torch
    ctx = api.Context(0)
    if fuse is not None:
        ctx.tunable_set("fuse_passes", fuse)
    if regen is not None:
        ctx.tunable_set("pt_regen", regen)
    hs.upload(ctx)
    accel = ctx.accel_build()
    ctx.lights_build_static()
    osc = util.feed_oracle(hs)
    cam = camera if camera is not None else api.make_camera(width, height, pos=(1.5, 5.0, 14.0), pitch=12.0, yaw=186.0)
    ocam = util.copy_struct(O.GfxCamera, cam)
    pb_gpu_init = util.PixelBuffers(width, height)
    pb_cpu = util.PixelBuffers(width, height)
    if env is not None:
        pb_gpu_init.set_env(*env)
        pb_cpu.set_env(*env, oracle_side=True)
    dev = util.DeviceBuffers(pb_gpu_init)
    s_gpu = dev.static_params()
    s_cpu = pb_cpu.host_static_params()
    stream = torch.cuda.current_stream().cuda_stream
    diffs = []
    for frame in range(frames):
        kw = dict(frameIndex=frame, bufferIndex=frame % 2, resetFlowBuffer=int(frame == 0), numAccumFrames=frame,
                  enableJittering=jitter, enableEnvLight=int(env is not None), envLightRotation=env_rotation)
        f_gpu = util.frame_params(api.GfxRestirFrameParams, api.GfxCamera, width, height, cam, travHandle=accel, **kw)
        f_cpu = util.frame_params(O.GfxRestirFrameParams, O.GfxCamera, width, height, ocam, travHandle=0, **kw)
        ctx.lights_build_instances(stream)
        ctx.restir_set_params(s_gpu, f_gpu, 0, 0, stream)
        bands = rows if rows else [(0, 0)]
        for pass_id in (api.PT_SETUP_GBUFFERS, api.PT_PATH_TRACE_BASELINE):
            for (rb, re) in bands:
                ctx.pt_launch(pass_id, width, height, max_len, rb, re, stream)
            osc.pt_launch(s_cpu, f_cpu, pass_id, max_len)
        got, want = dev.download(), pb_cpu.arrays()
        for k in ("rng", "beauty", f"gb0_{frame % 2}", f"gb1_{frame % 2}", "albedo", "normal"):
            a = np.ascontiguousarray(got[k]).view(np.uint8).reshape(-1)
            b = np.ascontiguousarray(want[k]).view(np.uint8).reshape(-1)
            if not np.array_equal(a, b):
                item = want[k].dtype.itemsize
                nbad = len(np.unique(np.nonzero(a != b)[0] // item))
                diffs.append(f"frame {frame}: {k}: {nbad} of {want[k].size} elements differ")
    run_pt_both.last_beauty = pb_cpu.beauty.copy()
    return diffs


@pytest.mark.gpu
@pytest.mark.parametrize("fuse", [1, 2])
def test_wavefront_and_one_kernel_forms_of_the_path_tracer(built_lib, fuse):
    """fuse_passes 1: k_pt_first, two k_trace launches, k_pt_apply_nee and k_pt_bounce per bounce through the ray queues (what a full-HD
    frame runs); 2: the whole path of a pixel in k_pt_fused (what a small launch runs).  Both against the oracle, street scene with an
    environment map (implicit environment hits with MIS), path length 6, three frames with accumulation."""
    sky = api.env_make_sky(64, 32)
    diffs = run_pt_both(util.small_street(), 160, 96, frames=3, max_len=6, camera=api.make_camera(160, 96, pos=(2.0, 5.0, 26.0), pitch=6.0, yaw=184.0),
                        env=(sky, 64, 32), env_rotation=0.4, fuse=fuse)
    assert not diffs, "\n".join(diffs)


@pytest.mark.gpu
@pytest.mark.parametrize("regen,max_len", [(1, 6), (1, 1), (2, 3), (0, 6)])
def test_path_regeneration_changes_no_pixel(built_lib, regen, max_len):
    """k_pt_regen: a launch of `regen` blocks per CU whose lanes draw pixels from a ticket until none are left (what configs[1] runs),
    against the oracle -- 352 x 256 is more blocks than one (two) per CU, so lanes do regenerate; regen 0 keeps k_pt_fused covered.
    Street scene with an environment map, accumulation over three frames; path length 1 = the forced single extension."""
    sky = api.env_make_sky(64, 32)
    w, h = (352, 256) if regen < 2 else (640, 400)
    diffs = run_pt_both(util.small_street(), w, h, frames=3, max_len=max_len, camera=api.make_camera(w, h, pos=(2.0, 5.0, 26.0), pitch=6.0, yaw=184.0),
                        env=(sky, 64, 32), env_rotation=0.4, fuse=2, regen=regen)
    assert not diffs, "\n".join(diffs)


@pytest.mark.gpu
@pytest.mark.parametrize("max_len", [2, 5, 15])
def test_bunny_path_tracer_bit_exact(built_lib, max_len):
    diffs = run_pt_both(util.bunny_scene(), 128, 96, frames=2, max_len=max_len)
    assert not diffs, "\n".join(diffs)
    b = run_pt_both.last_beauty
    assert np.all(np.isfinite(b)) and b[:, :3].max() > 0.0


@pytest.mark.gpu
def test_path_tracer_config2_resolution_bit_exact(built_lib):
    """BASELINE.json configs[1]: bunny-class scene, 512x512, max path length 5."""
    diffs = run_pt_both(util.bunny_scene(), 512, 512, frames=1, max_len=5, jitter=1,
                        camera=api.make_camera(512, 512, pos=(1.5, 5.0, 14.0), pitch=12.0, yaw=186.0))
    assert not diffs, "\n".join(diffs)


@pytest.mark.gpu
def test_path_tracer_env_light_bit_exact(built_lib):
    w, h = 64, 32
    sky = api.env_make_sky(w, h)
    diffs = run_pt_both(util.bunny_scene(), 96, 64, frames=2, max_len=4, env=(sky, w, h), env_rotation=0.7)
    assert not diffs, "\n".join(diffs)


@pytest.mark.gpu
def test_path_tracer_street_instances_bit_exact(built_lib):
    hs = util.small_street()
    cam = api.make_camera(96, 64, pos=(2.0, 5.0, 26.0), pitch=6.0, yaw=184.0)
    diffs = run_pt_both(hs, 96, 64, frames=2, max_len=5, camera=cam)
    assert not diffs, "\n".join(diffs)


@pytest.mark.gpu
def test_path_tracer_row_bands_equal_full_frame(built_lib):
    """Row bands (the multi-GPU unit) launched one after the other reproduce the full frame."""
    diffs = run_pt_both(util.bunny_scene(), 96, 64, frames=2, max_len=5, rows=[(0, 24), (24, 64)])
    assert not diffs, "\n".join(diffs)


@pytest.mark.gpu
def test_headless_driver_path_trace_mode_with_accumulation(built_lib):
    """gfxh_restir with renderer = GFXH_PATH_TRACE_BASELINE (the frame loop of path_tracing_main.cpp)
    accumulating 3 frames equals the oracle sequenced by the harness."""
    import torch
    width, height, frames = 96, 64, 3
    hs = util.bunny_scene()
    diffs = run_pt_both(hs, width, height, frames=frames, max_len=5)
    assert not diffs, "\n".join(diffs)
    want = run_pt_both.last_beauty
    ctx = api.Context(0)
    hs.upload(ctx)
    cfg = api.RestirRenderer.default_config(width, height, api.RENDERER_PATH_TRACE)
    assert cfg.maxPathLength == 5
    cfg.enableAccumulation = 1
    cfg.camera = api.make_camera(width, height, pos=(1.5, 5.0, 14.0), pitch=12.0, yaw=186.0)
    r = api.RestirRenderer(ctx, cfg)
    for _ in range(frames):
        r.render_frame()
    torch.cuda.synchronize()
    out = ctx.read_device(r.beauty_ptr(), width * height * 16).view(np.float32).reshape(-1, 4)
    util.assert_same_bits("driver beauty", out, want)


@pytest.mark.gpu
def test_solid_angle_triangle_sampling_bit_exact(built_lib):
    """sampleLight<true> + the solid-angle hypothetical density of computeSurfacePoint (restir_di_shared.h:417-483,
    path_tracing_shared.h:319-385, 550-568): compiled out in the reference (useSolidAngleSampling = false), a run-time
    switch here.  Bit-exact against the oracle, and a different image from area sampling."""
    with util.frame_overrides(useSolidAngleSampling=1):
        diffs = run_pt_both(util.bunny_scene(), 128, 80, frames=2, max_len=5)
    assert not diffs, "\n".join(diffs[:10])
    solid = run_pt_both.last_beauty.copy()
    diffs = run_pt_both(util.bunny_scene(), 128, 80, frames=2, max_len=5)
    assert not diffs, "\n".join(diffs[:10])
    area = run_pt_both.last_beauty
    assert np.isfinite(solid).all() and not np.array_equal(solid, area)
    # same estimator in expectation: the image means agree within the noise of two 1-spp frames
    assert abs(solid[:, :3].mean() - area[:, :3].mean()) < 0.25 * area[:, :3].mean()
