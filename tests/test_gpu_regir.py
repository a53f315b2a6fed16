"""-m gpu: ReGIR (grid-cell streaming RIS + the ReGIR path tracer) through the C ABI against the
CPU oracle: light-slot reservoirs, slot RNGs, cell bookkeeping, pixel RNGs and the beauty buffer are
compared bit for bit after every pass."""
import numpy as np
import pytest

from gfxexp_amd import api
from oracle import oracle as O
from tests import util


def run_regir_both(hs, width, height, frames, max_len, temporal=True, dims=(8, 4, 8), randomize=1, env=None, camera=None,
                   log2_slot=3, log2_cell=2, fuse=None):
    import torch
    ctx = api.Context(0)
    if fuse is not None:
        ctx.tunable_set("fuse_passes", fuse)
    hs.upload(ctx)
    accel = ctx.accel_build()
    ctx.lights_build_static()
    osc = util.feed_oracle(hs)
    cam = camera if camera is not None else api.make_camera(width, height, pos=(1.5, 5.0, 14.0), pitch=12.0, yaw=186.0)
    ocam = util.copy_struct(O.GfxCamera, cam)
    pb_gpu_init, pb_cpu = util.PixelBuffers(width, height), util.PixelBuffers(width, height)
    if env is not None:
        pb_gpu_init.set_env(*env)
        pb_cpu.set_env(*env, oracle_side=True)
    dev = util.DeviceBuffers(pb_gpu_init)
    s_gpu, s_cpu = dev.static_params(), pb_cpu.host_static_params()
    rb_gpu = util.RegirBuffers(hs.bounds(), dims, log2_slot, log2_cell, randomize)
    rb_cpu = util.RegirBuffers(hs.bounds(), dims, log2_slot, log2_cell, randomize)
    ctx.regir_set_params(rb_gpu.device_params())
    osc.regir_set_params(rb_cpu.host_params())
    stream = torch.cuda.current_stream().cuda_stream
    diffs = []

    def compare(tag):
        got, want = dev.download(), pb_cpu.arrays()
        got.update(rb_gpu.download()); want.update(rb_cpu.arrays())
        for k in want:
            if k.startswith(("res_", "info_", "vis_", "presample", "gb2", "gb3")):
                continue
            a = np.ascontiguousarray(got[k]).view(np.uint8).reshape(-1)
            b = np.ascontiguousarray(want[k]).view(np.uint8).reshape(-1)
            if not np.array_equal(a, b):
                item = want[k].dtype.itemsize
                nbad = len(np.unique(np.nonzero(a != b)[0] // item))
                diffs.append(f"{tag}: {k}: {nbad} of {want[k].size} elements differ")

    for frame in range(frames):
        kw = dict(frameIndex=frame, bufferIndex=frame % 2, resetFlowBuffer=int(frame == 0), numAccumFrames=frame,
                  enableEnvLight=int(env is not None))
        f_gpu = util.frame_params(api.GfxRestirFrameParams, api.GfxCamera, width, height, cam, travHandle=accel, **kw)
        f_cpu = util.frame_params(O.GfxRestirFrameParams, O.GfxCamera, width, height, ocam, travHandle=0, **kw)
        ctx.lights_build_instances(stream)
        ctx.restir_set_params(s_gpu, f_gpu, 0, 0, stream)
        build = api.PT_REGIR_BUILD_CELLS_TEMPORAL if (temporal and frame > 0) else api.PT_REGIR_BUILD_CELLS
        for tag, pass_id in (("gbuffer", api.PT_SETUP_GBUFFERS), ("build cells", build),
                             ("path trace", api.PT_PATH_TRACE_REGIR), ("update last access", api.PT_REGIR_UPDATE_LAST_ACCESS)):
            ctx.pt_launch(pass_id, width, height, max_len, 0, 0, stream)
            osc.pt_launch(s_cpu, f_cpu, pass_id, max_len)
            compare(f"frame {frame} {tag}")
    return diffs


@pytest.mark.gpu
@pytest.mark.parametrize("max_len", [2, 5])
def test_regir_bunny_bit_exact(built_lib, max_len):
    diffs = run_regir_both(util.bunny_scene(), 128, 96, 3, max_len)
    assert not diffs, "\n".join(diffs[:12])


@pytest.mark.gpu
@pytest.mark.parametrize("fuse", [1, 2])
def test_regir_wavefront_and_one_kernel_forms(built_lib, fuse):
    """The ReGIR path tracer as a wavefront of launches (fuse_passes 1: what a full-HD frame runs) and as k_pt_fused<REGIR> (2): cell
    reservoirs, access counters, RNG states and beauty against the oracle, street scene, path length 5, three frames."""
    cam = api.make_camera(128, 80, pos=(2.0, 5.0, 26.0), pitch=6.0, yaw=184.0)
    diffs = run_regir_both(util.small_street(), 128, 80, 3, 5, camera=cam, dims=(8, 2, 8), fuse=fuse)
    assert not diffs, "\n".join(diffs[:12])


@pytest.mark.gpu
def test_regir_without_temporal_reuse_or_randomization(built_lib):
    diffs = run_regir_both(util.bunny_scene(), 96, 64, 2, 4, temporal=False, randomize=0, log2_slot=2, log2_cell=3)
    assert not diffs, "\n".join(diffs[:12])


@pytest.mark.gpu
def test_regir_cells_expire_after_eight_frames(built_lib):
    diffs = run_regir_both(util.bunny_scene(), 64, 48, 10, 3, dims=(4, 2, 4))
    assert not diffs, "\n".join(diffs[:12])


@pytest.mark.gpu
def test_regir_street_with_env_light(built_lib):
    w, h = 64, 32
    sky = api.env_make_sky(w, h)
    cam = api.make_camera(96, 64, pos=(2.0, 5.0, 26.0), pitch=6.0, yaw=184.0)
    diffs = run_regir_both(util.small_street(), 96, 64, 3, 5, env=(sky, w, h), camera=cam, dims=(8, 2, 8))
    assert not diffs, "\n".join(diffs[:12])


@pytest.mark.gpu
def test_headless_driver_regir_mode(built_lib):
    """gfxh_restir with renderer = GFXH_PATH_TRACE_REGIR (frame loop of regir_main.cpp:2021-2066)
    reproduces the oracle sequenced by the harness."""
    import torch
    hs = util.bunny_scene()
    width, height, frames, dims = 96, 64, 3, (8, 4, 8)
    ctx = api.Context(0)
    hs.upload(ctx)
    cam = api.make_camera(width, height, pos=(1.5, 5.0, 14.0), pitch=12.0, yaw=186.0)
    cfg = api.RestirRenderer.default_config(width, height, api.RENDERER_PATH_TRACE_REGIR)
    assert list(cfg.regirGridDimension) == [32, 8, 32] and cfg.regirLog2CandidatesPerLightSlot == 3
    cfg.camera = cam
    b = hs.bounds()
    for k in range(3):
        cfg.regirAabbMin[k] = float(b[k]); cfg.regirAabbMax[k] = float(b[3 + k]); cfg.regirGridDimension[k] = dims[k]
    r = api.RestirRenderer(ctx, cfg)
    for _ in range(frames):
        r.render_frame()
    torch.cuda.synchronize()
    out = ctx.read_device(r.beauty_ptr(), width * height * 16).view(np.float32).reshape(-1, 4)

    osc = util.feed_oracle(hs)
    pb = util.PixelBuffers(width, height)
    rb = util.RegirBuffers(hs.bounds(), dims)
    s = pb.host_static_params()
    osc.regir_set_params(rb.host_params())
    ocam = util.copy_struct(O.GfxCamera, cam)
    for frame in range(frames):
        f = util.frame_params(O.GfxRestirFrameParams, O.GfxCamera, width, height, ocam, frameIndex=frame, bufferIndex=frame % 2,
                              resetFlowBuffer=int(frame == 0), numAccumFrames=0)
        build = api.PT_REGIR_BUILD_CELLS_TEMPORAL if frame > 0 else api.PT_REGIR_BUILD_CELLS
        for pass_id in (api.PT_SETUP_GBUFFERS, build, api.PT_PATH_TRACE_REGIR, api.PT_REGIR_UPDATE_LAST_ACCESS):
            osc.pt_launch(s, f, pass_id, 5)
    util.assert_same_bits("driver beauty", out, pb.beauty)
