"""-m gpu: the rearchitected ReSTIR passes (light pre-sampling, per-pixel RIS on 8x8 tiles,
traceShadowRays<T,S,U>, shadeAndResample<T,S>) through the C ABI against the CPU oracle; every
buffer is compared bit for bit after every pass."""
import numpy as np
import pytest

from gfxexp_amd import api
from oracle import oracle as O
from tests import util


def run_rearch_both(hs, width, height, frames, temporal, spatial, unbiased, low_discrepancy=True,
                    reuse_vis_spatiotemporal=False, camera=None, env=None):
    import torch
    ctx = api.Context(0)
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
    s_gpu, s_cpu = dev.static_params(), pb_cpu.host_static_params()
    stream = torch.cuda.current_stream().cuda_stream
    diffs = []

    def compare(tag):
        got, want = dev.download(), pb_cpu.arrays()
        for k in want:
            a = np.ascontiguousarray(got[k]).view(np.uint8).reshape(-1)
            b = np.ascontiguousarray(want[k]).view(np.uint8).reshape(-1)
            if not np.array_equal(a, b):
                item = want[k].dtype.itemsize
                nbad = len(np.unique(np.nonzero(a != b)[0] // item))
                diffs.append(f"{tag}: {k}: {nbad} of {want[k].size} elements differ")

    last_res, last_base = 1, 0
    for frame in range(frames):
        kw = dict(frameIndex=frame, bufferIndex=frame % 2, resetFlowBuffer=int(frame == 0), numAccumFrames=0,
                  numSpatialNeighbors=1, enableTemporalReuse=int(temporal), enableSpatialReuse=int(spatial),
                  useUnbiasedEstimator=int(unbiased), useLowDiscrepancyNeighbors=int(low_discrepancy),
                  reuseVisibilityForSpatiotemporal=int(reuse_vis_spatiotemporal), enableEnvLight=int(env is not None))
        f_gpu = util.frame_params(api.GfxRestirFrameParams, api.GfxCamera, width, height, cam, travHandle=accel, **kw)
        f_cpu = util.frame_params(O.GfxRestirFrameParams, O.GfxCamera, width, height, ocam, travHandle=0, **kw)
        ctx.lights_build_instances(stream)
        cur = (last_res + 1) % 2
        trace_pass, shade_pass = api.rearch_passes(temporal, spatial, unbiased, frame == 0)
        ctx.restir_set_params(s_gpu, f_gpu, cur, last_base, stream)
        for tag, pass_id in (("gbuffer", api.PASS_SETUP_GBUFFERS), ("presample", api.PASS_LIGHT_PRESAMPLING),
                             ("per-pixel RIS", api.PASS_PER_PIXEL_RIS), ("trace shadow rays", trace_pass),
                             ("shade and resample", shade_pass)):
            ctx.restir_launch(pass_id, width, height, stream)
            osc.restir_launch(s_cpu, f_cpu, cur, last_base, pass_id)
            compare(f"frame {frame} {tag}")
        last_base += 1
        last_res = cur
    return diffs


@pytest.mark.gpu
@pytest.mark.parametrize("temporal,spatial", [(False, False), (True, False), (False, True), (True, True)])
@pytest.mark.parametrize("unbiased", [False, True])
def test_rearchitected_bunny_bit_exact(built_lib, temporal, spatial, unbiased):
    diffs = run_rearch_both(util.bunny_scene(), 128, 96, 3, temporal, spatial, unbiased)
    assert not diffs, "\n".join(diffs[:12])


@pytest.mark.gpu
def test_rearchitected_random_neighbours_and_spatial_visibility_reuse(built_lib):
    diffs = run_rearch_both(util.bunny_scene(), 96, 64, 3, True, True, False, low_discrepancy=False, reuse_vis_spatiotemporal=True)
    assert not diffs, "\n".join(diffs[:12])


@pytest.mark.gpu
def test_rearchitected_street_with_env_light_unbiased(built_lib):
    w, h = 64, 32
    sky = api.env_make_sky(w, h)
    cam = api.make_camera(96, 64, pos=(2.0, 5.0, 26.0), pitch=6.0, yaw=184.0)
    diffs = run_rearch_both(util.small_street(), 96, 64, 3, True, True, True, camera=cam, env=(sky, w, h))
    assert not diffs, "\n".join(diffs[:12])


@pytest.mark.gpu
def test_rearchitected_ragged_image_size(built_lib):
    """Width/height that are not multiples of the 8x8 tile."""
    diffs = run_rearch_both(util.bunny_scene(), 101, 67, 2, True, True, False)
    assert not diffs, "\n".join(diffs[:12])


@pytest.mark.gpu
@pytest.mark.parametrize("renderer,unbiased", [(api.RENDERER_REARCH_BIASED, False), (api.RENDERER_REARCH_UNBIASED, True)])
def test_headless_driver_rearchitected_renderers(built_lib, renderer, unbiased):
    """gfxh_restir with the rearchitected renderers (frame loop restir_di_main.cpp:2423-2487) equals
    a hand-sequenced run of the same passes (which the other tests tie to the oracle)."""
    import torch
    width, height, frames = 96, 64, 3
    hs = util.bunny_scene()
    ctx = api.Context(0)
    hs.upload(ctx)
    cfg = api.RestirRenderer.default_config(width, height, renderer)
    assert (cfg.numSpatialReusePasses, cfg.numSpatialNeighbors, cfg.log2NumCandidateSamples) == (1, 1, 5)
    cam = api.make_camera(width, height, pos=(1.5, 5.0, 14.0), pitch=12.0, yaw=186.0)
    cfg.camera = cam
    r = api.RestirRenderer(ctx, cfg)
    for _ in range(frames):
        r.render_frame()
    torch.cuda.synchronize()
    out = ctx.read_device(r.beauty_ptr(), width * height * 16).view(np.float32).reshape(-1, 4)

    # the same frames through the oracle
    osc = util.feed_oracle(hs)
    pb = util.PixelBuffers(width, height)
    s = pb.host_static_params()
    ocam = util.copy_struct(O.GfxCamera, cam)
    last_res, last_base = 1, 0
    for frame in range(frames):
        f = util.frame_params(O.GfxRestirFrameParams, O.GfxCamera, width, height, ocam, frameIndex=frame,
                              bufferIndex=frame % 2, resetFlowBuffer=int(frame == 0), numAccumFrames=0,
                              numSpatialNeighbors=1, useUnbiasedEstimator=int(unbiased), reuseVisibilityForSpatiotemporal=0)
        cur = (last_res + 1) % 2
        trace_pass, shade_pass = api.rearch_passes(True, True, unbiased, frame == 0)
        for pass_id in (api.PASS_SETUP_GBUFFERS, api.PASS_LIGHT_PRESAMPLING, api.PASS_PER_PIXEL_RIS, trace_pass, shade_pass):
            osc.restir_launch(s, f, cur, last_base, pass_id)
        last_base += 1
        last_res = cur
    util.assert_same_bits("driver beauty", out, pb.beauty)


@pytest.mark.gpu
def test_light_sampling_pathological_distribution(built_lib):
    """131072 pre-sampled lights + per-pixel RIS + shading on a scene built to stress the light-distribution
    tables (guide table brackets, tabulated probabilities, zero-weight runs), bit for bit."""
    hs = util.pathological_light_scene()
    cam = api.make_camera(48, 32, pos=(0.0, 9.0, 38.0), pitch=10.0, yaw=180.0)
    assert run_rearch_both(hs, 48, 32, 2, temporal=True, spatial=False, unbiased=False, camera=cam) == []
