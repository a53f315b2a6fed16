"""-m gpu: the BASELINE configurations at their full size (1920x1080, the 2.5 M-triangle street stand-in of bench.py).

Parity at this size is established by
  * whole frames (test_every_pixel_*): on the GPU box's 128 host threads the oracle renders a 1920x1080 pass in seconds, so
    every pixel of every buffer is compared after every pass of two (animated: three) frames for configs[2], configs[4], the
    animated workload, the rearchitected set, the path tracer and the NRC render side (textured street), and for a 3840x2160 frame;
  * an oracle run restricted to a window of the frame (plus the margin the reuse passes read), compared bit for bit with the
    same pixels of the full-frame GPU run after every pass -- configs[4] on the other workloads, border windows, the quick form;
  * properties that do not depend on the size: run-to-run determinism (ray-queue slots are handed out by
    atomics in a different order every run), pipelined == serial frame loop, band split == full frame.
"""
import numpy as np
import pytest

from gfxexp_amd import api
from oracle import oracle as O
from tests import util

pytestmark = pytest.mark.gpu
W, H = 1920, 1080
CAM = dict(pos=(1.5, 2.2, 52.0), pitch=4.0, yaw=181.5)        # bench.py's camera


def _window_mask(x0, y0, x1, y1):
    m = np.zeros((H, W), bool)
    m[y0:y1, x0:x1] = True
    return m.reshape(-1)


# The workloads bench.py times.  "textured" is its default (albedo / smoothness / normal maps with bump mapping on, float
# emittance maps on the signs), "plain" is --plain (the round-1 constant-colour scene), "cluttered" is --cluttered
# (textured + trees of leaf cards, cables, railings).
WORKLOADS = {"plain": dict(textured=False, cluttered=False, bump=0),
             "textured": dict(textured=True, cluttered=False, bump=1),
             "cluttered": dict(textured=True, cluttered=True, bump=1)}
BASE_INSTANCES = 2745          # instances of the street without the clutter; trees / cables / railings come after them


def _scene(workload):
    w = WORKLOADS[workload]
    return util.bench_street(textured=w["textured"], cluttered=w["cluttered"])


_windows = {}


def _choose_window(workload, hs, size=(80, 40)):
    """An 8-aligned window of the bench frame that shows what the workload adds: for the textured street at least one
    sign (a material with an emittance MAP) and a bump-mapped facade / ground / crate (a material with a normal map); for
    the cluttered street also leaf cards / cables / railings.  Found on the GPU's own primary-hit G-buffer (which the
    test then compares with the oracle inside the window like every other buffer); the plain street keeps the fixed
    window of round 2."""
    if workload == "plain":
        x0, y0 = 900 * W // 1920 // 8 * 8, 560 * H // 1080 // 8 * 8
        return (x0, y0, x0 + size[0], y0 + size[1])
    if (workload, size, W, H) in _windows:
        return _windows[(workload, size, W, H)]
    import torch
    ctx = api.Context(0)
    hs.upload(ctx)
    accel = ctx.accel_build()
    ctx.lights_build_static()
    dev = util.DeviceBuffers(util.PixelBuffers(W, H))
    cam = api.make_camera(W, H, **CAM)
    stream = torch.cuda.current_stream().cuda_stream
    f = util.frame_params(api.GfxRestirFrameParams, api.GfxCamera, W, H, cam, travHandle=accel, frameIndex=0, bufferIndex=0, resetFlowBuffer=1)
    ctx.lights_build_instances(stream)
    ctx.restir_set_params(dev.static_params(), f, 0, 0, stream)
    ctx.restir_launch(api.PASS_SETUP_GBUFFERS, W, H, stream)
    got = dev.download()
    mats = hs.materials()
    sign_mat = np.array([m.texEmittance != 0 for m in mats] + [False])
    bump_mat = np.array([m.texNormal != 0 for m in mats] + [False])
    inst = got["gb0_0"]["instSlot"].reshape(H, W)
    surface = inst != 0xFFFFFFFF
    mat = np.where(surface, got["gb3_0"]["matSlot"].reshape(H, W), len(mats)).astype(np.int64)
    layers = [sign_mat[mat], bump_mat[mat]]
    need = [24, 600]
    if WORKLOADS[workload]["cluttered"]:
        layers.append(surface & (inst >= BASE_INSTANCES))
        need.append(200)
    ww, wh = size
    sums = []
    for layer in layers:                                   # window sums on the 8-pixel grid through an integral image
        ii = np.zeros((H + 1, W + 1), np.int64)
        ii[1:, 1:] = np.cumsum(np.cumsum(layer.astype(np.int64), 0), 1)
        ys, xs = np.arange(0, H - wh + 1, 8), np.arange(0, W - ww + 1, 8)
        sums.append(ii[ys[:, None] + wh, xs[None, :] + ww] - ii[ys[:, None], xs[None, :] + ww] - ii[ys[:, None] + wh, xs[None, :]] + ii[ys[:, None], xs[None, :]])
    ok = np.ones_like(sums[0], bool)
    for ssum, n in zip(sums, need):
        ok &= ssum >= n
    assert ok.any(), "no window of the %s frame shows %s" % (workload, "signs, bump-mapped surfaces" + (", clutter" if len(need) > 2 else ""))
    ys, xs = np.nonzero(ok)
    # the candidate closest to the frame centre, away from the frame border (the reuse margins stay inside the image)
    cy, cx = ys * 8 + wh / 2 - H / 2, xs * 8 + ww / 2 - W / 2
    k = int(np.argmin(cx * cx + cy * cy))
    win = (int(xs[k]) * 8, int(ys[k]) * 8, int(xs[k]) * 8 + ww, int(ys[k]) * 8 + wh)
    _windows[(workload, size, W, H)] = win
    return win


def _assert_window_shows_the_workload(workload, hs, gb0, gb3, window):
    """The oracle's own G-buffer inside the window holds a sign, a bump-mapped surface and (cluttered) clutter."""
    if workload == "plain":
        return
    mats = hs.materials()
    x0, y0, x1, y1 = window
    inst = gb0["instSlot"].reshape(H, W)[y0:y1, x0:x1]
    mat = gb3["matSlot"].reshape(H, W)[y0:y1, x0:x1][inst != 0xFFFFFFFF]
    used = set(int(m) for m in np.unique(mat))
    assert any(mats[m].texEmittance for m in used), "no emittance-mapped sign inside the window"
    assert any(mats[m].texNormal for m in used), "no bump-mapped surface inside the window"
    if WORKLOADS[workload]["cluttered"]:
        assert ((inst != 0xFFFFFFFF) & (inst >= BASE_INSTANCES)).sum() > 100, "no leaf cards / cables / railings inside the window"


def _pick(arr, mask, n):
    a = np.asarray(arr)
    if a.ndim >= 2 and a.shape[0] == 3 and a.shape[1] == n:      # reservoir planes [3][n][4]
        return a[:, mask]
    return a[mask]


@pytest.mark.parametrize("workload", ["plain", "textured", "cluttered"])
@pytest.mark.parametrize("config", ["configs[2]: biased", "configs[4]: unbiased + 2048x1024 environment map"])
def test_window_of_the_full_frame_matches_the_oracle(built_lib, config, workload):
    with util.frame_overrides(enableBumpMapping=WORKLOADS[workload]["bump"]):
        _window_of_the_full_frame(config, workload)


@pytest.mark.parametrize("workload", ["plain", "textured"])
def test_window_of_a_3840x2160_frame_matches_the_oracle(built_lib, monkeypatch, workload):
    """The same check at four times the pixels of BASELINE's configuration (8.3 M pixels: padded launch slots, ray-queue
    sizes, the 8x8 tile / XCD supertile map with a different supertile grid): a window of the 4K frame, bit for bit, after every pass of two frames."""
    import sys
    mod = sys.modules[__name__]
    monkeypatch.setattr(mod, "W", 3840)
    monkeypatch.setattr(mod, "H", 2160)
    with util.frame_overrides(enableBumpMapping=WORKLOADS[workload]["bump"]):
        _window_of_the_full_frame("configs[2]: biased", workload)


# Windows that are NOT where the workload's defining geometry is: the frame's corner and edges (the reuse margins are clipped by the
# image there, launch slots of padded 16 x 16 blocks / XCD supertiles lie next to them) in other supertiles than the chosen window.
def test_every_pixel_of_a_3840x2160_frame_matches_the_oracle(built_lib, monkeypatch):
    """Four times BASELINE's pixels without a window: all 8 294 400 pixels of two frames of the textured street (launch sizes,
    ray-queue capacities and the supertile grid of a 4K frame; the oracle needs ~20 s a frame on 128 threads)."""
    import sys
    mod = sys.modules[__name__]
    monkeypatch.setattr(mod, "W", 3840)
    monkeypatch.setattr(mod, "H", 2160)
    with util.frame_overrides(enableBumpMapping=1):
        _window_of_the_full_frame("configs[2]: biased", "textured", inner=(0, 0, 3840, 2160), frames=2)


BORDER_WINDOWS = {"bottom-left corner": lambda: (0, H - 40, 80, H),
                  "right edge": lambda: (W - 80, 600 * H // 1080 // 8 * 8, W, 600 * H // 1080 // 8 * 8 + 40),
                  "top edge, left of centre": lambda: (640 * W // 1920 // 8 * 8, 0, 640 * W // 1920 // 8 * 8 + 80, 40)}


@pytest.mark.parametrize("where", sorted(BORDER_WINDOWS))
def test_border_window_of_the_full_frame_matches_the_oracle(built_lib, where):
    """A second net at full size (textured workload, configs[2]): windows on the frame's border -- clipped reuse margins, padded launch
    slots -- in other XCD supertiles than the window test_window_of_the_full_frame_matches_the_oracle picks."""
    with util.frame_overrides(enableBumpMapping=1):
        _window_of_the_full_frame("configs[2]: biased", "textured", inner=BORDER_WINDOWS[where](), workload_window=False)


@pytest.mark.parametrize("config,workload", [("configs[2]: biased", "textured"), ("configs[2]: biased", "plain"), ("configs[2]: biased", "cluttered"),
                                             ("configs[4]: unbiased + 2048x1024 environment map", "textured")])
def test_every_pixel_of_the_full_frame_matches_the_oracle(built_lib, config, workload):
    """No window: all 2 073 600 pixels of two frames of the bench workloads (the textured street = the default; --plain;
    --cluttered), every buffer after every pass.  The oracle renders a pass over its host's cores (OpenMP over pixels); on
    the 128-thread GPU box a frame of it takes a few seconds, which is what ReGIR's whole-frame comparison already relies on."""
    with util.frame_overrides(enableBumpMapping=WORKLOADS[workload]["bump"]):
        _window_of_the_full_frame(config, workload, inner=(0, 0, W, H), frames=2)


def test_every_pixel_of_the_animated_full_frame_matches_the_oracle(built_lib):
    """`bench.py --animate` without a window: the moving rectangle light in the animated subtree (rebuilt every frame on both
    sides) and the orbiting camera, three frames, every pixel of every buffer after every pass -- temporal reuse through motion
    vectors, the temporal hint of the traversal and the previous camera, frame-wide."""
    with util.frame_overrides(enableBumpMapping=1):
        _window_of_the_full_frame("configs[2]: biased", "textured", inner=(0, 0, W, H), frames=3, animated=True, workload_window=False)


def _animation(frame):
    """bench.py --animate: the reference command line's moving rectangle light (restir_di_main.cpp:7-12) at 60 frames per second and
    the slowly orbiting camera, as bench.py applies them before frame `frame`."""
    import bench
    return bench.light_transform(api, frame / 60.0), bench.orbit_camera(api, W, H, frame)


def test_animated_window_of_the_full_frame_matches_the_oracle(built_lib):
    """bench.py --animate under the oracle at BASELINE's size: the textured street + the moving rectangle light in the animated BVH
    subtree (rebuilt in place every frame) + the orbiting camera, three frames -- motion vectors, the temporal hint and the temporal
    reuse across the motion, every buffer of a window lit by the moving light after every frame (restir_di_main.cpp:2249-2264)."""
    with util.frame_overrides(enableBumpMapping=1):
        _window_of_the_full_frame("configs[2]: biased", "textured", frames=3, animated=True, workload_window=False)


def _window_lit_by_the_moving_light(hs, light_slot, size=(80, 40)):
    """An 8-aligned window around the pixel that sees the ground under the light's position of frame 1 (found on the GPU's own
    G-buffer of that frame; the test then compares that G-buffer with the oracle like every other buffer)."""
    import torch
    ctx = api.Context(0)
    hs.upload(ctx)
    ctx.instance_set_dynamic(light_slot)
    xfm, cam = _animation(1)
    ctx.instance_set_transform(light_slot, xfm)
    accel = ctx.accel_build()
    ctx.lights_build_static()
    dev = util.DeviceBuffers(util.PixelBuffers(W, H))
    stream = torch.cuda.current_stream().cuda_stream
    f = util.frame_params(api.GfxRestirFrameParams, api.GfxCamera, W, H, cam, travHandle=accel, frameIndex=0, bufferIndex=0, resetFlowBuffer=1)
    ctx.lights_build_instances(stream)
    ctx.restir_set_params(dev.static_params(), f, 0, 0, stream)
    ctx.restir_launch(api.PASS_SETUP_GBUFFERS, W, H, stream)
    got = dev.download()
    pos = np.asarray(got["gb2_0"]).view(np.float32).reshape(H, W, 4)[:, :, :3]
    inst = got["gb0_0"]["instSlot"].reshape(H, W)
    under = np.array([xfm[3], 0.0, xfm[11]], np.float32)           # the light's position projected on the ground plane
    d = np.where(inst != 0xFFFFFFFF, np.linalg.norm(np.nan_to_num(pos) - under, axis=2), 1e30)
    y, x = np.unravel_index(int(np.argmin(d)), d.shape)
    assert d[y, x] < 1.5, "the ground under the moving light is not in view"
    x0 = min(max(0, (x - size[0] // 2) // 8 * 8), W - size[0])
    y0 = min(max(0, (y - size[1] // 2) // 8 * 8), H - size[1])
    ctx.close()
    return (int(x0), int(y0), int(x0) + size[0], int(y0) + size[1])


def _window_of_the_full_frame(config, workload, inner=None, frames=2, animated=False, workload_window=True):
    import torch
    unbiased = "unbiased" in config
    hs = _scene(workload)
    light_slot = None
    if animated:
        light_slot = hs.add_instance(hs.add_rectangle(1.5, 1.5, (60, 60, 60)), _animation(0)[0])
        if inner is None:
            inner = _window_lit_by_the_moving_light(hs, light_slot)
    if inner is None:
        inner = _choose_window(workload, hs)
    ctx = api.Context(0)
    hs.upload(ctx)
    if light_slot is not None:
        ctx.instance_set_dynamic(light_slot)
    accel = ctx.accel_build()
    ctx.lights_build_static()
    if (inner[2] - inner[0]) * (inner[3] - inner[1]) > 256 * 256:
        with util.every_host_thread():              # a whole frame: every host thread (the results do not depend on the count)
            osc = util.feed_oracle(hs)
    else:
        osc = util.feed_oracle(hs)
    cam = api.make_camera(W, H, **CAM)
    ocam = util.copy_struct(O.GfxCamera, cam)
    pb_init, pb_cpu = util.PixelBuffers(W, H), util.PixelBuffers(W, H)
    env_kw = {}
    if unbiased:
        ew, eh = 2048, 1024                            # analytic sky + sun, lat-long
        sky = api.env_make_sky(ew, eh)
        pb_init.set_env(sky, ew, eh)
        pb_cpu.set_env(sky, ew, eh, oracle_side=True)
        env_kw = dict(enableEnvLight=1, envLightPowerCoeff=0.6, envLightRotation=0.4)
    dev = util.DeviceBuffers(pb_init)
    s_gpu, s_cpu = dev.static_params(), pb_cpu.host_static_params()
    stream = torch.cuda.current_stream().cuda_stream
    n = W * H
    radius = 20
    motion = 24 if animated else 0                     # pixels a motion vector may span per frame (bench.py ANIMATE_MAX_MOTION_ROWS; asserted below)
    passes, nb = (1, 3) if unbiased else (2, 5)        # restir_di_main.cpp:1965-1967
    spatial = api.PASS_SPATIAL_UNBIASED if unbiased else api.PASS_SPATIAL_BIASED
    # margins: a frame's final reservoirs are exact `radius * passes` pixels inside the region its first passes
    # covered, and the next frame's temporal pass reads them (`motion` pixels away at most) -- so every frame starts that much wider than the next
    pads = [radius * passes + 8]
    for _ in range(frames - 1):
        pads.insert(0, pads[0] + radius * passes + motion)

    def grow(r, d):
        return (max(0, r[0] - d), max(0, r[1] - d), min(W, r[2] + d), min(H, r[3] + d))

    diffs = []
    last_res, last_base = 1, 0
    prev_cam = prev_ocam = None
    for frame in range(frames):
        if animated:
            # InstanceController::update + updateASs (restir_di_main.cpp:2258-2264), then the camera
            xfm, cam = _animation(frame)
            ocam = util.copy_struct(O.GfxCamera, cam)
            ctx.instance_set_transform(light_slot, xfm)
            osc.set_instance_transform(light_slot, xfm)
            assert ctx.accel_build(handle=accel) == accel
            osc.commit()
        kw = dict(frameIndex=frame, bufferIndex=frame % 2, resetFlowBuffer=int(frame == 0), numAccumFrames=0,
                  numSpatialNeighbors=nb, useUnbiasedEstimator=int(unbiased), useLowDiscrepancyNeighbors=1, reuseVisibility=1, **env_kw)
        f_gpu = util.frame_params(api.GfxRestirFrameParams, api.GfxCamera, W, H, cam, prev_cam=prev_cam, travHandle=accel, **kw)
        f_cpu = util.frame_params(O.GfxRestirFrameParams, O.GfxCamera, W, H, ocam, prev_cam=prev_ocam, travHandle=0, **kw)
        if animated:
            prev_cam, prev_ocam = cam, ocam
        ctx.lights_build_instances(stream)
        cur = (last_res + 1) % 2
        pad = pads[frame]
        entry = api.PASS_INITIAL_RIS
        if frame > 0:
            entry = api.PASS_INITIAL_TEMPORAL_UNBIASED if unbiased else api.PASS_INITIAL_TEMPORAL_BIASED
        plan = [(api.PASS_SETUP_GBUFFERS, 0, pad), (entry, 0, pad)]
        plan += [(spatial, nb * i, pad - radius * (i + 1)) for i in range(passes)]
        plan += [(api.PASS_SHADING, nb * passes, pad - radius * passes)]
        for pass_id, base_off, margin in plan:
            base = last_base + base_off
            ctx.restir_set_params(s_gpu, f_gpu, cur, base, stream)
            ctx.restir_launch(pass_id, W, H, stream)
            osc.restir_launch(s_cpu, f_cpu, cur, base, pass_id, rect=grow(inner, margin))
            if pass_id == spatial:
                cur = (cur + 1) % 2
        last_base += nb * passes
        last_res = cur
        got, want = dev.download(), pb_cpu.arrays()
        mask = _window_mask(*inner)
        for key in ("rng", "beauty", "albedo", "normal", f"gb0_{frame % 2}", f"gb1_{frame % 2}", f"gb2_{frame % 2}", f"gb3_{frame % 2}",
                    f"res_{cur}", f"info_{cur}"):
            a = np.ascontiguousarray(_pick(got[key].reshape(want[key].shape), mask, n)).view(np.uint8)
            b = np.ascontiguousarray(_pick(want[key], mask, n)).view(np.uint8)
            if not np.array_equal(a, b):
                diffs.append(f"frame {frame}: {key}: {np.count_nonzero(a != b)} bytes differ inside the window")
        if animated and frame > 0 and inner != (0, 0, W, H):
            # the margins assume |motion vector| <= `motion` inside the region the oracle rendered; and the window does see motion
            big = grow(inner, pads[frame])
            mv = np.nan_to_num(np.asarray(pb_cpu.gb1[frame % 2]).view(np.float32).reshape(H, W, 2)[big[1]:big[3], big[0]:big[2]])
            assert np.abs(mv).max() <= motion, f"frame {frame}: a motion vector of {np.abs(mv).max():.1f} pixels exceeds the margin"
            assert np.abs(mv).max() > 0.25, "nothing moves inside the window"
    assert not diffs, "\n".join(diffs)
    beauty = pb_cpu.beauty.reshape(H, W, 4)[inner[1]:inner[3], inner[0]:inner[2], :3]
    assert np.isfinite(beauty).all()
    if workload_window:
        assert beauty.mean() > 1e-4                                  # the window is lit, not background
        _assert_window_shows_the_workload(workload, hs, pb_cpu.gb0[(frames - 1) % 2], pb_cpu.gb3[(frames - 1) % 2], inner)
    if animated and inner != (0, 0, W, H):
        # the moving light lights the window (60 W/m2 sr over 1.5 x 1.5 m, a few metres above the ground)
        assert beauty.mean() > 1e-2, "the window is not lit by the moving light"
        gb0 = pb_cpu.gb0[(frames - 1) % 2]["instSlot"].reshape(H, W)
        assert (gb0[inner[1]:inner[3], inner[0]:inner[2]] != 0xFFFFFFFF).mean() > 0.5
    ctx.close()


def _render(frames, serial=False, band=None, monkeypatch=None):
    import torch
    ctx = api.Context(0)
    util.bench_street().upload(ctx)
    cfg = api.RestirRenderer.default_config(W, H, api.RENDERER_BIASED)
    cfg.camera = api.make_camera(W, H, **CAM)
    if band is not None:
        cfg.rowBegin, cfg.rowEnd = band
    if monkeypatch is not None:
        monkeypatch.setenv("GFX_SERIAL_FRAMES", "1" if serial else "0")
    r = api.RestirRenderer(ctx, cfg)
    for _ in range(frames):
        r.render_frame()
    torch.cuda.synchronize()
    out = ctx.read_device(r.beauty_ptr(), W * H * 16).view(np.float32).reshape(-1, 4).copy()
    return out, ctx, r


def test_full_size_frames_are_deterministic_and_pipelining_changes_nothing(built_lib, monkeypatch):
    a, ctx_a, ra = _render(3, serial=False, monkeypatch=monkeypatch)
    b, ctx_b, rb = _render(3, serial=False, monkeypatch=monkeypatch)
    util.assert_same_bits("run-to-run", a, b)
    del ctx_b, rb
    c, ctx_c, rc = _render(3, serial=True, monkeypatch=monkeypatch)
    util.assert_same_bits("pipelined vs serial frame loop", a, c)
    assert np.isfinite(a).all() and a[:, :3].mean() > 1e-3


@pytest.mark.parametrize("workload", ["plain", "textured"])
def test_path_tracer_window_of_the_full_frame_matches_the_oracle(built_lib, workload):
    """Baseline path tracer (max path length 5) on the bench scene at 1920x1080: paths never read a neighbour's
    state, so the oracle's window needs no margin; two frames with accumulation."""
    with util.frame_overrides(enableBumpMapping=WORKLOADS[workload]["bump"]):
        _path_tracer_window(workload)


def _path_tracer_window(workload):
    import torch
    hs = _scene(workload)
    window = _choose_window(workload, hs, (128, 64)) if workload != "plain" else (880, 540, 1008, 604)
    ctx = api.Context(0)
    hs.upload(ctx)
    accel = ctx.accel_build()
    ctx.lights_build_static()
    osc = util.feed_oracle(hs)
    cam = api.make_camera(W, H, **CAM)
    ocam = util.copy_struct(O.GfxCamera, cam)
    pb_init, pb_cpu = util.PixelBuffers(W, H), util.PixelBuffers(W, H)
    dev = util.DeviceBuffers(pb_init)
    s_gpu, s_cpu = dev.static_params(), pb_cpu.host_static_params()
    stream = torch.cuda.current_stream().cuda_stream
    mask = _window_mask(*window)
    diffs = []
    for frame in range(2):
        kw = dict(frameIndex=frame, bufferIndex=frame % 2, resetFlowBuffer=int(frame == 0), numAccumFrames=frame)
        f_gpu = util.frame_params(api.GfxRestirFrameParams, api.GfxCamera, W, H, cam, travHandle=accel, **kw)
        f_cpu = util.frame_params(O.GfxRestirFrameParams, O.GfxCamera, W, H, ocam, travHandle=0, **kw)
        ctx.lights_build_instances(stream)
        ctx.restir_set_params(s_gpu, f_gpu, 0, 0, stream)
        for pass_id in (api.PT_SETUP_GBUFFERS, api.PT_PATH_TRACE_BASELINE):
            ctx.pt_launch(pass_id, W, H, 5, 0, 0, stream)
            osc.pt_launch(s_cpu, f_cpu, pass_id, 5, rect=window)
        got, want = dev.download(), pb_cpu.arrays()
        for key in ("rng", "beauty", "albedo", "normal", f"gb0_{frame % 2}", f"gb1_{frame % 2}"):
            a = np.ascontiguousarray(_pick(got[key].reshape(want[key].shape), mask, W * H)).view(np.uint8)
            b = np.ascontiguousarray(_pick(want[key], mask, W * H)).view(np.uint8)
            if not np.array_equal(a, b):
                diffs.append(f"frame {frame}: {key}: {np.count_nonzero(a != b)} bytes differ inside the window")
    assert not diffs, "\n".join(diffs)
    beauty = pb_cpu.beauty.reshape(H, W, 4)[window[1]:window[3], window[0]:window[2], :3]
    assert np.isfinite(beauty).all() and beauty.mean() > 1e-4
    _assert_window_shows_the_workload(workload, hs, pb_cpu.gb0[1], pb_cpu.gb3[1], window)


@pytest.mark.parametrize("workload", ["plain", "textured"])
@pytest.mark.parametrize("unbiased", [False, True])
def test_rearchitected_window_of_the_full_frame_matches_the_oracle(built_lib, unbiased, workload):
    with util.frame_overrides(enableBumpMapping=WORKLOADS[workload]["bump"]):
        _rearchitected_window(unbiased, workload)


def _rearchitected_window(unbiased, workload, whole=False):
    """Rearchitected ReSTIR at 1920x1080 on the bench scene: the 131072 pre-sampled lights are compared in full, the
    per-pixel passes on an 8-aligned window plus a 48-pixel margin (frame 1's temporal and spatiotemporal
    neighbours lie within 20 pixels of a pixel; frame 0 reads no neighbour) -- or, `whole`, on every pixel."""
    import torch
    hs = _scene(workload)
    if whole:
        inner = (48, 48, W - 48, H - 48)               # region = the frame; `inner` is widened to it below
    else:
        inner = _choose_window(workload, hs) if workload != "plain" else (896, 560, 976, 600)
    ctx = api.Context(0)
    hs.upload(ctx)
    accel = ctx.accel_build()
    ctx.lights_build_static()
    osc = util.feed_oracle(hs)
    cam = api.make_camera(W, H, **CAM)
    ocam = util.copy_struct(O.GfxCamera, cam)
    pb_init, pb_cpu = util.PixelBuffers(W, H), util.PixelBuffers(W, H)
    dev = util.DeviceBuffers(pb_init)
    s_gpu, s_cpu = dev.static_params(), pb_cpu.host_static_params()
    stream = torch.cuda.current_stream().cuda_stream
    region = (inner[0] - 48, inner[1] - 48, inner[2] + 48, inner[3] + 48)
    if whole:
        inner = region
    mask = _window_mask(*inner)
    n = W * H
    diffs = []
    last_res, last_base = 1, 0
    for frame in range(2):
        kw = dict(frameIndex=frame, bufferIndex=frame % 2, resetFlowBuffer=int(frame == 0), numAccumFrames=0,
                  numSpatialNeighbors=1, enableTemporalReuse=1, enableSpatialReuse=1, useUnbiasedEstimator=int(unbiased),
                  useLowDiscrepancyNeighbors=1, reuseVisibilityForSpatiotemporal=0)
        f_gpu = util.frame_params(api.GfxRestirFrameParams, api.GfxCamera, W, H, cam, travHandle=accel, **kw)
        f_cpu = util.frame_params(O.GfxRestirFrameParams, O.GfxCamera, W, H, ocam, travHandle=0, **kw)
        ctx.lights_build_instances(stream)
        cur = (last_res + 1) % 2
        trace_pass, shade_pass = api.rearch_passes(True, True, unbiased, frame == 0)
        ctx.restir_set_params(s_gpu, f_gpu, cur, last_base, stream)
        for pass_id in (api.PASS_SETUP_GBUFFERS, api.PASS_LIGHT_PRESAMPLING, api.PASS_PER_PIXEL_RIS, trace_pass, shade_pass):
            ctx.restir_launch(pass_id, W, H, stream)
            osc.restir_launch(s_cpu, f_cpu, cur, last_base, pass_id, rect=None if pass_id == api.PASS_LIGHT_PRESAMPLING else region)
        last_base += 1
        last_res = cur
        got, want = dev.download(), pb_cpu.arrays()
        for key in ("presample_rngs", "presampled"):
            if not np.array_equal(np.ascontiguousarray(got[key]).view(np.uint8).reshape(-1), np.ascontiguousarray(want[key]).view(np.uint8).reshape(-1)):
                diffs.append(f"frame {frame}: {key} differs")
        for key in ("rng", "beauty", f"gb0_{frame % 2}", f"gb2_{frame % 2}", f"res_{cur}", f"info_{cur}", f"vis_{frame % 2}"):
            a = np.ascontiguousarray(_pick(got[key].reshape(want[key].shape), mask, n)).view(np.uint8)
            b = np.ascontiguousarray(_pick(want[key], mask, n)).view(np.uint8)
            if not np.array_equal(a, b):
                diffs.append(f"frame {frame}: {key}: {np.count_nonzero(a != b)} bytes differ inside the window")
    assert not diffs, "\n".join(diffs)


@pytest.mark.parametrize("unbiased", [False, True])
def test_every_pixel_of_the_rearchitected_full_frame_matches_the_oracle(built_lib, unbiased):
    """The rearchitected set without a window: all 2 073 600 pixels of two frames of the textured street."""
    with util.frame_overrides(enableBumpMapping=1), util.every_host_thread():
        _rearchitected_window(unbiased, "textured", whole=True)


def test_every_pixel_of_the_path_traced_full_frame_matches_the_oracle(built_lib):
    """The path tracer (max path length 5, jittered) without a window: every pixel of two accumulated frames of the textured
    street at 1920x1080 -- the big-launch path of k_pt_fused / the wavefront queues."""
    from tests.test_gpu_pathtrace import run_pt_both
    with util.frame_overrides(enableBumpMapping=1), util.every_host_thread():
        diffs = run_pt_both(_scene("textured"), W, H, frames=2, max_len=5, jitter=1, camera=api.make_camera(W, H, **CAM))
    assert not diffs, "\n".join(diffs[:16])


def test_every_pixel_of_the_nrc_full_frame_matches_the_oracle(built_lib):
    """BASELINE configs[3]'s render side without a window: tile / training-path selection, the NRC path tracer, radiance
    queries, terminal infos and the training chains (canonical form: the record indices come out of an unordered atomic counter)
    of two frames of the textured street at 1920x1080, frame 1's adaptive tile size from each side's own frame-0 record count."""
    from tests.test_gpu_nrc_render import run_nrc_both
    with util.frame_overrides(enableBumpMapping=1), util.every_host_thread():
        diffs = run_nrc_both(_scene("textured"), W, H, frames=2, max_len=5, camera=api.make_camera(W, H, **CAM))
    assert not diffs, "\n".join(diffs[:16])


@pytest.mark.parametrize("workload", ["plain", "textured"])
def test_nrc_window_of_the_full_frame_matches_the_oracle(built_lib, workload):
    with util.frame_overrides(enableBumpMapping=WORKLOADS[workload]["bump"]):
        _nrc_window(workload)


def _nrc_window(workload):
    """BASELINE configs[3] at its full size: the NRC path tracer (tile / training-path selection, radiance queries,
    terminal infos, per-frame contribution) at 1920x1080 on the bench scene; the oracle renders a 64 x 40 window (whole
    8 x 8 tiles of frame 0) and every per-pixel buffer of that window is bit-equal over two frames.  The training-record
    indices come from one frame-wide atomic counter and are not comparable for a window: frame 1's adaptive tile size is
    derived from the GPU's own frame-0 record count on both sides."""
    import torch
    from tests.test_gpu_nrc_render import _compare_exact
    hs = _scene(workload)
    window = _choose_window(workload, hs, (64, 40)) if workload != "plain" else (896, 560, 960, 600)
    ctx = api.Context(0)
    hs.upload(ctx)
    accel = ctx.accel_build()
    ctx.lights_build_static()
    osc = util.feed_oracle(hs)
    cam = api.make_camera(W, H, **CAM)
    ocam = util.copy_struct(O.GfxCamera, cam)
    pb_init, pb_cpu = util.PixelBuffers(W, H), util.PixelBuffers(W, H)
    dev = util.DeviceBuffers(pb_init)
    nb_gpu, nb_cpu = util.NrcBuffers(W, H, hs.bounds()), util.NrcBuffers(W, H, hs.bounds())
    nb_gpu.to_device()
    s_gpu, s_cpu = dev.static_params(), pb_cpu.host_static_params()
    stream = torch.cuda.current_stream().cuda_stream
    mask = _window_mask(*window)
    n = W * H
    offsets = np.random.default_rng(99)
    diffs = []
    for frame in range(2):
        b = frame % 2
        kw = dict(frameIndex=frame, bufferIndex=b, resetFlowBuffer=int(frame == 0), numAccumFrames=frame)
        f_gpu = util.frame_params(api.GfxRestirFrameParams, api.GfxCamera, W, H, cam, travHandle=accel, **kw)
        f_cpu = util.frame_params(O.GfxRestirFrameParams, O.GfxCamera, W, H, ocam, travHandle=0, **kw)
        ou, ot = int(offsets.integers(0, 1 << 31)), int(offsets.integers(0, 1 << 31))
        if frame > 0:   # the oracle only traced a window: give it the frame-wide state the preprocess pass reads
            g = nb_gpu.download()
            for k in ("nrc_num_0", "nrc_num_1", "nrc_tile_0", "nrc_tile_1"):
                nb_cpu.a[k][:] = g[k]
        ctx.lights_build_instances(stream)
        ctx.restir_set_params(s_gpu, f_gpu, 0, 0, stream)
        ctx.nrc_set_render_params(nb_gpu.device_params(ou, ot, frame == 0))
        osc.nrc_set_render_params(nb_cpu.host_params(ou, ot, frame == 0))
        for pass_id in (api.PT_SETUP_GBUFFERS, api.PT_NRC_PREPROCESS, api.PT_PATH_TRACE_NRC):
            ctx.pt_launch(pass_id, W, H, 5, 0, 0, stream)
            osc.pt_launch(s_cpu, f_cpu, pass_id, 5, rect=window)
        got, want = dev.download(), pb_cpu.arrays()
        got.update(nb_gpu.download()); want.update(nb_cpu.arrays())
        _compare_exact(diffs, f"frame {frame}", got, want, [f"nrc_tile_{b}", "nrc_off_unbiased", "nrc_off_training"])
        for key in ("rng", f"gb0_{b}", "nrc_contribution", "nrc_terminal"):
            a = np.ascontiguousarray(_pick(got[key].reshape(want[key].shape), mask, n)).view(np.uint8)
            c = np.ascontiguousarray(_pick(want[key], mask, n)).view(np.uint8)
            if not np.array_equal(a, c):
                diffs.append(f"frame {frame}: {key}: {np.count_nonzero(a != c)} bytes differ inside the window")
        hq = ((want["nrc_terminal"][:, 3].view(np.uint32) & 1) == 1) & mask
        assert hq.sum() > 0.2 * mask.sum()          # most paths of the window end in the cache
        if not np.array_equal(got["nrc_queries"][:n][hq].view(np.uint32), want["nrc_queries"][:n][hq].view(np.uint32)):
            diffs.append(f"frame {frame}: rendering-path queries differ inside the window")
        assert int(got[f"nrc_num_{b}"][0]) > 10000   # the full frame produced training records
    assert not diffs, "\n".join(diffs)


@pytest.mark.parametrize("config", ["configs[2]: biased", "configs[4]: unbiased + 2048x1024 environment map"])
def test_eight_band_strip_exchange_at_the_bench_configuration(built_lib, config):
    """bench.py --gpus 8 on one GPU: eight band renderers (7 x 136 + 128 rows of the 1920x1080 frame, radius-20 strips, the
    textured bench scene) driven by eight host threads through the loop-back transport reproduce the whole-frame renderer
    bit for bit over three frames -- the same descriptors tilesplit.StripExchange sends over RCCL.  configs[4] is the
    workload BASELINE.json names for the 8-GPU split: the unbiased estimator (1 x 3 neighbours with MIS rays) under a
    2048 x 1024 environment map."""
    import torch
    from gfxexp_amd import scenes, tilesplit
    from tests import loopback
    hs = scenes.bench_street(textured=True)
    world, frames = 8, 3
    unbiased = "unbiased" in config
    sky = api.env_make_sky(2048, 1024) if unbiased else None

    def make(band):
        ctx = api.Context(0)
        hs.upload(ctx)
        cfg = api.RestirRenderer.default_config(W, H, api.RENDERER_UNBIASED if unbiased else api.RENDERER_BIASED)
        cfg.camera = api.make_camera(W, H, **CAM)
        cfg.enableBumpMapping = 1
        cfg.rowBegin, cfg.rowEnd = band
        r = api.RestirRenderer(ctx, cfg)
        if unbiased:
            r.set_env(sky, 2048, 1024, power_coeff=0.6, rotation=0.4)
        return ctx, r

    ctx_full, full = make((0, 0))
    for _ in range(frames):
        full.render_frame()
    torch.cuda.synchronize()
    want = ctx_full.read_device(full.beauty_ptr(), W * H * 16).view(np.float32).reshape(H, W, 4).copy()
    del full, ctx_full
    bands = tilesplit.band_rows(H, world)
    assert bands[0] == (0, 136) and bands[-1] == (952, 1080)
    made = [make(b) for b in bands]
    ex = loopback.LoopbackExchange(world, timeout=300.0)
    for rank, (_, r) in enumerate(made):
        r.set_exchange(ex.callback(rank), 0)
    loopback.run_bands([r for _, r in made], frames)
    for rank, (ctx, r) in enumerate(made):
        got = ctx.read_device(r.beauty_ptr(), W * H * 16).view(np.float32).reshape(H, W, 4)
        util.assert_same_bits(f"band {rank} gathered HDR frame", got, want)
    assert np.isfinite(want).all() and want[..., :3].mean() > 1e-3
    kinds = [k for k, _ in ex.calls[0]]
    # G-buffer strips + ONE reservoir exchange per frame: the unbiased estimator has one spatial pass, the biased one recomputes its first on the halo (stripMode 3)
    assert kinds.count(api.EXCHANGE_STRIPS) == frames * 2 and kinds.count(api.EXCHANGE_GATHER_BANDS) == frames


def test_regir_at_the_reference_grid_size_matches_the_oracle(built_lib):
    """ReGIR at the reference's own scale (regir/regir_main.cpp:1112, regir_shared.h:7: 32 x 8 x 32 cells x 512 light slots =
    4 194 304 slot reservoirs, 2^3 candidates per slot, 2^2 per cell, cell randomisation on) on the textured bench street at
    1920x1080, two frames with temporal slot reuse, max path length 5.  Nothing is windowed: the oracle builds every slot and
    path-traces every pixel (128 host threads), and after every pass of both frames ALL 4.19 M slot reservoirs and infos, the slot
    RNGs, the per-cell access counters, the last-access frames, the active-cell counts, the pixel RNGs and the whole beauty /
    albedo / normal buffers are compared bit for bit -- the big-launch paths of k_regir_build (one thread per slot, per-wave merge
    of the cell counters) that the small-grid cases of test_gpu_regir.py never reach."""
    from tests.test_gpu_regir import run_regir_both
    hs = _scene("textured")
    with util.frame_overrides(enableBumpMapping=1):
        diffs = run_regir_both(hs, W, H, frames=2, max_len=5, temporal=True, dims=(32, 8, 32), randomize=1,
                               camera=api.make_camera(W, H, **CAM), log2_slot=3, log2_cell=2)
    assert not diffs, "\n".join(diffs[:16])
