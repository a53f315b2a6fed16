"""-m gpu: every renderer on a scene built to break things (tests/util.py pathological_light_scene):
emitter importance over ~12 decades, runs of non-emissive instances, groups listing coplanar geometry
out of slot order (exact closest-hit ties), emissive geometry first / middle / last in a group.
All buffers bit for bit against the oracle after every pass."""
import os

import numpy as np
import pytest

from gfxexp_amd import api
from tests import util
from tests.test_gpu_nrc_render import run_nrc_both
from tests.test_gpu_pathtrace import run_pt_both
from tests.test_gpu_regir import run_regir_both
from tests.test_gpu_restir import run_sequence_both

pytestmark = pytest.mark.gpu
W, H = 48, 32


def _cam():
    return api.make_camera(W, H, pos=(0.0, 9.0, 38.0), pitch=10.0, yaw=180.0)


@pytest.mark.parametrize("renderer", [api.RENDERER_BIASED, api.RENDERER_UNBIASED])
def test_original_restir(built_lib, renderer):
    diffs = run_sequence_both(util.pathological_light_scene(), W, H, frames=3, renderer=renderer, camera=_cam())
    assert not diffs, "\n".join(diffs[:12])


def test_path_tracer(built_lib):
    diffs = run_pt_both(util.pathological_light_scene(), W, H, frames=2, max_len=6, camera=_cam())
    assert not diffs, "\n".join(diffs[:12])


def test_regir(built_lib):
    diffs = run_regir_both(util.pathological_light_scene(), W, H, frames=3, max_len=4, camera=_cam())
    assert not diffs, "\n".join(diffs[:12])


def test_nrc_render(built_lib):
    diffs = run_nrc_both(util.pathological_light_scene(), W, H, frames=2, max_len=5, camera=_cam())
    assert not diffs, "\n".join(diffs[:12])


def test_closest_hit_ties_follow_the_flattened_order(built_lib):
    from tests.test_gpu_trace import _compare_closest, _gpu_trace, _tri_ids
    hs = util.pathological_light_scene()
    ctx = api.Context(0)
    hs.upload(ctx)
    accel = ctx.accel_build()
    org, dirs = util.pinhole_rays(160, 96, (0.0, 9.0, 38.0), (0.0, 3.0, 0.0), 50.0)
    gpu = _gpu_trace(ctx, accel, api.TRACE_CLOSEST, org, dirs)
    ids = _tri_ids(ctx, accel)
    for brute in (False, True):
        osc = util.feed_oracle(hs, brute_force=brute)
        _compare_closest(gpu, ids, osc.trace(api.TRACE_CLOSEST, org, dirs), osc.tri_ids(), f"brute={brute}")


def _odd_transform_scene():
    """Non-uniformly scaled, sheared and mirrored (negative determinant) instances of emitters and receivers."""
    rng = np.random.default_rng(5)
    s = util.bunny_scene(with_light=False)
    lights = [s.add_rectangle(1.0, 0.6, e) for e in ((40, 30, 20), (5, 10, 30))]
    for k in range(10):
        m = np.eye(3) + rng.uniform(-0.45, 0.45, (3, 3))
        m *= rng.uniform(0.4, 2.5, 3)[None, :]
        if k % 2:
            m[:, 0] = -m[:, 0]                      # mirror
        x = np.zeros((3, 4), np.float32)
        x[:, :3] = m
        x[:, 3] = (rng.uniform(-8, 8), rng.uniform(4, 11), rng.uniform(-6, 8))
        s.add_instance(lights[k % 2], x.reshape(12))
    g = s.load_obj(__import__("os").path.join(util.ASSETS, "teapot.obj"))
    x = np.zeros((3, 4), np.float32)
    x[:, :3] = np.array([[0.9, 0.3, 0.0], [0.0, -1.4, 0.2], [0.1, 0.0, 0.6]])
    x[:, 3] = (5.0, 3.0, 2.0)
    s.add_instance(g, x.reshape(12))
    return s


def test_sheared_and_mirrored_instances(built_lib):
    cam = api.make_camera(W, H, pos=(1.5, 6.0, 18.0), pitch=12.0, yaw=186.0)
    diffs = run_sequence_both(_odd_transform_scene(), W, H, frames=2, renderer=api.RENDERER_UNBIASED, camera=cam)
    diffs += run_pt_both(_odd_transform_scene(), W, H, frames=1, max_len=5, camera=cam)
    assert not diffs, "\n".join(diffs[:12])


def _all_bsdf_scene(seed=21):
    """Lambert, DiffuseAndSpecular and SimplePBR receivers (rough, mirror-like, metallic) under two lights: every
    BSDF type's evaluate / sampleThroughput / evaluatePDF / albedo estimate runs on the GPU."""
    rng = np.random.default_rng(seed)
    s = api.HostScene()

    def material(kind, a, b, smoothness=0.0):
        m = api.GfxMaterial()
        m.bsdfType = kind
        m.a[:] = a
        m.b[:] = b
        m.smoothness = smoothness
        return s.add_material(m)

    mats = [material(0, (0.7, 0.6, 0.5), (0, 0, 0)),                        # Lambert
            material(1, (0.5, 0.1, 0.1), (0.04, 0.04, 0.04), 0.3),          # DiffuseAndSpecular, rough
            material(1, (0.05, 0.05, 0.05), (0.9, 0.8, 0.3), 0.995),        # ... almost a mirror (smoothness clamps at 0.999)
            material(2, (0.8, 0.8, 0.9), (1.0, 0.4, 0.0)),                  # SimplePBR: (occlusion, roughness, metallic) dielectric
            material(2, (0.95, 0.7, 0.3), (1.0, 0.15, 1.0)),                # SimplePBR metal, smooth
            material(2, (0.3, 0.9, 0.4), (1.0, 1.0, 0.5))]                  # SimplePBR, fully rough, half metallic
    obj = os.path.join(util.ASSETS, "stanford_bunny_309_faces.obj")
    ground = np.zeros(4, api.VERTEX_DTYPE)
    ground["position"] = [(-15, 0, -15), (15, 0, -15), (15, 0, 15), (-15, 0, 15)]
    ground["normal"] = (0, 1, 0); ground["texCoord0Dir"] = (1, 0, 0); ground["texCoord"] = [(0, 0), (1, 0), (1, 1), (0, 1)]
    s.add_instance(s.add_group([s.add_geom(ground, [(0, 2, 1), (0, 3, 2)], mats[0])]), api.make_transform())
    bunny = s.load_obj(obj)                                                 # its own material (DiffuseAndSpecular)
    s.add_instance(bunny, api.make_transform(scale=0.08, pos=(-5.0, 0.0, 0.0)))
    # slabs of every material, tilted at random
    slab = np.zeros(4, api.VERTEX_DTYPE)
    slab["position"] = [(-1, 0, -1), (1, 0, -1), (1, 0, 1), (-1, 0, 1)]
    slab["normal"] = (0, 1, 0); slab["texCoord0Dir"] = (1, 0, 0); slab["texCoord"] = [(0, 0), (1, 0), (1, 1), (0, 1)]
    for k, m in enumerate(mats):
        g = s.add_group([s.add_geom(slab, [(0, 2, 1), (0, 3, 2)], m)])
        for j in range(2):
            s.add_instance(g, api.make_transform(scale=float(rng.uniform(1.0, 2.2)), pitch=float(rng.uniform(-50, 50)), roll=float(rng.uniform(-40, 40)),
                                                 yaw=float(rng.uniform(0, 360)), pos=(-6.0 + 2.6 * k + 0.8 * j, float(rng.uniform(0.6, 3.0)), float(rng.uniform(-4, 4)))))
    s.add_instance(s.add_rectangle(1.5, 1.5, (60, 55, 50)), api.make_transform(pos=(0.0, 9.0, 1.0)))
    s.add_instance(s.add_rectangle(1.0, 2.0, (5, 15, 40)), api.make_transform(pitch=-70.0, pos=(5.0, 5.0, 7.0)))
    return s


def test_every_bsdf_type(built_lib):
    cam = api.make_camera(W, H, pos=(0.5, 6.0, 17.0), pitch=14.0, yaw=182.0)
    diffs = run_sequence_both(_all_bsdf_scene(), W, H, frames=2, renderer=api.RENDERER_UNBIASED, camera=cam)
    diffs += run_pt_both(_all_bsdf_scene(), W, H, frames=2, max_len=7, camera=cam)
    diffs += run_regir_both(_all_bsdf_scene(), W, H, frames=2, max_len=4, camera=cam)
    diffs += run_nrc_both(_all_bsdf_scene(), W, H, frames=1, max_len=5, camera=cam)
    assert not diffs, "\n".join(diffs[:12])


def _smooth_emitter_scene():
    """A smooth-shaded emissive mesh (the teapot with every triangle an emitter: three different vertex normals per record,
    the EmitterRecExtra path of light_fetch) next to a flat one, lighting a ground plane and a bunny."""
    s = util.bunny_scene(with_light=False)
    pot = util.teapot_scene(emissive=True)
    mat = s.add_material_traditional((0.01, 0.01, 0.01), (0, 0, 0), 0.3, (4.0, 3.0, 2.0))
    geoms = [s.add_geom(v, t, mat) for v, t, _ in pot.geoms()]
    x = np.zeros((3, 4), np.float32)
    x[:, :3] = np.array([[0.05, 0.01, 0.0], [0.0, 0.04, -0.01], [0.005, 0.0, 0.06]])     # sheared: the normal matrix matters
    x[:, 3] = (-3.0, 5.0, 3.0)
    s.add_instance(s.add_group(geoms), x.reshape(12))
    s.add_instance(s.add_rectangle(1.0, 1.0, (30, 30, 30)), api.make_transform(pos=(4.0, 8.0, 1.0)))
    return s


def test_smooth_shaded_emitters(built_lib):
    hs = _smooth_emitter_scene()
    vn = hs.geoms()[-2][0]["normal"]           # a teapot geometry: its vertex normals really differ
    assert len(np.unique(vn.round(4), axis=0)) > 100
    cam = api.make_camera(W, H, pos=(1.5, 6.0, 18.0), pitch=12.0, yaw=186.0)
    diffs = run_sequence_both(_smooth_emitter_scene(), W, H, frames=2, renderer=api.RENDERER_BIASED, camera=cam)
    diffs += run_pt_both(_smooth_emitter_scene(), W, H, frames=1, max_len=4, camera=cam)
    with util.frame_overrides(useSolidAngleSampling=1):
        diffs += run_pt_both(_smooth_emitter_scene(), W, H, frames=1, max_len=3, camera=cam)
    diffs += run_regir_both(_smooth_emitter_scene(), W, H, frames=2, max_len=3, camera=cam)
    assert not diffs, "\n".join(diffs[:12])
