"""-m gpu: the texture path through the C ABI against the oracle, bit for bit.
(1) the software tex2DLod / tex2Dgather of the kernels (gfx_texture_sample runs the shading kernels' own functions)
    for every texel format, wrap cases and texel-edge / texel-centre coordinates;
(2) every renderer on the textured street (albedo + smoothness + normal maps on ground, facades and crates, float
    emittance maps on the signs: setupBSDFBody's three reads, emittance reads of sampleLight / shading /
    computeTriangleImportance), with bump mapping on (readModifiedNormalFromNormalMap + applyBumpMapping) and off;
(3) the two other bump-map kinds (two-channel normal map, height map via tex2Dgather) and a left-handed normal map."""
import numpy as np
import pytest

from gfxexp_amd import api
from oracle import oracle as O
from tests import util
from tests.test_gpu_nrc_render import run_nrc_both
from tests.test_gpu_pathtrace import run_pt_both
from tests.test_gpu_regir import run_regir_both
from tests.test_gpu_restir import run_sequence_both
from tests.test_gpu_restir_rearch import run_rearch_both
from tests.test_oracle_textures import _random_texture

pytestmark = pytest.mark.gpu
F = np.float32


@pytest.mark.parametrize("fmt", [api.TEX_RGBA8_SRGB, api.TEX_RGBA8_UNORM, api.TEX_R8_UNORM, api.TEX_RG8_UNORM, api.TEX_RGBA32F])
def test_sampler_bit_exact(built_lib, fmt):
    import torch
    rng = np.random.default_rng(100 + fmt)
    ctx = api.Context(0)
    osc = O.OracleScene()
    sizes = [(1, 1), (7, 5), (64, 32), (3, 129)]
    for slot, (w, h) in enumerate(sizes, 1):
        tex = _random_texture(rng, fmt, w, h)
        ctx.texture_set(slot, tex, fmt)
        osc.set_texture(slot, w, h, fmt, tex)
    for slot, (w, h) in enumerate(sizes, 1):
        uv = np.concatenate([rng.random((20000, 2)) * 8 - 4,
                             rng.integers(-8, 9, (1000, 2)) / np.array([w, h]),
                             (rng.integers(-8, 9, (1000, 2)) + 0.5) / np.array([w, h]),
                             np.array([[0.0, 0.0], [1.0, 1.0], [-1e-9, 1 - 1e-9], [1e30, -1e30]])]).astype(F)
        d_uv = torch.from_numpy(uv).cuda()
        d_out = torch.zeros((len(uv), 4), dtype=torch.float32, device="cuda")
        for gather in (False, True):
            ctx.texture_sample(slot, d_uv.data_ptr(), len(uv), d_out.data_ptr(), gather)
            torch.cuda.synchronize()
            util.assert_same_bits(f"fmt {fmt} {w}x{h} gather={gather}", d_out.cpu().numpy(), osc.texture_sample(slot, uv, gather))


def _street():
    return util.small_street(textured=True)


@pytest.mark.parametrize("bump", [0, 1])
@pytest.mark.parametrize("renderer", [api.RENDERER_BIASED, api.RENDERER_UNBIASED])
def test_original_restir_on_textured_street(built_lib, renderer, bump):
    with util.frame_overrides(enableBumpMapping=bump):
        diffs = run_sequence_both(_street(), 160, 96, frames=3, renderer=renderer, scene_kind="street")
    assert not diffs, "\n".join(diffs[:12])
    beauty = run_sequence_both.last_beauty
    assert np.isfinite(beauty).all() and beauty[:, :3].mean() > 1e-3


def test_textures_change_the_image(built_lib):
    with util.frame_overrides(enableBumpMapping=1):
        assert not run_sequence_both(_street(), 96, 64, frames=1, scene_kind="street")
    textured = run_sequence_both.last_beauty.copy()
    assert not run_sequence_both(util.small_street(), 96, 64, frames=1, scene_kind="street")
    plain = run_sequence_both.last_beauty
    assert np.mean(np.abs(textured[:, :3] - plain[:, :3])) > 1e-3 * np.mean(np.abs(plain[:, :3]))


@pytest.mark.parametrize("unbiased", [False, True])
def test_rearchitected_restir_on_textured_street(built_lib, unbiased):
    with util.frame_overrides(enableBumpMapping=1):
        diffs = run_rearch_both(_street(), 128, 80, frames=3, temporal=True, spatial=True, unbiased=unbiased)
    assert not diffs, "\n".join(diffs[:12])


def test_path_tracer_on_textured_street(built_lib):
    with util.frame_overrides(enableBumpMapping=1):
        diffs = run_pt_both(_street(), 128, 80, frames=2, max_len=5)
    assert not diffs, "\n".join(diffs[:12])


def test_regir_on_textured_street(built_lib):
    with util.frame_overrides(enableBumpMapping=1):
        diffs = run_regir_both(_street(), 96, 64, frames=3, max_len=4)
    assert not diffs, "\n".join(diffs[:12])


def test_nrc_on_textured_street(built_lib):
    with util.frame_overrides(enableBumpMapping=1):
        diffs = run_nrc_both(_street(), 96, 64, frames=2, max_len=5)
    assert not diffs, "\n".join(diffs[:12])


def _bump_scene(kind):
    """A lit floor whose material carries the given kind of bump map (and a SimplePBR material with both maps)."""
    rng = np.random.default_rng(3 + kind)
    s = api.HostScene()
    n = 32
    if kind == api.BUMP_HEIGHT_MAP:
        tex = s.add_texture(rng.integers(0, 256, (n, n), dtype=np.uint8), api.TEX_R8_UNORM)
    elif kind == api.BUMP_NORMAL_MAP_2CH:
        tex = s.add_texture((128 + rng.integers(-40, 41, (n, n, 2))).astype(np.uint8), api.TEX_RG8_UNORM)
    else:
        nm = np.concatenate([(128 + rng.integers(-40, 41, (n, n, 2))), np.full((n, n, 1), 230), np.full((n, n, 1), 255)], -1).astype(np.uint8)
        tex = s.add_texture(nm, api.TEX_RGBA8_UNORM)
    base = s.add_texture(rng.integers(30, 256, (16, 16, 4), dtype=np.uint8), api.TEX_RGBA8_SRGB)
    orm = s.add_texture(rng.integers(0, 256, (8, 8, 4), dtype=np.uint8), api.TEX_RGBA8_UNORM)
    m = api.GfxMaterial()
    m.bsdfType = api.BSDF_SIMPLE_PBR if hasattr(api, "BSDF_SIMPLE_PBR") else 2
    m.a[:] = (0.5, 0.5, 0.5)
    m.b[:] = (1.0, 0.6, 0.1)
    m.texA, m.texB, m.texNormal = base, orm, tex
    m.bumpMapType = kind | (api.BUMP_LEFT_HANDED if kind == api.BUMP_NORMAL_MAP else 0)
    mat = s.add_material(m)
    v = np.zeros(4, api.VERTEX_DTYPE)
    v["position"] = [(-6, 0, -6), (6, 0, -6), (6, 0, 6), (-6, 0, 6)]
    v["normal"] = (0, 1, 0)
    v["texCoord0Dir"] = (1, 0, 0)
    v["texCoord"] = [(0, 0), (3, 0), (3, 3), (0, 3)]
    s.add_instance(s.add_group([s.add_geom(v, [(0, 2, 1), (0, 3, 2)], mat)]), api.make_transform())
    s.add_instance(s.add_rectangle(1.5, 1.5, (40, 36, 30)), api.make_transform(pos=(0.5, 4.0, 0.5)))
    return s


@pytest.mark.parametrize("kind", [api.BUMP_NORMAL_MAP, api.BUMP_NORMAL_MAP_2CH, api.BUMP_HEIGHT_MAP])
def test_every_bump_map_kind(built_lib, kind):
    cam = api.make_camera(128, 80, pos=(0.0, 5.0, 9.0), pitch=25.0, yaw=180.0)
    with util.frame_overrides(enableBumpMapping=1):
        diffs = run_sequence_both(_bump_scene(kind), 128, 80, frames=2, camera=cam)
    assert not diffs, "\n".join(diffs[:12])
    with util.frame_overrides(enableBumpMapping=1):
        diffs = run_pt_both(_bump_scene(kind), 96, 64, frames=1, max_len=4, camera=api.make_camera(96, 64, pos=(0.0, 5.0, 9.0), pitch=25.0, yaw=180.0))
    assert not diffs, "\n".join(diffs[:12])
