"""-m gpu: the emitter interval table (gfxexp_amd/csrc/emitter_spans.h, built by gfx_lights_build_instances) is
in use and has passed its own on-device verification -- every interval end was cross-checked against the
reference's three nested searches (light_locate_3level) by k_span_finish.  The bit-exact renderer tests then
exercise the lookup on millions of candidates; this test makes sure they are not silently running the fallback."""
import numpy as np
import pytest

from gfxexp_amd import api
from tests import util

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("scene", ["bunny", "small_street", "pathological"])
def test_table_is_usable_and_fully_verified(built_lib, scene):
    hs = {"bunny": util.bunny_scene, "small_street": util.small_street, "pathological": util.pathological_light_scene}[scene]()
    ctx = api.Context(0)
    hs.upload(ctx)
    ctx.lights_build_static()
    ctx.lights_build_instances()
    info = ctx.lights_table_info()
    assert info["records"] > 0
    assert info["usable"] == 1, info
    assert info["verified"] == info["records"], info
    assert info["cells"] >= 256 and info["cells"] & (info["cells"] - 1) == 0
    assert 0 < info["interior_cells"] < info["cells"], info      # both lookup paths are in use
    assert info["matrices"] >= 1, info


def test_scene_without_emitters_keeps_the_search_path(built_lib):
    s = api.HostScene()
    m = s.add_material_traditional((0.5, 0.5, 0.5), (0.04, 0.04, 0.04), 0.3)
    v = np.zeros(3, api.VERTEX_DTYPE)
    v["position"] = [(0, 0, 0), (1, 0, 0), (0, 1, 0)]
    v["normal"] = (0, 0, 1)
    v["texCoord0Dir"] = (1, 0, 0)
    g = s.add_group([s.add_geom(v, [(0, 1, 2)], m)])
    s.add_instance(g, api.make_transform())
    ctx = api.Context(0)
    s.upload(ctx)
    ctx.lights_build_static()
    ctx.lights_build_instances()
    info = ctx.lights_table_info()
    assert info["records"] == 0 and info["usable"] == 0
