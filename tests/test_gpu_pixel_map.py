"""-m gpu: the pixel -> thread mapping of the per-pixel kernels (gfx_tunable_set "pixel_map",
gfxexp_amd/csrc/restir_common.hip.h) changes which pixels share a wave / block / XCD and nothing else.

Every renderer is run under each mapping -- scan lines (rounds 1-2), 8 x 8 tiles, tiles + XCD supertiles (the
default) -- at image sizes that are no multiple of the tile, block or supertile size, and compared bit for bit with
the CPU oracle after every pass.  The reference tiles its per-tile light subsets the same way
(restir_di/gpu_kernels/per_pixel_ris.cu:44-61); everything else there is one OptiX launch index per pixel.
"""
import pytest

from gfxexp_amd import api
from tests import util
from tests.test_gpu_nrc_render import run_nrc_both
from tests.test_gpu_pathtrace import run_pt_both
from tests.test_gpu_regir import run_regir_both
from tests.test_gpu_restir import run_sequence_both
from tests.test_gpu_restir_rearch import run_rearch_both

MODES = [0, 1, 2]


@pytest.fixture
def pixel_map(monkeypatch, request):
    """gfx_ctx_create reads the mapping from the environment; the harnesses create their own contexts."""
    mode, sx, sy = request.param
    monkeypatch.setenv("GFX_PIXEL_MAP", str(mode))
    monkeypatch.setenv("GFX_SUPER_X", str(sx))
    monkeypatch.setenv("GFX_SUPER_Y", str(sy))
    return mode


def _ids(p):
    return "map%d-super%dx%d" % p


# mode 2 four times: a supertile larger than these small images (8 x 4 blocks: 7 of 8 XCDs get nothing), the default
# 4 x 4, 1 x 1 blocks (every XCD gets blocks) and 2 x 1 (the last supertile row / column is ragged)
CASES = [(0, 2, 2), (1, 2, 2), (2, 3, 2), (2, 2, 2), (2, 0, 0), (2, 1, 0)]


@pytest.mark.gpu
@pytest.mark.parametrize("pixel_map", CASES, indirect=True, ids=_ids)
@pytest.mark.parametrize("renderer", [api.RENDERER_BIASED, api.RENDERER_UNBIASED])
def test_original_restir_under_every_pixel_map(built_lib, pixel_map, renderer):
    diffs = run_sequence_both(util.bunny_scene(), 150, 91, frames=2, renderer=renderer)
    assert not diffs, "\n".join(diffs)


@pytest.mark.gpu
@pytest.mark.parametrize("pixel_map", CASES, indirect=True, ids=_ids)
def test_rearchitected_restir_under_every_pixel_map(built_lib, pixel_map):
    diffs = run_rearch_both(util.bunny_scene(), 150, 91, 2, True, True, True)
    assert not diffs, "\n".join(diffs)


@pytest.mark.gpu
@pytest.mark.parametrize("pixel_map", CASES, indirect=True, ids=_ids)
def test_path_tracer_and_row_bands_under_every_pixel_map(built_lib, pixel_map):
    diffs = run_pt_both(util.bunny_scene(), 150, 91, frames=2, max_len=5)
    assert not diffs, "\n".join(diffs)
    # bands that start and end off the tile grid
    diffs = run_pt_both(util.bunny_scene(), 150, 91, frames=1, max_len=3, rows=[(0, 37), (37, 60), (60, 91)])
    assert not diffs, "\n".join(diffs)


@pytest.mark.gpu
@pytest.mark.parametrize("pixel_map", CASES, indirect=True, ids=_ids)
def test_regir_under_every_pixel_map(built_lib, pixel_map):
    diffs = run_regir_both(util.bunny_scene(), 150, 91, 2, 4)
    assert not diffs, "\n".join(diffs)


@pytest.mark.gpu
@pytest.mark.parametrize("pixel_map", CASES, indirect=True, ids=_ids)
def test_nrc_renderer_under_every_pixel_map(built_lib, pixel_map):
    diffs = run_nrc_both(util.bunny_scene(), 150, 91, 2, 5)
    assert not diffs, "\n".join(diffs)


@pytest.mark.gpu
def test_tunable_set_rejects_unknown_names_and_values(built_lib):
    ctx = api.Context(0)
    ctx.tunable_set("pixel_map", 1)
    with pytest.raises(api.GfxError):
        ctx.tunable_set("pixel_map", 3)
    with pytest.raises(api.GfxError):
        ctx.tunable_set("no_such_knob", 1)


@pytest.mark.gpu
def test_temporal_hints_change_the_work_not_the_frames(built_lib, monkeypatch):
    """gfx_tunable_set "temporal_hints" (trace.hip): a pixel's primary ray first tests the triangle it hit one frame ago.  With the
    hints off every buffer still equals the oracle's (the default -- on -- is what every other multi-frame test runs); with them on
    the second frame of a static camera fetches fewer nodes for its primary rays, the first frame the same number."""
    import torch
    monkeypatch.setenv("GFX_TEMPORAL_HINTS", "0")
    diffs = run_sequence_both(util.bunny_scene(), 150, 91, frames=2, renderer=api.RENDERER_BIASED)
    assert not diffs, "\n".join(diffs)
    monkeypatch.delenv("GFX_TEMPORAL_HINTS")
    monkeypatch.setenv("GFX_SERIAL_FRAMES", "1")   # one frame's passes per render_frame call: the counters below are per frame
    fetched = {}
    for hints in (0, 1):
        ctx = api.Context(0)
        ctx.tunable_set("temporal_hints", hints)
        hs = util.small_street()
        hs.upload(ctx)
        cfg = api.RestirRenderer.default_config(320, 180, api.RENDERER_BIASED)
        cfg.camera = api.make_camera(320, 180, pos=(2.0, 5.0, 26.0), pitch=4.0, yaw=180.0)
        r = api.RestirRenderer(ctx, cfg)
        ctx.counters_enable(True)
        per_frame = []
        for _ in range(3):
            ctx.counters_read(reset=True)
            r.render_frame()
            torch.cuda.synchronize()
            per_frame.append(ctx.counters_read(reset=True)["closest"]["nodeFetches"])
        fetched[hints] = per_frame
        beauty = ctx.read_device(r.beauty_ptr(), 320 * 180 * 16).copy()
        fetched[("beauty", hints)] = beauty
    assert fetched[0][0] == fetched[1][0], fetched                     # nothing to hint at in the first frame
    assert fetched[1][1] < 0.95 * fetched[0][1], fetched               # the second frame starts every ray at last frame's triangle
    assert (fetched[("beauty", 0)] == fetched[("beauty", 1)]).all()    # same frames, bit for bit
