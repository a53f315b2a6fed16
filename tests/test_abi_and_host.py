"""CPU: the C-ABI library loads and exports every symbol the headers declare (no compute calls),
and the host layer (scene builder, seeds, tables) behaves like the reference host code."""
import ctypes as C
import os
import re

import numpy as np

from gfxexp_amd import api
from oracle import oracle as O
from tests import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gfxh?_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported(built_lib):
    names = _declared("gfxexp.h") + _declared("gfxexp_host.h")
    assert len(names) > 40
    for n in names:
        assert hasattr(built_lib, n), f"{n} declared in include/ but not exported by libgfxexp.so"
    assert set(api.C_ABI_SYMBOLS) <= set(_declared("gfxexp.h"))
    assert set(api.HOST_ABI_SYMBOLS) <= set(_declared("gfxexp_host.h"))
    assert built_lib.gfx_version().decode().endswith("gfx950")


def test_ctypes_mirrors_match_the_compiled_headers(built_lib):
    """Every ctypes class of api.py against sizeof / offsetof of the struct it mirrors AS COMPILED into libgfxexp.so
    (gfxh_abi_layout, csrc/host/abi_layout.cpp): same field names in the same order, same offsets and sizes, same total size.
    A one-field edit of include/gfxexp.h or include/gfxexp_host.h fails here by struct and field name."""
    layout = api.abi_layout()
    mirrors = api.abi_mirrors()
    assert set(mirrors) <= set(layout)
    for name, cls in mirrors.items():
        want = layout[name]
        got = [(f[0].rstrip("_"), getattr(cls, f[0]).offset, getattr(cls, f[0]).size) for f in cls._fields_]
        assert got == want["fields"], f"{name}: api.py {cls.__name__} differs from the header: {[g for g, w in zip(got, want['fields']) if g != w][:3] or (len(got), len(want['fields']))}"
        assert C.sizeof(cls) == want["size"], f"{name}: sizeof {C.sizeof(cls)} in api.py, {want['size']} in the library"
    # the numpy record types of the G-buffer / hit / vertex arrays
    for name, dt in (("gfx_vertex", api.VERTEX_DTYPE), ("gfx_gbuffer0", api.GBUFFER0_DTYPE), ("gfx_gbuffer2", api.GBUFFER2_DTYPE),
                     ("gfx_gbuffer3", api.GBUFFER3_DTYPE), ("gfx_hit", api.HIT_DTYPE), ("gfx_tri_ids", api.TRI_IDS_DTYPE)):
        want = layout[name]
        assert dt.itemsize == want["size"], name
        assert [(n, dt.fields[n][1], dt.fields[n][0].itemsize) for n in dt.names] == want["fields"], name
    # the lookup form, and its failure modes
    off, size = C.c_uint64(), C.c_uint64()
    assert built_lib.gfxh_abi_layout(b"gfxh_exchange_desc", b"lane", C.byref(off), C.byref(size)) == 0 and (off.value, size.value) == (8, 4)
    assert built_lib.gfxh_abi_layout(b"gfx_material", None, C.byref(off), C.byref(size)) == 0 and size.value == 80
    assert built_lib.gfxh_abi_layout(b"gfx_material", b"no_such_field", C.byref(off), C.byref(size)) == 1
    assert built_lib.gfxh_abi_layout(b"no_such_struct", None, C.byref(off), C.byref(size)) == 1


def test_a_drifted_mirror_is_caught_by_name(built_lib):
    """The check above does catch a one-field drift: a copy of GfxhFrameStep without its last field, and one with two fields swapped."""
    layout = api.abi_layout()["gfxh_frame_step"]

    class Short(C.Structure):
        _fields_ = api.GfxhFrameStep._fields_[:-1]

    class Swapped(C.Structure):
        _fields_ = [api.GfxhFrameStep._fields_[1], api.GfxhFrameStep._fields_[0]] + api.GfxhFrameStep._fields_[2:]
    assert C.sizeof(Short) != layout["size"]
    got = [(f[0].rstrip("_"), getattr(Swapped, f[0]).offset, getattr(Swapped, f[0]).size) for f in Swapped._fields_]
    assert got != layout["fields"] and C.sizeof(Swapped) == layout["size"]


def test_oracle_bindings_mirror_the_same_structs():
    """The checker's own ctypes classes (oracle/oracle.py, over liboracle.so's copies of the structs) keep the product's layout."""
    for a, b in ((api.GfxMaterial, O.GfxMaterial), (api.GfxCamera, O.GfxCamera),
                 (api.GfxRestirStaticParams, O.GfxRestirStaticParams), (api.GfxRestirFrameParams, O.GfxRestirFrameParams)):
        assert C.sizeof(a) == C.sizeof(b)
        assert [f[0] for f in a._fields_] == [f[0] for f in b._fields_]


def test_context_creation_fails_loudly_without_gpu(built_lib):
    import torch
    if torch.cuda.is_available():
        return
    try:
        api.Context(0)
    except api.GfxError as e:
        assert "gfx_ctx_create" in str(e)
    else:
        raise AssertionError("gfx_ctx_create succeeded without a GPU")


def test_obj_loader_and_immediate_materials(built_lib):
    hs = util.bunny_scene(with_light=True)
    c = hs.counts()
    assert c["triangles"] == 309 + 2 + 2 + 2
    v, t, mat = hs.geoms()[0]
    assert len(t) == 309 and t.max() < len(v)
    np.testing.assert_allclose(np.linalg.norm(v["normal"], axis=1), 1.0, atol=1e-5)
    np.testing.assert_allclose(np.linalg.norm(v["texCoord0Dir"], axis=1), 1.0, atol=1e-5)
    assert np.abs(np.sum(v["normal"] * v["texCoord0Dir"], axis=1)).max() < 1e-4
    m = hs.materials()[0]
    # Kd 0.64 -> byte 163 -> 163/255 -> sRGB decode; Ns 96.078 -> sqrt/11 -> byte
    srgb = ((163 / 255 + 0.055) / 1.055) ** 2.4
    np.testing.assert_allclose(list(m.a), [srgb] * 3, rtol=1e-5)
    ks = ((127 / 255 + 0.055) / 1.055) ** 2.4
    np.testing.assert_allclose(list(m.b), [ks] * 3, rtol=1e-5)
    assert abs(m.smoothness - int(255 * (96.078453 ** 0.5 / 11)) / 255) < 1e-6
    assert m.hasEmittance == 0 and m.bsdfType == 1
    light = hs.materials()[2]
    assert light.hasEmittance == 1 and list(light.emittance) == [50.0, 50.0, 50.0]
    assert abs(light.smoothness - 76 / 255) < 1e-6       # rectangle lights: smoothness 0.3 (common_host.cpp:2444-2447)


def test_seeds_and_neighbour_table_agree_with_oracle(built_lib):
    assert np.array_equal(api.seed_rng_states(4096, util.PIXEL_RNG_SEED), O.seed_rngs(4096, util.PIXEL_RNG_SEED))
    t = api.spatial_neighbor_deltas()
    util.assert_same_bits("halton disk table", t, O.spatial_neighbor_deltas())
    assert np.linalg.norm(t, axis=1).max() <= 1.0 + 1e-6
    # Halton(2,3) of index 1 is (1/2, 1/3): concentric map -> (r=1/3 branch) check a few by hand
    assert np.allclose(t[0], [-np.sqrt(0.5), -np.sqrt(0.5)], atol=1e-6)


def test_street_scene_statistics(built_lib):
    s = util.small_street()
    c = s.counts()
    assert c["insts"] > 50 and c["triangles"] > 10000
    emissive = [i for i, m in enumerate(s.materials()) if m.hasEmittance]
    assert len(emissive) == 9
    # deterministic: same seed -> identical arrays
    s2 = util.small_street()
    for (v1, t1, m1), (v2, t2, m2) in zip(s.geoms(), s2.geoms()):
        assert m1 == m2 and np.array_equal(t1, t2) and np.array_equal(v1.view(np.uint8), v2.view(np.uint8))
    for (g1, x1), (g2, x2) in zip(s.instances(), s2.instances()):
        assert g1 == g2 and np.array_equal(x1, x2)
    osc = util.feed_oracle(s)
    w, cdf, integral = osc.lights_read(0)
    assert (w > 0).sum() == 24 + 12 and integral > 0


def test_output_chain_tone_map_and_files(built_lib, tmp_path):
    """saveImage's SDR conversion (common_host.cpp:2859-2897) against a numpy restatement, and the file writers."""
    rng = np.random.default_rng(3)
    w, h = 37, 19
    img = (rng.random((h, w, 4)) ** 4 * 20).astype(np.float32)
    img[0, 0, :3] = np.nan; img[1, 1, :3] = 0.0; img[2, 2, :3] = (np.inf, 1.0, 1.0)
    cfg = api.sdr_config(brightness=0.7, tone_map=True, gamma=True, flip_y=True)
    got = api.tonemap_sdr(img, w, h, cfg)
    src = img[::-1].astype(np.float32).copy()
    rgb = src[..., :3].copy()
    bad = ~np.isfinite(rgb).all(axis=2)
    rgb[bad] = 0
    lum = (np.float32(0.2126729) * rgb[..., 0] + np.float32(0.7151522) * rgb[..., 1] + np.float32(0.0721750) * rgb[..., 2]).astype(np.float32)
    lum_t = (1 - np.exp(-(np.float32(0.7) * lum))).astype(np.float32)
    s = np.where(lum > 0, lum_t / np.where(lum > 0, lum, 1), 0).astype(np.float32)
    rgb = rgb * s[..., None]
    gamma = np.where(rgb <= 0.0031308, 12.92 * rgb, 1.055 * np.power(np.maximum(rgb, 0), 1 / 2.4) - 0.055)
    q = np.minimum((np.maximum(gamma, 0) * 255).astype(np.uint32), 255)
    ref = q[..., 0] | (q[..., 1] << 8) | (q[..., 2] << 16)
    diff = np.abs(((got & 0xFFFFFF).astype(np.int64) >> np.array([0, 8, 16])[:, None, None] & 255) -
                  ((ref.astype(np.int64) >> np.array([0, 8, 16])[:, None, None]) & 255))
    assert diff.max() <= 1                                  # float32 vs float64 pow at a quantisation step
    assert (got >> 24).min() >= 0
    api.save_image_sdr(str(tmp_path / "a.bmp"), img, w, h, cfg)
    api.save_image_sdr(str(tmp_path / "a.ppm"), img, w, h, cfg)
    api.save_image_hdr(str(tmp_path / "a.pfm"), img, w, h, 2.0)
    bmp = (tmp_path / "a.bmp").read_bytes()
    assert bmp[:2] == b"BM" and len(bmp) == 54 + ((3 * w + 3) & ~3) * h
    ppm = (tmp_path / "a.ppm").read_bytes()
    assert ppm.startswith(b"P6\n37 19\n255\n") and len(ppm) == len(b"P6\n37 19\n255\n") + 3 * w * h
    pfm = (tmp_path / "a.pfm").read_bytes()
    head = b"PF\n37 19\n-1.0\n"
    assert pfm.startswith(head)
    data = np.frombuffer(pfm[len(head):], np.float32).reshape(h, w, 3)
    assert np.array_equal(data[::-1][3:, 3:], (np.float32(2.0) * img[3:, 3:, :3]))      # finite region, bottom-up rows


def test_image_headers_with_absurd_dimensions_are_refused(built_lib, tmp_path):
    """Asset headers are untrusted: dimensions are bounded (16384, the 14-bit TexDimInfo limit) before any size arithmetic, so
    32-bit width x height products cannot wrap and -INT_MIN is never formed."""
    import struct
    import pytest
    s = api.HostScene()
    p = tmp_path / "huge.ppm"
    p.write_bytes(b"P6\n70000 70000\n255\n" + bytes(64))
    with pytest.raises(api.GfxError, match="larger than 16384"):
        s.load_texture(str(p))
    p = tmp_path / "wrap.ppm"                     # 65536 x 65536 x 3 wraps a 32-bit product to 0
    p.write_bytes(b"P6\n65536 65536\n255\n" + bytes(64))
    with pytest.raises(api.GfxError, match="larger than 16384"):
        s.load_texture(str(p))
    p = tmp_path / "huge.pfm"
    p.write_bytes(b"PF\n40000 40000\n-1.0\n" + bytes(64))
    with pytest.raises(api.GfxError, match="larger than 16384"):
        s.load_texture(str(p))
    hdr = bytearray(54)
    hdr[0:2] = b"BM"
    struct.pack_into("<I", hdr, 10, 54)
    struct.pack_into("<ii", hdr, 18, 4, -2147483648)     # height = INT_MIN
    struct.pack_into("<H", hdr, 28, 24)
    p = tmp_path / "intmin.bmp"
    p.write_bytes(bytes(hdr) + bytes(64))
    with pytest.raises(api.GfxError, match="larger than 16384"):
        s.load_texture(str(p))
    ok = tmp_path / "ok.ppm"                      # a sane file still loads
    ok.write_bytes(b"P6\n2 2\n255\n" + bytes(range(12)))
    assert s.load_texture(str(ok)) == 1
    big = np.zeros(4, np.uint8)
    assert api.lib().gfxh_scene_add_texture(s.h, C.c_uint32(20000), C.c_uint32(2), C.c_uint32(api.TEX_RGBA8_UNORM), big.ctypes.data_as(C.c_void_p)) == 0


def test_counter_files_carry_the_hash_of_the_sources_they_were_measured_on(tmp_path, monkeypatch):
    """bench.py prints, next to every figure it takes from a committed counter file, which commit and which kernel sources the file
    belongs to and whether those are the running library's (roofline.pmc_head / pmc_matches_sources): profiles/make_pmc_json.py and
    bench.py hash gfxexp_amd/csrc the same way, a stale or foreign file shows as a mismatch, a file without provenance as None."""
    import importlib.util
    import json
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    spec = importlib.util.spec_from_file_location("make_pmc_json", os.path.join(root, "profiles", "make_pmc_json.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    sha = bench.sources_sha16()
    assert len(sha) == 16 and sha == mk.sources_sha16()
    prof = tmp_path / "profiles"
    prof.mkdir()
    (prof / "ok.json").write_text(json.dumps({"sources_sha16": sha, "git_head": "abc123", "kernels": {"k_trace_any": {}}}))
    (prof / "stale.json").write_text(json.dumps({"sources_sha16": "0" * 16, "git_head": "def456", "kernels": {}}))
    (prof / "old.json").write_text(json.dumps({"kernels": {}}))
    monkeypatch.setattr(bench, "_profile_value", lambda name, key: json.load(open(prof / name)).get(key))
    ok = bench._pmc_provenance("profiles/ok.json")
    assert ok["pmc_matches_sources"] is True and ok["pmc_head"] == "abc123" and ok["run_sources_sha16"] == sha
    stale = bench._pmc_provenance("profiles/stale.json")
    assert stale["pmc_matches_sources"] is False and stale["pmc_head"] == "def456"
    assert bench._pmc_provenance("profiles/old.json")["pmc_matches_sources"] is None
    assert bench._pmc_provenance(None) == {"pmc_head": None, "pmc_sources_sha16": None, "pmc_matches_sources": None}
    # every BASELINE configuration has a counter-file slot, newest round first
    assert set(bench.PMC_FILES) == {1, 2, 4, "animate"} and bench.PMC_FILES[2][0].startswith("r06_")


def test_the_committed_bench_line_keeps_the_driver_contract():
    """profiles/r06_bench_default.json is what `python bench.py` printed on an MI355X at the end of the round: ONE JSON line with the
    fields the driver reads, the roofline and cpu_baseline objects of the measurement section, every other BASELINE configuration inside
    it, and counter figures that belong to the kernel sources of this tree."""
    import json
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    lines = [ln for ln in open(os.path.join(root, "profiles", "r06_bench_default.json")).read().splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None and d["data"] == "synthetic" and d["dtype"] == "f32"
    assert d["unit"] == "Mpaths/s" and "workload" in d["config"] and "configs[2]" in d["config"]["workload"]
    assert abs(d["value"] - d["config"]["width"] * d["config"]["height"] / d["ms_per_step"] / 1e3) < 0.01 * d["value"]      # value and ms_per_step are one measurement
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_ms", "algorithmic_bytes_per_launch", "frac_nominal_hbm", "frac_hbm_counter"):
        assert key in r, key
    assert abs(r["frac_hbm_counter"] - r["traffic"] / (r["avg_launch_ms"] * 1e-3) / 1e9 / 8000.0) < 1e-3      # counter bytes over the live launch time
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0 < r["frac"] <= 1
    assert r["pmc_matches_sources"] is True and r["run_sources_sha16"] == bench.sources_sha16()     # the counters are this tree's
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0
    assert set(d["other_configs"]) == {"configs[1]", "configs[3]", "configs[4]", "configs[2] --animate", "configs[2] --cluttered"}
    assert d["gpu_bvh_build_ms"]["warm_ms"] > 0 and c["gpu_bvh_build_ms"] == d["gpu_bvh_build_ms"] and c["builds"]["parity"]["bvh_build_s"] > 0      # GPU LBVH beside the CPU SAH build
    assert d["config"]["light_inst_distribution"].startswith("cached")
    for name, o in d["other_configs"].items():
        assert "error" not in o and o["value"] > 0 and o["ms_per_step"] > 0, name
    assert d["mse"]["ref_spp"] == 65536 and d["mse"]["mse"] > 0
