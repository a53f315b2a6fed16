"""An independent check of the host scene builder's OBJ / MTL conversion (gfxh_scene_load_obj, scene_builder.cpp): a reader
written here from the Wavefront format alone (no shared code, no vertex welding) must describe the same triangle soup,
material by material, and the same immediate material values (createTriangleMeshes, common/common_host.cpp:2181-2430:
one geometry per material in order of first use, assimp's FlipUVs / JoinIdenticalVertices / face normals where the file
has none; 8-bit immediate textures read through the sRGB sampler, :1045-1073)."""
import os

import numpy as np
import pytest

from gfxexp_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASSETS = os.path.join(ROOT, "tests", "golden", "assets")

SYNTHETIC_OBJ = """\
mtllib two.mtl
v 0 0 0
v 1 0 0
v 1 1 0
v 0 1 0
v 0 0 1
v 1 0 1
v 1 1 1
v 0 1 1
vt 0 0
vt 1 0
vt 1 0.75
vt 0.25 1
vn 0 0 -1
vn 0 0 2
usemtl red
f 1/1/1 4/4/1 3/3/1 2/2/1
usemtl shiny
f 5/1/2 6/2/2 7/3/2 8/4/2
f -8 -7 -3
usemtl red
f 2//1 3//1 7//1
f 1/2 5/3 8/4
"""
SYNTHETIC_MTL = """\
newmtl red
Kd 0.8 0.1 0.05
Ks 0.02 0.02 0.02
Ns 25
newmtl shiny
Kd 0.2 0.2 0.2
Ks 0.9 0.6 0.3
Ke 3 2 1
Ns 400
"""


def read_obj(path):
    """{material: [triangles of 3 corners (position, uv or None, normal or None)]} in order of first use; + mtl dict."""
    pos, uv, nrm, mats, order = [], [], [], {}, []
    mtl, cur, cur_mtl = {}, "", None
    base = os.path.dirname(path)
    for line in open(path):
        tok = line.split()
        if not tok or tok[0].startswith("#"):
            continue
        if tok[0] == "v":
            pos.append([float(x) for x in tok[1:4]])
        elif tok[0] == "vt":
            uv.append([float(tok[1]), float(tok[2]) if len(tok) > 2 else 0.0])
        elif tok[0] == "vn":
            nrm.append([float(x) for x in tok[1:4]])
        elif tok[0] == "mtllib":
            for ml in open(os.path.join(base, tok[1])):
                mt = ml.split()
                if not mt:
                    continue
                if mt[0] == "newmtl":
                    cur_mtl = mtl.setdefault(mt[1], {"Kd": [0, 0, 0], "Ks": [0, 0, 0], "Ke": [0, 0, 0], "Ns": 0.0})
                elif mt[0] in ("Kd", "Ks", "Ke"):
                    cur_mtl[mt[0]] = [float(x) for x in mt[1:4]]
                elif mt[0] == "Ns":
                    cur_mtl["Ns"] = float(mt[1])
        elif tok[0] == "usemtl":
            cur = tok[1]
        elif tok[0] == "f":
            corners = []
            for c in tok[1:]:
                idx = (c.split("/") + ["", ""])[:3]
                ref = []
                for k, pool in zip(idx, (pos, uv, nrm)):
                    if k == "":
                        ref.append(None)
                    else:
                        i = int(k)
                        ref.append(pool[i - 1] if i > 0 else pool[len(pool) + i])
                corners.append(ref)
            if cur not in mats:
                mats[cur] = []
                order.append(cur)
            for k in range(1, len(corners) - 1):           # fan triangulation
                mats[cur].append((corners[0], corners[k], corners[k + 1]))
    return order, mats, mtl


def expected_soup(tris):
    """float32 arrays [n, 3, 3] positions, [n, 3, 2] texcoords (v flipped), [n, 3, 3] unit normals (face normal where missing)."""
    n = len(tris)
    P, T, N = np.zeros((n, 3, 3), np.float32), np.zeros((n, 3, 2), np.float32), np.zeros((n, 3, 3), np.float64)
    for i, tri in enumerate(tris):
        for k, (p, t, nn) in enumerate(tri):
            P[i, k] = p
            if t is not None:
                T[i, k] = (np.float32(t[0]), np.float32(1.0) - np.float32(t[1]))
        if any(c[2] is None for c in tri):
            p = P[i].astype(np.float64)
            fn = np.cross(p[1] - p[0], p[2] - p[0])
            N[i, :] = fn / np.linalg.norm(fn)
        else:
            for k in range(3):
                v = np.array(tri[k][2], np.float64)
                N[i, k] = v / np.linalg.norm(v)
    return P, T, N


def srgb_immediate(v):
    q = min(int(np.float32(255) * np.float32(v)), 255) / 255.0
    return q / 12.92 if q <= 0.04045 else ((q + 0.055) / 1.055) ** 2.4


@pytest.fixture(scope="module")
def synthetic(tmp_path_factory):
    d = tmp_path_factory.mktemp("obj")
    (d / "two.obj").write_text(SYNTHETIC_OBJ)
    (d / "two.mtl").write_text(SYNTHETIC_MTL)
    return str(d / "two.obj")


@pytest.mark.parametrize("which", ["synthetic", "stanford_bunny_309_faces.obj", "teapot.obj"])
def test_obj_conversion_against_an_independent_reader(built_lib, synthetic, which):
    path = synthetic if which == "synthetic" else os.path.join(ASSETS, which)
    order, mats, mtl = read_obj(path)
    hs = api.HostScene()
    group = hs.load_obj(path)
    geoms, materials = hs.geoms(), hs.materials()
    assert len(geoms) == len(order) and list(hs.groups()[group]) == list(range(len(order)))
    for gi, name in enumerate(order):
        v, t, mat_slot = geoms[gi]
        P, T, N = expected_soup(mats[name])
        assert len(t) == len(P), (name, len(t), len(P))
        soup = v[np.asarray(t, np.int64)]                                   # [n, 3] records: the builder's indexing undone
        assert np.array_equal(soup["position"], P), name
        assert np.array_equal(soup["texCoord"], T), name
        assert np.abs(soup["normal"].astype(np.float64) - N).max() < 3e-7, name
        nn = v["normal"].astype(np.float64)
        tt = v["texCoord0Dir"].astype(np.float64)
        assert np.abs(np.linalg.norm(nn, axis=1) - 1).max() < 1e-6 and np.abs(np.linalg.norm(tt, axis=1) - 1).max() < 1e-6
        assert np.abs((nn * tt).sum(1)).max() < 1e-5
        # aiProcess_JoinIdenticalVertices: one vertex per distinct (position, texcoord, normal) reference of the file
        keys = set()
        for fi, tri in enumerate(mats[name]):
            face_normal = any(c[2] is None for c in tri)
            for p, tc, n_ in tri:
                keys.add((tuple(p), None if tc is None else tuple(tc), ("face", fi) if face_normal else tuple(n_)))
        assert len(v) <= len(keys) and len(np.unique(v.view(np.uint8).reshape(len(v), -1), axis=0)) == len(v)
        m = materials[mat_slot]
        d = mtl.get(name, {"Kd": [0, 0, 0], "Ks": [0, 0, 0], "Ke": [0, 0, 0], "Ns": 0.0})
        assert m.bsdfType == 1                                              # DiffuseAndSpecular ("trad")
        np.testing.assert_allclose(list(m.a), [srgb_immediate(x) for x in d["Kd"]], rtol=2e-6, atol=1e-9)
        np.testing.assert_allclose(list(m.b), [srgb_immediate(x) for x in d["Ks"]], rtol=2e-6, atol=1e-9)
        smooth = min(int(np.float32(255) * (np.sqrt(np.float32(d["Ns"])) / np.float32(11.0))), 255) / 255.0
        assert abs(m.smoothness - smooth) < 1e-6
        assert list(m.emittance) == [np.float32(x) for x in d["Ke"]] and m.hasEmittance == int(any(x != 0 for x in d["Ke"]))


def test_simple_pbr_convention(built_lib, synthetic):
    """"-obj <path> <scale> simple_pbr": Kd is base colour behind the sRGB sampler, Ks is (occlusion, roughness, metallic)
    behind the normalised-float sampler -- 8-bit quantised, no degamma (createSimplePBRMaterial, common_host.cpp:1689-1760)."""
    hs = api.HostScene()
    hs.load_obj(synthetic, simple_pbr=True)
    m = hs.materials()[1]                                                   # "shiny"
    assert m.bsdfType == 2
    np.testing.assert_allclose(list(m.a), [srgb_immediate(0.2)] * 3, rtol=2e-6)
    np.testing.assert_allclose(list(m.b), [int(255 * np.float32(x)) / 255.0 for x in (0.9, 0.6, 0.3)], rtol=1e-6)


def test_tangents_follow_the_texture_coordinates(built_lib, tmp_path):
    """aiProcess_CalcTangentSpace (common_host.cpp:2163, 2346-2368): a vertex with texture coordinates carries dP/du as texCoord0Dir --
    the axis a normal map's red channel tilts the normal along -- not the frame made up from the normal.  Planar quads whose u axis
    runs along +x, along +z and along -x (mirrored), and one without texture coordinates, which keeps the made-up frame."""
    quads = {"u_along_x": [(0, 0), (1, 0), (1, 1), (0, 1)], "u_along_z": [(0, 0), (0, 1), (1, 1), (1, 0)], "u_mirrored": [(1, 0), (0, 0), (0, 1), (1, 1)]}
    want = {"u_along_x": (1, 0, 0), "u_along_z": (0, 0, 1), "u_mirrored": (-1, 0, 0)}
    lines, nv, nt = ["vn 0 1 0"], 0, 0
    for k, (name, uvs) in enumerate(quads.items()):
        x0 = 3.0 * k
        for (x, z) in [(0, 0), (2, 0), (2, 2), (0, 2)]:
            lines.append("v %g 0 %g" % (x0 + x, z))
        for (u, v) in uvs:
            lines.append("vt %g %g" % (u, v))
        lines.append("usemtl " + name)
        # counter-clockwise seen from +y: (0,0) (0,2) (2,2) (2,0) -> corners 1 4 3 2
        lines.append("f %d/%d/1 %d/%d/1 %d/%d/1 %d/%d/1" % (nv + 1, nt + 1, nv + 4, nt + 4, nv + 3, nt + 3, nv + 2, nt + 2))
        nv += 4; nt += 4
    for (x, z) in [(20, 0), (22, 0), (22, 2), (20, 2)]:
        lines.append("v %g 0 %g" % (x, z))
    lines += ["usemtl no_uv", "f %d//1 %d//1 %d//1 %d//1" % (nv + 1, nv + 4, nv + 3, nv + 2)]
    p = tmp_path / "quads.obj"
    p.write_text("\n".join(lines) + "\n")
    hs = api.HostScene()
    hs.load_obj(str(p))
    geoms = hs.geoms()
    assert len(geoms) == 4
    for gi, name in enumerate(list(quads) + ["no_uv"]):
        v, t, _ = geoms[gi]
        tt = v["texCoord0Dir"].astype(np.float64)
        assert np.abs(np.linalg.norm(tt, axis=1) - 1).max() < 1e-6 and np.abs(tt[:, 1]).max() < 1e-6      # unit, in the quad's plane
        if name in want:
            assert np.abs(tt - np.asarray(want[name], np.float64)).max() < 1e-6, (name, tt)
    # a curved mesh with texture coordinates: the vertex tangents lie on the side of their triangles' dP/du
    path = os.path.join(ASSETS, "teapot.obj")
    order, mats, _ = read_obj(path)
    hs = api.HostScene()
    hs.load_obj(path)
    checked = agree = 0
    for gi, name in enumerate(order):
        v, t, _ = hs.geoms()[gi]
        P, T, N = expected_soup(mats[name])
        if T is None or not np.isfinite(np.asarray(T, np.float64)).all():
            continue
        P3, T3 = np.asarray(P, np.float64).reshape(-1, 3, 3), np.asarray(T, np.float64).reshape(-1, 3, 2)
        e1, e2 = P3[:, 1] - P3[:, 0], P3[:, 2] - P3[:, 0]
        d1, d2 = T3[:, 1] - T3[:, 0], T3[:, 2] - T3[:, 0]
        det = d1[:, 0] * d2[:, 1] - d2[:, 0] * d1[:, 1]
        ok = np.abs(det) > 1e-12
        dpdu = (e1 * d2[:, 1:2] - e2 * d1[:, 1:2]) / np.where(ok, det, 1.0)[:, None]
        tri = np.asarray(t, np.int64).reshape(-1, 3)
        tt = v["texCoord0Dir"].astype(np.float64)
        for k in range(3):
            dots = (tt[tri[ok, k]] * dpdu[ok]).sum(1)
            checked += len(dots); agree += int((dots > 0).sum())
    if checked:
        assert agree > 0.9 * checked, (agree, checked)


def test_windows_line_ends_backslashes_and_two_material_libraries(built_lib, tmp_path):
    """What exported OBJ files look like in practice: CRLF line ends, a texture path written with backslashes, `mtllib` naming two files."""
    (tmp_path / "tex").mkdir()
    (tmp_path / "tex" / "albedo.ppm").write_bytes(b"P6\n2 2\n255\n" + bytes(range(12)))
    (tmp_path / "a.mtl").write_bytes(b"newmtl red\r\nKd 1 0 0\r\nmap_Kd tex\\albedo.ppm\r\n")
    (tmp_path / "b.mtl").write_bytes(b"newmtl blue\r\nKd 0 0 1\r\n")
    (tmp_path / "q.obj").write_bytes(b"mtllib a.mtl b.mtl\r\nv 0 0 0\r\nv 1 0 0\r\nv 1 1 0\r\nv 0 1 0\r\nvt 0 0\r\nvt 1 0\r\nvt 1 1\r\nvt 0 1\r\n"
                                     b"usemtl red\r\nf 1/1 2/2 3/3\r\nusemtl blue\r\nf 1/1 3/3 4/4\r\n")
    hs = api.HostScene()
    hs.load_obj(str(tmp_path / "q.obj"))
    geoms, mats = hs.geoms(), hs.materials()
    assert len(geoms) == 2
    red, blue = mats[geoms[0][2]], mats[geoms[1][2]]
    assert list(red.a) == [1.0, 0.0, 0.0] and red.texA == 1                  # the texture was found behind the backslash path
    assert list(blue.a) == [0.0, 0.0, 1.0] and blue.texA == 0                # the second library was read


def test_material_constants_outside_the_unit_interval_are_defined(built_lib, tmp_path):
    """The reference's 8-bit immediate-texture value is `min<uint32_t>(255 * v, 255)`: undefined for a negative or non-finite constant.
    The builder spells out what the reference's x86-64 build does (truncate to 64 bits, keep the low word): negative -> 255, NaN -> 0,
    huge -> 0; and an error message that quotes bytes of the file which are not UTF-8 is still a GfxError."""
    (tmp_path / "m.mtl").write_bytes(b"newmtl odd\nKd -0.5 nan 1e30\nKs 0.5 2 -1e30\n")
    (tmp_path / "q.obj").write_bytes(b"mtllib m.mtl\nv 0 0 0\nv 1 0 0\nv 1 1 0\nusemtl odd\nf 1 2 3\n")
    hs = api.HostScene()
    hs.load_obj(str(tmp_path / "q.obj"))
    m = hs.materials()[hs.geoms()[0][2]]
    assert list(m.a) == [1.0, 0.0, 0.0], list(m.a)
    assert 0.2 < m.b[0] < 0.22 and m.b[1] == 1.0 and m.b[2] == 0.0, list(m.b)          # (127 / 255 through the sRGB decode; clamp; low word 0)
    (tmp_path / "bad.obj").write_bytes(b"mtllib \xa7\xff.mtl\nv 0 0 0\nusemtl \xa7\nf 1 2 9\n")
    with pytest.raises(api.GfxError):
        api.HostScene().load_obj(str(tmp_path / "bad.obj"))
