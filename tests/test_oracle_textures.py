"""CPU: the software texture path.  (1) the oracle's tex2DLod / tex2Dgather (oracle/orc_texture.h) against an
independent numpy statement of the written sampler contract (include/gfxexp.h, gfx_texture_set): bilinear, repeat
wrap, 8 fraction bits in the weights, per-texel decode (c / 255, or the sRGB formula) before filtering, fp32 in the
stated operation order; (2) 1x1 textures return their constant (the reference's immediate textures,
common_host.cpp:1045-1073); (3) the host image decoders (PPM / PGM / PFM / BMP / TGA) and the MTL texture maps of the
OBJ loader; (4) the textured street scene; (5) bump mapping keeps the frame orthonormal and is the identity for the
flat (0.5, 0.5, 1) normal.  The GPU side of the same contract is tests/test_gpu_textures.py."""
import os
import struct

import numpy as np
import pytest

from gfxexp_amd import api
from oracle import oracle as O
from tests import util

F = np.float32


def _decode(texels, fmt):
    """numpy decode of a whole texture to (H, W, 4) float32 per the contract."""
    t = np.asarray(texels)
    if fmt == api.TEX_RGBA32F:
        return t.astype(F)
    c = t.astype(F) / F(255.0)
    if fmt == api.TEX_RGBA8_SRGB:
        # the decode table of the contract: degamma(c / 255) in fp32 with the C library's powf (numpy's float32 power
        # is a different implementation and differs in the last bit for some bytes)
        import ctypes
        powf = ctypes.CDLL("libm.so.6").powf
        powf.restype, powf.argtypes = ctypes.c_float, (ctypes.c_float, ctypes.c_float)
        lut = np.array([(F(b) / F(255.0)) / F(12.92) if F(b) / F(255.0) <= F(0.04045) else
                        powf(float((F(b) / F(255.0) + F(0.055)) / F(1.055)), 2.4) for b in range(256)], F)
        return np.concatenate([lut[t[..., :3]], c[..., 3:4]], -1)
    if fmt == api.TEX_R8_UNORM:
        z = np.zeros_like(c)
        return np.stack([c, z, z, np.ones_like(c)], -1)
    if fmt == api.TEX_RG8_UNORM:
        z = np.zeros_like(c[..., 0])
        return np.stack([c[..., 0], c[..., 1], z, np.ones_like(z)], -1)
    return c


def _sample_numpy(dec, uv):
    H, W = dec.shape[:2]
    u, v = uv[:, 0].astype(F), uv[:, 1].astype(F)
    x = (u - np.floor(u)) * F(W) - F(0.5)
    y = (v - np.floor(v)) * F(H) - F(0.5)
    fx, fy = np.floor(x), np.floor(y)
    a = np.floor((x - fx) * F(256) + F(0.5)) / F(256)
    b = np.floor((y - fy) * F(256) + F(0.5)) / F(256)
    i0, j0 = fx.astype(np.int64) % W, fy.astype(np.int64) % H
    i1, j1 = (fx.astype(np.int64) + 1) % W, (fy.astype(np.int64) + 1) % H
    w00, w10, w01, w11 = (1 - a) * (1 - b), a * (1 - b), (1 - a) * b, a * b
    out = (w00[:, None] * dec[j0, i0] + w10[:, None] * dec[j0, i1]) + w01[:, None] * dec[j1, i0]
    out = out + w11[:, None] * dec[j1, i1]
    gather = np.stack([dec[j1, i0][:, 0], dec[j1, i1][:, 0], dec[j0, i1][:, 0], dec[j0, i0][:, 0]], -1)
    return out.astype(F), gather.astype(F)


def _random_texture(rng, fmt, w, h):
    if fmt == api.TEX_RGBA32F:
        return (rng.random((h, w, 4)) * 40).astype(F)
    if fmt == api.TEX_R8_UNORM:
        return rng.integers(0, 256, (h, w), dtype=np.uint8)
    if fmt == api.TEX_RG8_UNORM:
        return rng.integers(0, 256, (h, w, 2), dtype=np.uint8)
    return rng.integers(0, 256, (h, w, 4), dtype=np.uint8)


@pytest.mark.parametrize("fmt", [api.TEX_RGBA8_SRGB, api.TEX_RGBA8_UNORM, api.TEX_R8_UNORM, api.TEX_RG8_UNORM, api.TEX_RGBA32F])
@pytest.mark.parametrize("size", [(1, 1), (7, 5), (64, 32)])
def test_oracle_sampler_equals_the_written_contract(oracle_lib, fmt, size):
    rng = np.random.default_rng(fmt * 31 + size[0])
    w, h = size
    tex = _random_texture(rng, fmt, w, h)
    osc = O.OracleScene()
    osc.set_texture(1, w, h, fmt, tex)
    uv = np.concatenate([rng.random((4000, 2)) * 6 - 3,                    # repeat wrap incl. negative coordinates
                         (rng.integers(-8, 9, (500, 2)) / np.array([w, h])),  # exactly on texel edges
                         (rng.integers(-8, 9, (500, 2)) + 0.5) / np.array([w, h])]).astype(F)   # exactly on texel centres
    want, want_gather = _sample_numpy(_decode(tex, fmt), uv)
    util.assert_same_bits("tex2DLod", osc.texture_sample(1, uv), want)
    util.assert_same_bits("tex2Dgather", osc.texture_sample(1, uv, gather=True), want_gather)
    if size == (1, 1):   # a 1x1 texture returns its texel for every coordinate, up to the rounding of the four weighted terms
        texel = _decode(tex, fmt)[0, 0]
        assert np.all(np.abs(osc.texture_sample(1, uv) - texel) <= 2.4e-7 * np.maximum(np.abs(texel), 1e-30))


def test_srgb_immediate_constant_equals_a_1x1_srgb_texture(oracle_lib):
    """Kd 0.64 -> byte 163 -> 0.366 linear: the constant of gfxh_scene_add_material_traditional is
    what a 1x1 sRGB texture of that byte samples to."""
    s = api.HostScene()
    m = s.add_material_traditional((0.64, 0.2, 0.9), (0.5, 0.5, 0.5), 0.3)
    mat = s.materials()[m]
    osc = O.OracleScene()
    tex = np.array([[[int(255 * 0.64), int(255 * 0.2), int(255 * 0.9), 255]]], np.uint8)
    osc.set_texture(1, 1, 1, api.TEX_RGBA8_SRGB, tex)
    got = osc.texture_sample(1, np.array([[0.5, 0.5]], F))[0]        # texel centre: weights (1, 0, 0, 0), exact
    assert tuple(got[:3]) == (mat.a[0], mat.a[1], mat.a[2])
    assert abs(mat.a[0] - 0.3663) < 2e-3


def _write_images(tmp, rgb, grey, hdr):
    h, w = rgb.shape[:2]
    with open(os.path.join(tmp, "a.ppm"), "wb") as f:
        f.write(b"P6\n# comment\n%d %d\n255\n" % (w, h) + rgb.tobytes())
    with open(os.path.join(tmp, "g.pgm"), "wb") as f:
        f.write(b"P5 %d %d 255\n" % (w, h) + grey.tobytes())
    with open(os.path.join(tmp, "h.pfm"), "wb") as f:
        f.write(b"PF\n%d %d\n-1.0\n" % (w, h) + hdr[::-1].astype("<f4").tobytes())
    with open(os.path.join(tmp, "hb.pfm"), "wb") as f:      # a positive scale line: big-endian samples
        f.write(b"PF\n%d %d\n1.0\n" % (w, h) + hdr[::-1].astype(">f4").tobytes())
    row = (3 * w + 3) & ~3
    with open(os.path.join(tmp, "b.bmp"), "wb") as f:
        f.write(b"BM" + struct.pack("<IHHI", 54 + row * h, 0, 0, 54) + struct.pack("<IiiHHIIiiII", 40, w, h, 1, 24, 0, row * h, 0, 0, 0, 0))
        for y in range(h - 1, -1, -1):
            f.write(rgb[y, :, ::-1].tobytes() + b"\0" * (row - 3 * w))
    with open(os.path.join(tmp, "t.tga"), "wb") as f:
        f.write(struct.pack("<BBBHHBHHHHBB", 0, 0, 2, 0, 0, 0, 0, 0, w, h, 32, 0x28))
        f.write(np.concatenate([rgb[..., ::-1], np.full((h, w, 1), 200, np.uint8)], -1).tobytes())


def test_host_image_decoders_and_mtl_maps(tmp_path, built_lib):
    rng = np.random.default_rng(5)
    w, h = 9, 6
    rgb = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    grey = rng.integers(0, 256, (h, w), dtype=np.uint8)
    hdr = (rng.random((h, w, 3)) * 30).astype(F)
    tmp = str(tmp_path)
    _write_images(tmp, rgb, grey, hdr)
    s = api.HostScene()
    slots = {n: s.load_texture(os.path.join(tmp, n), f) for n, f in (("a.ppm", api.TEX_RGBA8_SRGB), ("b.bmp", api.TEX_RGBA8_SRGB), ("t.tga", api.TEX_RGBA8_UNORM),
                                                                     ("g.pgm", api.TEX_R8_UNORM), ("h.pfm", api.TEX_RGBA8_SRGB))}
    assert s.load_texture(os.path.join(tmp, "a.ppm"), api.TEX_RGBA8_SRGB) == slots["a.ppm"]      # cached per path
    tex = {slot: (tw, th, fmt, data) for slot, tw, th, fmt, data in s.textures()}
    for n in ("a.ppm", "b.bmp"):
        tw, th, fmt, data = tex[slots[n]]
        assert (tw, th, fmt) == (w, h, api.TEX_RGBA8_SRGB)
        px = data.reshape(h, w, 4)
        assert np.array_equal(px[..., :3], rgb) and np.all(px[..., 3] == 255)
    tw, th, fmt, data = tex[slots["t.tga"]]
    assert fmt == api.TEX_RGBA8_UNORM and np.array_equal(data.reshape(h, w, 4)[..., :3], rgb) and np.all(data.reshape(h, w, 4)[..., 3] == 200)
    tw, th, fmt, data = tex[slots["g.pgm"]]
    assert fmt == api.TEX_R8_UNORM and np.array_equal(data.reshape(h, w), grey)
    tw, th, fmt, data = tex[slots["h.pfm"]]
    assert fmt == api.TEX_RGBA32F and np.array_equal(data.view(F).reshape(h, w, 4)[..., :3], hdr)
    big = s.load_texture(os.path.join(tmp, "hb.pfm"), api.TEX_RGBA8_SRGB)
    _, tw, th, fmt, data = [t for t in s.textures() if t[0] == big][0]
    assert fmt == api.TEX_RGBA32F and np.array_equal(data.view(F).reshape(h, w, 4)[..., :3], hdr)
    with pytest.raises(api.GfxError):
        s.load_texture(os.path.join(tmp, "missing.png"))
    # OBJ + MTL with maps: diffuse, bump and emissive maps land in the material
    with open(os.path.join(tmp, "q.mtl"), "w") as f:
        f.write("newmtl Pavement_Brick_BLENDSHADER\nKd 0.5 0.5 0.5\nKs 0.1 0.1 0.1\nNs 100\nmap_Kd a.ppm\nmap_bump -bm 1.0 t.tga\nmap_Ke b.bmp\n")
    with open(os.path.join(tmp, "q.obj"), "w") as f:
        f.write("mtllib q.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0 0\nvt 1 0\nvt 0 1\nusemtl Pavement_Brick_BLENDSHADER\nf 1/1 2/2 3/3\n")
    s.load_obj(os.path.join(tmp, "q.obj"))
    m = s.materials()[-1]
    assert m.texA == slots["a.ppm"] and m.texNormal != 0 and m.texEmittance == slots["b.bmp"] and m.hasEmittance == 1
    assert abs(m.smoothness - int(255 * 0.2) / 255) < 1e-6          # the four Bistro pavement names are pinned to 0.2
    with open(os.path.join(tmp, "bad.obj"), "w") as f:
        f.write("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 9\n")
    with pytest.raises(api.GfxError):
        s.load_obj(os.path.join(tmp, "bad.obj"))
    with open(os.path.join(tmp, "bad2.obj"), "w") as f:
        f.write("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 x3\n")
    with pytest.raises(api.GfxError):
        s.load_obj(os.path.join(tmp, "bad2.obj"))


def test_textured_street_has_every_kind_of_map(built_lib):
    s = util.small_street(textured=True)
    plain = util.small_street()
    assert s.counts() == plain.counts()
    fmts = [t[3] for t in s.textures()]
    assert api.TEX_RGBA8_SRGB in fmts and api.TEX_R8_UNORM in fmts and api.TEX_RGBA8_UNORM in fmts and api.TEX_RGBA32F in fmts
    mats = s.materials()
    assert any(m.texA and m.texSmoothness and m.texNormal for m in mats)
    assert any(m.texEmittance and m.hasEmittance for m in mats)
    assert all(m.texA == 0 and m.texEmittance == 0 for m in plain.materials())
