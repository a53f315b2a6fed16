"""CPU: tools/dds_convert.py, the offline half of the texture path (block-compressed textures decoded offline).

The blocks are built here bit by bit and decoded twice: by the converter (vectorised numpy) and by scalar restatements of the format
definitions written in this file (BC1-BC5), or against values worked out by hand (BC7: one-subset modes 5 and 6, a two-subset mode-1
block, structural checks of the partition / anchor tables); then a .dds goes through the converter and the product's host loader
(gfxh_scene_load_texture) and comes back texel for texel."""
import os
import struct
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import dds_convert as D  # noqa: E402


def _dds(fmt_fourcc, w, h, payload, dx10=None, mips=1):
    hdr = bytearray(128)
    hdr[:4] = b"DDS "
    struct.pack_into("<II", hdr, 4, 124, 0x1007 | (0x20000 if mips > 1 else 0))
    struct.pack_into("<II", hdr, 12, h, w)
    struct.pack_into("<I", hdr, 28, mips)
    struct.pack_into("<II4s", hdr, 76, 32, 0x4, b"DX10" if dx10 else fmt_fourcc)
    out = bytes(hdr)
    if dx10:
        out += struct.pack("<IIIII", dx10, 3, 0, 1, 0)
    return out + bytes(payload)


def _expand565(c):
    r, g, b = (c >> 11) & 31, (c >> 5) & 63, c & 31
    return [(r << 3) | (r >> 2), (g << 2) | (g >> 4), (b << 3) | (b >> 2)]


def _scalar_bc1(block, always_four=False):
    c0, c1, bits = struct.unpack("<HHI", bytes(block))
    p0, p1 = _expand565(c0), _expand565(c1)
    if c0 > c1 or always_four:
        pal = [p0 + [255], p1 + [255], [(2 * a + b) // 3 for a, b in zip(p0, p1)] + [255], [(a + 2 * b) // 3 for a, b in zip(p0, p1)] + [255]]
    else:
        pal = [p0 + [255], p1 + [255], [(a + b) // 2 for a, b in zip(p0, p1)] + [255], [0, 0, 0, 0]]
    return [pal[(bits >> (2 * t)) & 3] for t in range(16)]


def _scalar_alpha(block, signed=False):
    a0, a1 = int(block[0]), int(block[1])
    if signed:
        a0, a1 = max(-127, a0 - 256 if a0 > 127 else a0), max(-127, a1 - 256 if a1 > 127 else a1)
    bits = int.from_bytes(bytes(block[2:8]), "little")
    if a0 > a1:
        pal = [a0, a1] + [((7 - k) * a0 + k * a1) / 7.0 for k in range(1, 7)]
    else:
        pal = [a0, a1] + [((5 - k) * a0 + k * a1) / 5.0 for k in range(1, 5)] + ([-127.0, 127.0] if signed else [0.0, 255.0])
    vals = [pal[(bits >> (3 * t)) & 7] for t in range(16)]
    if signed:
        vals = [(v / 127.0 * 0.5 + 0.5) * 255.0 for v in vals]
    return [int(min(255, max(0, np.floor(v + 0.5)))) for v in vals]


def test_bc1_bc2_bc3_against_a_scalar_restatement():
    rng = np.random.default_rng(4)
    w, h = 12, 8                                     # 3 x 2 blocks
    blocks = rng.integers(0, 256, (6, 8), dtype=np.uint8)
    blocks[1, :4] = [0x10, 0x20, 0x30, 0xF0]         # colour0 < colour1: three colours + transparent
    img, fmt = D.decode(_dds(b"DXT1", w, h, blocks.tobytes()))
    assert fmt == "BC1" and img.shape == (h, w, 4)
    for b in range(6):
        want = _scalar_bc1(blocks[b])
        bx, by = b % 3, b // 3
        for t in range(16):
            assert list(img[4 * by + t // 4, 4 * bx + t % 4]) == want[t], (b, t)
    assert (img[:4, 4:8, 3] == 0).any()               # the transparent entry was used somewhere in block 1 (random indices)
    # BC3: interpolated alpha in front of a four-colour block; BC2: explicit 4-bit alpha
    b16 = rng.integers(0, 256, (6, 16), dtype=np.uint8)
    b16[2, 0], b16[2, 1] = 20, 200                    # alpha0 <= alpha1: six-value palette + 0 and 255
    img3, fmt3 = D.decode(_dds(b"DXT5", w, h, b16.tobytes()))
    img2, fmt2 = D.decode(_dds(b"DXT3", w, h, b16.tobytes()))
    assert (fmt3, fmt2) == ("BC3", "BC2")
    for b in range(6):
        col, alpha = _scalar_bc1(b16[b, 8:], always_four=True), _scalar_alpha(b16[b, :8])
        bx, by = b % 3, b // 3
        for t in range(16):
            px3, px2 = img3[4 * by + t // 4, 4 * bx + t % 4], img2[4 * by + t // 4, 4 * bx + t % 4]
            assert list(px3[:3]) == col[t][:3] and px3[3] == alpha[t], (b, t)
            nibble = (b16[b, t // 2] >> (4 * (t & 1))) & 15
            assert list(px2[:3]) == col[t][:3] and px2[3] == nibble * 17, (b, t)


def test_bc4_bc5_unsigned_and_signed():
    rng = np.random.default_rng(5)
    blocks = rng.integers(0, 256, (4, 16), dtype=np.uint8)
    blocks[0, 0], blocks[0, 1] = 250, 3               # eight-value palette
    blocks[1, 0], blocks[1, 1] = 3, 250               # six-value palette + the two constants
    blocks[2, 0] = 0x80                               # signed: -128 is clamped to -127
    for four, signed in ((b"BC4U", False), (b"BC4S", True)):
        img, _ = D.decode(_dds(four, 8, 8, blocks[:, :8].tobytes()))
        for b in range(4):
            want = _scalar_alpha(blocks[b, :8], signed)
            for t in range(16):
                px = img[4 * (b // 2) + t // 4, 4 * (b % 2) + t % 4]
                assert px[0] == want[t] and px[1] == want[t] and px[3] == 255, (four, b, t)
    for four, dx10, signed in ((b"ATI2", None, False), (b"BC5S", None, True), (None, 83, False)):
        img, fmt = D.decode(_dds(four, 8, 8, blocks.tobytes(), dx10=dx10))
        assert fmt.startswith("BC5")
        for b in range(4):
            wx, wy = _scalar_alpha(blocks[b, :8], signed), _scalar_alpha(blocks[b, 8:], signed)
            for t in range(16):
                px = img[4 * (b // 2) + t // 4, 4 * (b % 2) + t % 4]
                assert (px[0], px[1], px[2], px[3]) == (wx[t], wy[t], 0, 255), (four, b, t)


def _pack(fields):
    """[(value, bits)] least significant first -> 16 bytes."""
    v, pos = 0, 0
    for value, bits in fields:
        assert 0 <= value < (1 << bits)
        v |= value << pos
        pos += bits
    assert pos == 128, pos
    return v.to_bytes(16, "little")


def test_bc7_blocks_worked_out_by_hand():
    # mode 6: one subset, 7-bit RGBA endpoints + one p-bit each, 4-bit indices (texel 0's has 3 bits: the anchor)
    e0, e1 = (10, 20, 30, 127), (100, 90, 80, 0)
    fields = [(1 << 6, 7)]
    for ch in range(4):
        fields += [(e0[ch], 7), (e1[ch], 7)]
    fields += [(1, 1), (0, 1)]                                                   # p-bits
    idx = [0, 15, 5, 10] + [3] * 12
    fields += [(idx[0], 3)] + [(i, 4) for i in idx[1:]]
    img, fmt = D.decode(_dds(None, 4, 4, _pack(fields), dx10=98))
    assert fmt == "BC7"
    full0 = [(c << 1) | 1 for c in e0]
    full1 = [(c << 1) | 0 for c in e1]
    w4 = [0, 4, 9, 13, 17, 21, 26, 30, 34, 38, 43, 47, 51, 55, 60, 64]
    for t in range(16):
        want = [((64 - w4[idx[t]]) * a + w4[idx[t]] * b + 32) >> 6 for a, b in zip(full0, full1)]
        assert list(img[t // 4, t % 4]) == want, t
    # mode 5: 7-bit colour, 8-bit alpha, separate 2-bit index sets, channel rotation 1 (alpha <-> red)
    c0, c1, a0, a1 = (127, 0, 64), (0, 127, 64), 255, 0
    fields = [(1 << 5, 6), (1, 2)]
    for ch in range(3):
        fields += [(c0[ch], 7), (c1[ch], 7)]
    fields += [(a0, 8), (a1, 8)]
    ci = [0, 3, 1, 2] * 4
    ai = [1, 0, 3, 2] * 4
    fields += [(ci[0], 1)] + [(i, 2) for i in ci[1:]]
    fields += [(ai[0], 1)] + [(i, 2) for i in ai[1:]]
    img, _ = D.decode(_dds(None, 4, 4, _pack(fields), dx10=99))
    w2 = [0, 21, 43, 64]
    ex = lambda c: (c << 1) | (c >> 6)                                          # 7 -> 8 bits by bit replication
    for t in range(16):
        rgb = [((64 - w2[ci[t]]) * ex(x) + w2[ci[t]] * ex(y) + 32) >> 6 for x, y in zip(c0, c1)]
        a = ((64 - w2[ai[t]]) * a0 + w2[ai[t]] * a1 + 32) >> 6
        assert list(img[t // 4, t % 4]) == [a, rgb[1], rgb[2], rgb[0]], t      # rotation 1: red and alpha change places
    # mode 1: two subsets (partition 13: the upper half / the lower half), 6-bit colours + a shared p-bit per subset, 3-bit indices; every
    # index 0 except the last texel: each half shows its subset's first endpoint, the last texel the second endpoint of subset 1
    assert D._P2[13] == "0000000011111111" and D._A2[13] == 15
    ends = [(63, 0, 0), (0, 63, 0), (0, 0, 63), (63, 63, 63)]                    # subset 0: e0, e1; subset 1: e0, e1
    fields = [(1 << 1, 2), (13, 6)]
    for ch in range(3):
        fields += [(e[ch], 6) for e in ends]
    fields += [(1, 1), (0, 1)]                                                   # p-bit of subset 0, of subset 1
    idx = [0] * 15 + [3]                                                         # anchors (texels 0 and 15) have 2 bits: 3 = the highest of them...
    fields += [(idx[t], 2 if t in (0, 15) else 3) for t in range(16)]
    img, _ = D.decode(_dds(None, 4, 4, _pack(fields), dx10=98))
    up = lambda c, p: (((c << 1) | p) << 1) | (((c << 1) | p) >> 6)              # 6 bits + p-bit -> 7 -> 8 by replication
    w3 = [0, 9, 18, 27, 37, 46, 55, 64]
    top, bottom0, bottom1 = [up(c, 1) for c in ends[0]], [up(c, 0) for c in ends[2]], [up(c, 0) for c in ends[3]]
    for t in range(8):
        assert list(img[t // 4, t % 4]) == top + [255], t
    for t in range(8, 15):
        assert list(img[t // 4, t % 4]) == bottom0 + [255], t
    last = [((64 - w3[3]) * a + w3[3] * b + 32) >> 6 for a, b in zip(bottom0, bottom1)]
    assert list(img[3, 3]) == last + [255]


def test_bc7_partition_and_anchor_tables_are_consistent():
    """The fix-up (anchor) index of a subset is a texel OF that subset, texel 0 is always in subset 0, every subset of a partition is used:
    192 constraints between tables that were written down separately."""
    for p in range(64):
        two, three = D._P2[p], D._P3[p]
        assert len(two) == 16 and set(two) == {"0", "1"} and two[0] == "0" and two[D._A2[p]] == "1", p
        assert len(three) == 16 and set(three) == {"0", "1", "2"} and three[0] == "0", p
        assert three[D._A3a[p]] == "1" and three[D._A3b[p]] == "2", p
    assert len(set(D._P2)) == 64 and len(set(D._P3)) == 64


def test_headers_mips_and_the_way_into_the_host_loader(built_lib, tmp_path):
    from gfxexp_amd import api
    rng = np.random.default_rng(6)
    # two mip levels of a 8 x 8 BC1 texture: the second starts 32 bytes behind the first
    m0, m1 = rng.integers(0, 256, (4, 8), dtype=np.uint8), rng.integers(0, 256, (1, 8), dtype=np.uint8)
    data = _dds(b"DXT1", 8, 8, m0.tobytes() + m1.tobytes(), mips=2)
    assert D.parse_header(data) == ("BC1", 8, 8, 2, 128)
    img1, _ = D.decode(data, mip=1)
    assert img1.shape == (4, 4, 4) and [list(px) for px in img1.reshape(16, 4)] == _scalar_bc1(m1[0])
    with pytest.raises(ValueError):
        D.decode(data, mip=2)
    with pytest.raises(ValueError):
        D.decode(_dds(None, 4, 4, bytes(16), dx10=95))        # BC6H
    with pytest.raises(ValueError):
        D.decode(b"not a dds file" * 20)
    # uncompressed BGRA8 through the masks
    hdr = bytearray(_dds(b"\\0\\0\\0\\0", 2, 2, b""))
    struct.pack_into("<II4sIIIII", hdr, 76, 32, 0x41, b"\\0\\0\\0\\0", 32, 0xFF0000, 0xFF00, 0xFF, 0xFF000000)
    img, fmt = D.decode(bytes(hdr) + bytes([1, 2, 3, 4] * 4))
    assert fmt == "BGRA8" and list(img[0, 0]) == [3, 2, 1, 4]
    # .dds -> .tga -> gfxh_scene_load_texture: the texels the renderer will sample
    img0, _ = D.decode(data)
    dds_path, tga_path = tmp_path / "t.dds", tmp_path / "t.tga"
    dds_path.write_bytes(data)
    assert D.main([str(dds_path), str(tga_path)]) == 0
    hs = api.HostScene()
    slot = hs.load_texture(str(tga_path), api.TEX_RGBA8_UNORM)
    (s, w, h, f, texels), = [t for t in hs.textures() if t[0] == slot]
    assert (w, h, f) == (8, 8, api.TEX_RGBA8_UNORM)
    assert np.array_equal(texels.reshape(8, 8, 4), img0)


# ---------------------------------------------------------------- PNG
def _png_chunk(kind, body):
    import zlib
    return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body) & 0xFFFFFFFF)


def _png(rows, w, h, depth, ctype, filters, palette=None, trns=None):
    """A PNG file from already packed sample rows (bytes per row), the row filters applied here -- the forward direction of the
    specification's five filters, written independently of the converter's reconstruction."""
    import zlib
    channels = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    bpp = max(1, channels * depth // 8)
    out = bytearray()
    prev = bytes(len(rows[0]))
    for y, row in enumerate(rows):
        ft = filters[y % len(filters)]
        enc = bytearray(len(row))
        for i, x in enumerate(row):
            a = row[i - bpp] if i >= bpp else 0
            b = prev[i]
            c = prev[i - bpp] if i >= bpp else 0
            if ft == 0:
                pred = 0
            elif ft == 1:
                pred = a
            elif ft == 2:
                pred = b
            elif ft == 3:
                pred = (a + b) // 2
            else:
                p = a + b - c
                pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                pred = a if pa <= pb and pa <= pc else b if pb <= pc else c
            enc[i] = (x - pred) & 255
        out += bytes([ft]) + enc
        prev = row
    comp = zlib.compress(bytes(out), 6)
    data = b"\x89PNG\r\n\x1a\n" + _png_chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 0))
    if palette is not None:
        data += _png_chunk(b"PLTE", bytes(palette))
    if trns is not None:
        data += _png_chunk(b"tRNS", bytes(trns))
    half = len(comp) // 2                                     # two IDAT chunks: the stream may be split anywhere
    return data + _png_chunk(b"IDAT", comp[:half]) + _png_chunk(b"IDAT", comp[half:]) + _png_chunk(b"IEND", b"")


def test_png_every_colour_type_and_filter():
    rng = np.random.default_rng(11)
    w, h = 13, 10
    filters = [0, 1, 2, 3, 4, 4, 3, 1]
    rgba = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    # RGBA 8, RGB 8, grey 8, grey + alpha 8
    img, _ = D.decode_png(_png([bytes(r) for r in rgba.reshape(h, -1)], w, h, 8, 6, filters))
    assert np.array_equal(img, rgba)
    img, _ = D.decode_png(_png([bytes(r) for r in rgba[..., :3].reshape(h, -1)], w, h, 8, 2, filters))
    assert np.array_equal(img[..., :3], rgba[..., :3]) and np.all(img[..., 3] == 255)
    img, _ = D.decode_png(_png([bytes(r) for r in rgba[..., 0]], w, h, 8, 0, filters))
    assert np.array_equal(img[..., 0], rgba[..., 0]) and np.array_equal(img[..., 1], img[..., 0]) and np.array_equal(img[..., 2], img[..., 0])
    img, _ = D.decode_png(_png([bytes(r) for r in rgba[..., :2].reshape(h, -1)], w, h, 8, 4, filters))
    assert np.array_equal(img[..., 0], rgba[..., 0]) and np.array_equal(img[..., 3], rgba[..., 1])
    # RGB 16: the high byte of every big-endian sample
    wide = rng.integers(0, 65536, (h, w, 3), dtype=np.uint16)
    img, _ = D.decode_png(_png([r.astype(">u2").tobytes() for r in wide.reshape(h, -1)], w, h, 16, 2, filters))
    assert np.array_equal(img[..., :3], (wide >> 8).astype(np.uint8))
    # palette, 4 bits per index, with tRNS for the first three entries
    palette = rng.integers(0, 256, (16, 3), dtype=np.uint8)
    idx = rng.integers(0, 16, (h, w), dtype=np.uint8)
    packed = []
    for r in idx:
        padded = np.concatenate([r, np.zeros((-len(r)) % 2, np.uint8)])
        packed.append(bytes((padded[0::2] << 4) | padded[1::2]))
    img, _ = D.decode_png(_png(packed, w, h, 4, 3, filters, palette=palette.reshape(-1), trns=[0, 128, 200]))
    assert np.array_equal(img[..., :3], palette[idx])
    assert np.array_equal(img[..., 3], np.array([0, 128, 200] + [255] * 13, np.uint8)[idx])
    # grey, 1 bit per pixel
    bits = rng.integers(0, 2, (h, w), dtype=np.uint8)
    img, _ = D.decode_png(_png([bytes(np.packbits(r)) for r in bits], w, h, 1, 0, filters))
    assert np.array_equal(img[..., 0], bits * 255)
    with pytest.raises(ValueError):
        D.decode_png(b"\x89PNG\r\n\x1a\n" + _png_chunk(b"IHDR", struct.pack(">IIBBBBB", 4, 4, 8, 6, 0, 0, 1)) + _png_chunk(b"IEND", b""))   # Adam7
    good = _png([bytes(r) for r in rgba.reshape(h, -1)], w, h, 8, 6, filters)
    at = good.index(b"IDAT") + 12
    with pytest.raises(ValueError):
        D.decode_png(good[:at] + bytes(8) + good[at + 8:])                  # a damaged deflate stream is a ValueError, not a crash of --dir
    with pytest.raises(ValueError):
        D.decode_png(good[:40])                                             # truncated after the header: no image data


def test_png_through_the_converter_and_the_host_loader(tmp_path, built_lib):
    from gfxexp_amd import api
    rng = np.random.default_rng(12)
    w, h = 24, 16
    rgba = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    src, dst = str(tmp_path / "albedo.png"), str(tmp_path / "albedo.tga")
    with open(src, "wb") as f:
        f.write(_png([bytes(r) for r in rgba.reshape(h, -1)], w, h, 8, 6, [4, 1, 3, 2, 0]))
    assert D.main([src, dst]) == 0
    s = api.HostScene()
    slot = s.load_texture(dst, api.TEX_RGBA8_UNORM)
    (_, tw, th, fmt, data), = [t for t in s.textures() if t[0] == slot]
    assert (tw, th, fmt) == (w, h, api.TEX_RGBA8_UNORM) and np.array_equal(data.reshape(h, w, 4), rgba)
    assert D.main(["--dir", str(tmp_path)]) == 0                  # the directory form takes .png as it takes .dds
