#!/usr/bin/env python
"""Offline half of the texture path ("BC-decoded offline"): decodes a .dds file to an image the host loader reads.

The reference uploads block-compressed DDS textures as they are and lets the texture unit decode them (common/dds_loader.cpp:207-346 reads the
header -- the classic FourCC codes DXT1 / DXT3 / DXT5 / BC4U / BC4S / ATI2 / BC5U / BC5S or the DX10 extension with a DXGI format -- and
common/common_host.cpp:1163-1244 creates the CUDA array).  This build samples textures in software from uncompressed texels, so the blocks
are decoded once, offline: BC1, BC2, BC3 (RGBA), BC4 (one channel), BC5 (two channels: tangent-space normal maps, z rebuilt by the
2-channel reader of the renderer), BC7 (all eight modes) and the uncompressed 32-bit RGBA / BGRA layouts; BC6H (HDR) is not decoded.
Output by extension: .tga (RGBA8, what `gfxh_scene_load_texture` reads with alpha), .ppm (RGB8), .pgm (first channel).
sRGB variants decode to the same bytes (the loader applies the sRGB table according to how the material uses the texture).

    python tools/dds_convert.py in.dds out.tga [--mip 0] [--dir DIR: convert every .dds / .png under DIR next to itself as .tga]

PNG (the other format the reference's assets use, read there through stb_image / assimp: common/common_host.cpp:1246-1313) converts the
same way: `python tools/dds_convert.py in.png out.tga`.  Non-interlaced files of every colour type (grey, RGB, palette with tRNS,
grey + alpha, RGBA) and bit depth (1 / 2 / 4 / 8 / 16: 16-bit samples keep their high byte); zlib comes with Python.
"""
import os
import struct
import sys
import zlib

import numpy as np

DXGI = {71: "BC1", 72: "BC1", 74: "BC2", 75: "BC2", 77: "BC3", 78: "BC3", 80: "BC4U", 81: "BC4S", 83: "BC5U", 84: "BC5S", 95: "BC6H", 96: "BC6H",
        98: "BC7", 99: "BC7", 28: "RGBA8", 29: "RGBA8", 87: "BGRA8", 91: "BGRA8"}
FOURCC = {b"DXT1": "BC1", b"DXT3": "BC2", b"DXT5": "BC3", b"BC4U": "BC4U", b"ATI1": "BC4U", b"BC4S": "BC4S", b"ATI2": "BC5U", b"BC5U": "BC5U",
          b"BC5S": "BC5S"}
BLOCK_BYTES = {"BC1": 8, "BC4U": 8, "BC4S": 8, "BC2": 16, "BC3": 16, "BC5U": 16, "BC5S": 16, "BC7": 16, "BC6H": 16}


def parse_header(data):
    """(format name, width, height, mip count, offset of the first mip) of a DDS file (dds_loader.cpp:207-300)."""
    if len(data) < 128 or data[:4] != b"DDS ":
        raise ValueError("not a DDS file")
    height, width = struct.unpack_from("<II", data, 12)
    mips = max(1, struct.unpack_from("<I", data, 28)[0])
    pf_flags, fourcc = struct.unpack_from("<I4s", data, 80)
    bit_count, rmask, gmask, bmask, amask = struct.unpack_from("<IIIII", data, 88)
    offset = 128
    if pf_flags & 0x4 and fourcc == b"DX10":
        dxgi = struct.unpack_from("<I", data, 128)[0]
        offset += 20
        if dxgi not in DXGI:
            raise ValueError("DXGI format %d is not handled" % dxgi)
        return DXGI[dxgi], width, height, mips, offset
    if pf_flags & 0x4:
        if fourcc not in FOURCC:
            raise ValueError("FourCC %r is not handled" % fourcc)
        return FOURCC[fourcc], width, height, mips, offset
    if bit_count == 32 and (rmask, gmask, bmask) == (0xFF, 0xFF00, 0xFF0000):
        return "RGBA8", width, height, mips, offset
    if bit_count == 32 and (rmask, gmask, bmask) == (0xFF0000, 0xFF00, 0xFF):
        return "BGRA8", width, height, mips, offset
    raise ValueError("uncompressed layout with %d bits and masks %x %x %x %x is not handled" % (bit_count, rmask, gmask, bmask, amask))


def mip_extent(width, height, mip):
    return max(1, width >> mip), max(1, height >> mip)


def mip_bytes(fmt, w, h):
    if fmt in BLOCK_BYTES:
        return ((w + 3) // 4) * ((h + 3) // 4) * BLOCK_BYTES[fmt]
    return 4 * w * h


# ---------------------------------------------------------------- BC1 / BC2 / BC3 colour blocks
def _rgb565(c):
    r, g, b = (c >> 11) & 31, (c >> 5) & 63, c & 31
    return np.stack([(r << 3) | (r >> 2), (g << 2) | (g >> 4), (b << 3) | (b >> 2)], -1).astype(np.int32)


def _color_block(blocks8, four_colour_only):
    """blocks8: [n, 8] bytes -> [n, 16, 4] RGBA8 (texel 4 y + x).  BC1: colour0 <= colour1 selects the 3-colour + transparent palette."""
    c0 = blocks8[:, 0].astype(np.uint32) | (blocks8[:, 1].astype(np.uint32) << 8)
    c1 = blocks8[:, 2].astype(np.uint32) | (blocks8[:, 3].astype(np.uint32) << 8)
    bits = (blocks8[:, 4].astype(np.uint32) | (blocks8[:, 5].astype(np.uint32) << 8) | (blocks8[:, 6].astype(np.uint32) << 16)
            | (blocks8[:, 7].astype(np.uint32) << 24))
    p0, p1 = _rgb565(c0), _rgb565(c1)
    four = (c0 > c1) | four_colour_only
    pal = np.zeros((len(blocks8), 4, 4), np.int32)
    pal[:, 0, :3], pal[:, 1, :3] = p0, p1
    pal[:, :, 3] = 255
    pal[:, 2, :3] = np.where(four[:, None], (2 * p0 + p1) // 3, (p0 + p1) // 2)
    pal[:, 3, :3] = np.where(four[:, None], (p0 + 2 * p1) // 3, 0)
    pal[:, 3, 3] = np.where(four, 255, 0)
    idx = (bits[:, None] >> (2 * np.arange(16, dtype=np.uint32))[None, :]) & 3
    return np.take_along_axis(pal[:, None, :, :].repeat(16, 1), idx[:, :, None, None].astype(np.int64).repeat(4, 3), 2)[:, :, 0, :].astype(np.uint8)


def _alpha_block(blocks8, signed=False):
    """BC3 alpha / BC4 / one half of BC5: [n, 8] bytes -> [n, 16] values (0..255; signed blocks are remapped from [-127, 127])."""
    a0, a1 = blocks8[:, 0].astype(np.int32), blocks8[:, 1].astype(np.int32)
    if signed:
        a0 = np.where(a0 > 127, a0 - 256, a0); a1 = np.where(a1 > 127, a1 - 256, a1)
        a0 = np.maximum(a0, -127); a1 = np.maximum(a1, -127)
    bits = np.zeros(len(blocks8), np.uint64)
    for k in range(6):
        bits |= blocks8[:, 2 + k].astype(np.uint64) << np.uint64(8 * k)
    pal = np.zeros((len(blocks8), 8), np.float64)
    pal[:, 0], pal[:, 1] = a0, a1
    eight = a0 > a1
    for k in range(1, 7):
        pal[:, 1 + k] = np.where(eight, ((7 - k) * a0 + k * a1) / 7.0, np.where(k <= 4, ((5 - k) * a0 + k * a1) / 5.0, 0))
    lo, hi = (-127.0, 127.0) if signed else (0.0, 255.0)
    pal[:, 6] = np.where(eight, pal[:, 6], lo)
    pal[:, 7] = np.where(eight, pal[:, 7], hi)
    idx = ((bits[:, None] >> (np.uint64(3) * np.arange(16, dtype=np.uint64))[None, :]) & np.uint64(7)).astype(np.int64)
    v = np.take_along_axis(pal, idx, 1)
    if signed:
        v = (v / 127.0 * 0.5 + 0.5) * 255.0
    return np.clip(np.floor(v + 0.5), 0, 255).astype(np.uint8)


# ---------------------------------------------------------------- BC7 (scalar per block; the eight modes of the format specification)
_BC7_MODES = [  # subsets, partition bits, rotation bits, index-selection bits, colour bits, alpha bits, endpoint p-bits, shared p-bits, index bits, 2nd index bits
    (3, 4, 0, 0, 4, 0, 1, 0, 3, 0), (2, 6, 0, 0, 6, 0, 0, 1, 3, 0), (3, 6, 0, 0, 5, 0, 0, 0, 2, 0), (2, 6, 0, 0, 7, 0, 1, 0, 2, 0),
    (1, 0, 2, 1, 5, 6, 0, 0, 2, 3), (1, 0, 2, 0, 7, 8, 0, 0, 2, 2), (1, 0, 0, 0, 7, 7, 1, 0, 4, 0), (2, 6, 0, 0, 5, 5, 1, 0, 2, 0)]
_W2, _W3, _W4 = [0, 21, 43, 64], [0, 9, 18, 27, 37, 46, 55, 64], [0, 4, 9, 13, 17, 21, 26, 30, 34, 38, 43, 47, 51, 55, 60, 64]
_P2 = ("""0011001100110011 0001000100010001 0111011101110111 0001001100110111 0000000100010011 0011011101111111 0001001101111111 0000000100110111
0000000000010011 0011011111111111 0000000101111111 0000000000010111 0001011111111111 0000000011111111 0000111111111111 0000000000001111
0000100011101111 0111000100000000 0000000010001110 0111001100010000 0011000100000000 0000100011001110 0000000010001100 0111001100110001
0011000100010000 0000100010001100 0110011001100110 0011011001101100 0001011111101000 0000111111110000 0111000110001110 0011100110011100
0101010101010101 0000111100001111 0101101001011010 0011001111001100 0011110000111100 0101010110101010 0110100101101001 0101101010100101
0111001111001110 0001001111001000 0011001001001100 0011101111011100 0110100110010110 0011110011000011 0110011010011001 0000011001100000
0100111001000000 0010011100100000 0000001001110010 0000010011100100 0110110010010011 0011011011001001 0110001110011100 0011100111000110
0110110011001001 0110001100111001 0111111010000001 0001100011100111 0000111100110011 0011001111110000 0010001011101110 0100010001110111""").split()
_P3 = ("""0011001102212222 0001001122112221 0000200122112211 0222002200110111 0000000011221122 0011001100220022 0022002211111111 0011001122112211
0000000011112222 0000111111112222 0000111122222222 0012001200120012 0112011201120112 0122012201220122 0011011211221222 0011200122002220
0001001101121122 0111001120012200 0000112211221122 0022002200221111 0111011102220222 0001000122212221 0000001101220122 0000110022102210
0122012200110000 0012001211222222 0110122112210110 0000011012211221 0022110211020022 0110011020022222 0011012201220011 0000200022112221
0000000211221222 0222002200120011 0011001200220222 0120012001200120 0000111122220000 0120120120120120 0120201212010120 0011220011220011
0011112222000011 0101010122222222 0000000021212121 0022112200221122 0022001100220011 0220122102201221 0101222222220101 0000212121212121
0101010101012222 0222011102220111 0002111200021112 0000211221122112 0222011101110222 0002111211120002 0110011001102222 0000000021122112
0110011022222222 0022001100110022 0022112211220022 0000000000002112 0002000100020001 0222122202221222 0101222222222222 0111201122012220""").split()
_A2 = [15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 2, 8, 2, 2, 8, 8, 15, 2, 8, 2, 2, 8, 8, 2, 2, 15, 15, 6, 8, 2, 8, 15, 15, 2, 8, 2, 2, 2, 15, 15, 6,
       6, 2, 6, 8, 15, 15, 2, 2, 15, 15, 15, 15, 15, 2, 2, 15]
_A3a = [3, 3, 15, 15, 8, 3, 15, 15, 8, 8, 6, 6, 6, 5, 3, 3, 3, 3, 8, 15, 3, 3, 6, 10, 5, 8, 8, 6, 8, 5, 15, 15, 8, 15, 3, 5, 6, 10, 8, 15, 15, 3, 15, 5, 15, 15, 15, 15,
        3, 15, 5, 5, 5, 8, 5, 10, 5, 10, 8, 13, 15, 12, 3, 3]
_A3b = [15, 8, 8, 3, 15, 15, 3, 8, 15, 15, 15, 15, 15, 15, 15, 8, 15, 8, 15, 3, 15, 8, 15, 8, 3, 15, 6, 10, 15, 15, 10, 8, 15, 3, 15, 10, 10, 8, 9, 10, 6, 15, 8, 15, 3, 6, 6, 8,
        15, 3, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 3, 15, 15, 8]


def _bc7_block(block):
    v = int.from_bytes(bytes(block), "little")
    pos = 0

    def take(n):
        nonlocal pos
        r = (v >> pos) & ((1 << n) - 1)
        pos += n
        return r
    mode = 0
    while mode < 8 and not take(1):
        mode += 1
    if mode == 8:
        return np.zeros((16, 4), np.uint8)
    ns, pb, rb, isb, cb, ab, epb, spb, ib, ib2 = _BC7_MODES[mode]
    part, rot, isel = take(pb), take(rb), take(isb)
    ends = np.zeros((2 * ns, 4), np.int64)
    for ch in range(3):
        for e in range(2 * ns):
            ends[e, ch] = take(cb)
    for e in range(2 * ns):
        ends[e, 3] = take(ab) if ab else 0
    cbits, abits = cb, ab
    if epb:
        for e in range(2 * ns):
            p = take(1)
            ends[e, :3] = (ends[e, :3] << 1) | p
            if ab:
                ends[e, 3] = (ends[e, 3] << 1) | p
        cbits += 1; abits += 1 if ab else 0
    elif spb:
        ps = [take(1), take(1)]
        for e in range(2 * ns):
            ends[e, :3] = (ends[e, :3] << 1) | ps[e >> 1]
        cbits += 1
    ends[:, :3] = (ends[:, :3] << (8 - cbits)) | (ends[:, :3] >> (2 * cbits - 8))
    ends[:, 3] = ((ends[:, 3] << (8 - abits)) | (ends[:, 3] >> (2 * abits - 8))) if ab else 255
    subset = [0] * 16 if ns == 1 else [int(c) for c in (_P2 if ns == 2 else _P3)[part]]
    anchors = {0} if ns == 1 else ({0, _A2[part]} if ns == 2 else {0, _A3a[part], _A3b[part]})
    idx1 = [take(ib - 1 if t in anchors else ib) for t in range(16)]
    idx2 = [take(ib2 - 1 if t == 0 else ib2) for t in range(16)] if ib2 else idx1
    w1 = {2: _W2, 3: _W3, 4: _W4}[ib]
    w2 = {2: _W2, 3: _W3, 4: _W4}[ib2] if ib2 else w1
    out = np.zeros((16, 4), np.uint8)
    for t in range(16):
        e0, e1 = ends[2 * subset[t]], ends[2 * subset[t] + 1]
        ci, ai = (idx2[t], idx1[t]) if isel else (idx1[t], idx2[t])
        cw, aw = (w2, w1) if isel else (w1, w2)
        rgb = [((64 - cw[ci]) * int(e0[c]) + cw[ci] * int(e1[c]) + 32) >> 6 for c in range(3)]
        a = ((64 - aw[ai]) * int(e0[3]) + aw[ai] * int(e1[3]) + 32) >> 6
        px = rgb + [a]
        if rot:
            px[rot - 1], px[3] = px[3], px[rot - 1]
        out[t] = px
    return out


# ---------------------------------------------------------------- whole images
def _blocks_to_image(texels, w, h):
    """[blocks, 16, C] (blocks in row-major order of 4 x 4 tiles) -> [h, w, C]."""
    bw, bh = (w + 3) // 4, (h + 3) // 4
    c = texels.shape[2]
    img = texels.reshape(bh, bw, 4, 4, c).transpose(0, 2, 1, 3, 4).reshape(bh * 4, bw * 4, c)
    return img[:h, :w]


def decode(data, mip=0):
    """DDS bytes -> (RGBA8 image [h, w, 4], format name).  One- and two-channel formats fill (v, v, v, 255) / (x, y, 0, 255)."""
    fmt, width, height, mips, offset = parse_header(data)
    if mip >= mips:
        raise ValueError("the file has %d mip levels" % mips)
    for m in range(mip):
        offset += mip_bytes(fmt, *mip_extent(width, height, m))
    w, h = mip_extent(width, height, mip)
    raw = np.frombuffer(data, np.uint8, mip_bytes(fmt, w, h), offset)
    if fmt in ("RGBA8", "BGRA8"):
        img = raw.reshape(h, w, 4).copy()
        return (img[:, :, [2, 1, 0, 3]] if fmt == "BGRA8" else img), fmt
    if fmt == "BC6H":
        raise ValueError("BC6H (HDR) blocks are not decoded; convert with the asset's authoring tool to .pfm")
    blocks = raw.reshape(-1, BLOCK_BYTES[fmt])
    if fmt == "BC1":
        tex = _color_block(blocks, False)
    elif fmt == "BC2":
        tex = _color_block(blocks[:, 8:], True)
        nib = np.stack([blocks[:, k // 2] >> (4 * (k & 1)) & 15 for k in range(16)], 1)
        tex[:, :, 3] = (nib * 17).astype(np.uint8)
    elif fmt == "BC3":
        tex = _color_block(blocks[:, 8:], True)
        tex[:, :, 3] = _alpha_block(blocks[:, :8])
    elif fmt in ("BC4U", "BC4S"):
        v = _alpha_block(blocks, fmt == "BC4S")
        tex = np.stack([v, v, v, np.full_like(v, 255)], -1)
    elif fmt in ("BC5U", "BC5S"):
        x, y = _alpha_block(blocks[:, :8], fmt == "BC5S"), _alpha_block(blocks[:, 8:], fmt == "BC5S")
        tex = np.stack([x, y, np.zeros_like(x), np.full_like(x, 255)], -1)
    else:
        tex = np.stack([_bc7_block(b) for b in blocks], 0)
    return _blocks_to_image(tex, w, h), fmt


def decode_png(data):
    """(H, W, 4) uint8 RGBA of a non-interlaced PNG file (PNG specification, ISO/IEC 15948: chunks, the five row filters, the colour types)."""
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError("not a PNG file")
    at, idat, palette, trns, ihdr = 8, [], None, None, None
    while at + 8 <= len(data):
        n, kind = struct.unpack_from(">I4s", data, at)
        body = data[at + 8:at + 8 + n]
        at += 12 + n
        if kind == b"IHDR":
            ihdr = struct.unpack(">IIBBBBB", body)
        elif kind == b"PLTE":
            palette = np.frombuffer(body, np.uint8).reshape(-1, 3)
        elif kind == b"tRNS":
            trns = np.frombuffer(body, np.uint8)
        elif kind == b"IDAT":
            idat.append(body)
        elif kind == b"IEND":
            break
    if ihdr is None:
        raise ValueError("PNG without IHDR")
    w, h, depth, ctype, _, _, interlace = ihdr
    if interlace:
        raise ValueError("interlaced PNG is not handled")
    channels = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}.get(ctype)
    if channels is None or depth not in (1, 2, 4, 8, 16) or (ctype == 3 and palette is None):
        raise ValueError("PNG colour type %d / depth %d is not handled" % (ctype, depth))
    bpp = max(1, channels * depth // 8)                      # filter distance in bytes
    stride = (w * channels * depth + 7) // 8
    try:
        raw = zlib.decompress(b"".join(idat))
    except zlib.error as e:
        raise ValueError("corrupt PNG data (%s)" % e)
    if len(raw) < h * (stride + 1):
        raise ValueError("truncated PNG data")
    rows = np.zeros((h, stride), np.uint8)
    prev = bytearray(stride)
    for y in range(h):
        ft = raw[y * (stride + 1)]
        cur = bytearray(raw[y * (stride + 1) + 1:(y + 1) * (stride + 1)])
        if ft == 1:                                             # Sub
            for i in range(bpp, stride):
                cur[i] = (cur[i] + cur[i - bpp]) & 255
        elif ft == 2:                                           # Up
            cur = bytearray(((np.frombuffer(bytes(cur), np.uint8).astype(np.uint16) + np.frombuffer(bytes(prev), np.uint8)) & 255).astype(np.uint8).tobytes())
        elif ft == 3:                                           # Average
            for i in range(stride):
                left = cur[i - bpp] if i >= bpp else 0
                cur[i] = (cur[i] + ((left + prev[i]) >> 1)) & 255
        elif ft == 4:                                           # Paeth
            for i in range(stride):
                a = cur[i - bpp] if i >= bpp else 0
                b = prev[i]
                c = prev[i - bpp] if i >= bpp else 0
                p = a + b - c
                pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                cur[i] = (cur[i] + (a if pa <= pb and pa <= pc else b if pb <= pc else c)) & 255
        elif ft != 0:
            raise ValueError("PNG row filter %d does not exist" % ft)
        rows[y] = np.frombuffer(bytes(cur), np.uint8)
        prev = cur
    if depth == 16:
        samples = rows.reshape(h, w * channels, 2)[:, :, 0]     # big-endian: the high byte
    elif depth == 8:
        samples = rows
    else:                                                       # 1 / 2 / 4 bits, most significant first
        bits = np.unpackbits(rows, axis=1)[:, :w * channels * depth].reshape(h, w * channels, depth)
        samples = (bits * (1 << np.arange(depth - 1, -1, -1, dtype=np.uint16))).sum(-1).astype(np.uint16)
        if ctype != 3:
            samples = samples * 255 // ((1 << depth) - 1)       # grey levels scale to 0 .. 255
        samples = samples.astype(np.uint8)
    samples = samples.reshape(h, w, channels)
    img = np.full((h, w, 4), 255, np.uint8)
    if ctype == 0:
        img[..., :3] = samples
        if trns is not None and len(trns) >= 2 and depth <= 8:
            img[..., 3] = np.where(samples[..., 0] == (int(trns[1]) * 255 // ((1 << depth) - 1) if depth < 8 else trns[1]), 0, 255)
    elif ctype == 2:
        img[..., :3] = samples
        if trns is not None and len(trns) >= 6 and depth == 8:
            img[..., 3] = np.where((samples == trns[1:6:2]).all(-1), 0, 255)
    elif ctype == 3:
        idx = samples[..., 0].astype(np.int64)
        if idx.max(initial=0) >= len(palette):
            raise ValueError("PNG palette index out of range")
        img[..., :3] = palette[idx]
        if trns is not None:
            alpha = np.full(len(palette), 255, np.uint8)
            alpha[:len(trns)] = trns[:len(palette)]
            img[..., 3] = alpha[idx]
    elif ctype == 4:
        img[..., :3] = samples[..., :1]
        img[..., 3] = samples[..., 1]
    else:
        img[...] = samples
    return img, "PNG %d-bit colour type %d" % (depth, ctype)


def decode_any(data, mip=0):
    return decode_png(data) if data[:4] == b"\x89PNG" else decode(data, mip)


def write_image(path, img):
    h, w = img.shape[:2]
    ext = os.path.splitext(path)[1].lower()
    with open(path, "wb") as f:
        if ext == ".tga":      # uncompressed true-colour, 32 bits, top-left origin
            f.write(struct.pack("<BBBHHBHHHHBB", 0, 0, 2, 0, 0, 0, 0, 0, w, h, 32, 0x28))
            f.write(np.ascontiguousarray(img[:, :, [2, 1, 0, 3]]).tobytes())
        elif ext == ".ppm":
            f.write(b"P6\n%d %d\n255\n" % (w, h)); f.write(np.ascontiguousarray(img[:, :, :3]).tobytes())
        elif ext == ".pgm":
            f.write(b"P5\n%d %d\n255\n" % (w, h)); f.write(np.ascontiguousarray(img[:, :, 0]).tobytes())
        else:
            raise ValueError("output extension must be .tga, .ppm or .pgm")


def main(argv):
    mip = int(argv[argv.index("--mip") + 1]) if "--mip" in argv else 0
    if "--dir" in argv:
        root = argv[argv.index("--dir") + 1]
        for dp, _, fns in os.walk(root):
            for fn in sorted(fns):
                if fn.lower().endswith((".dds", ".png")):
                    src = os.path.join(dp, fn)
                    try:
                        img, fmt = decode_any(open(src, "rb").read(), mip)
                        write_image(os.path.splitext(src)[0] + ".tga", img)
                        print("%s: %s %dx%d" % (src, fmt, img.shape[1], img.shape[0]))
                    except ValueError as e:
                        print("%s: skipped (%s)" % (src, e))
        return 0
    if len(argv) < 2:
        print(__doc__)
        return 1
    img, fmt = decode_any(open(argv[0], "rb").read(), mip)
    write_image(argv[1], img)
    print("%s: %s %dx%d -> %s" % (argv[0], fmt, img.shape[1], img.shape[0], argv[1]))
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
