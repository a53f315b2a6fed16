"""CPU: the host loader's OpenEXR reader (gfxexp_amd/csrc/host/scene_builder.cpp decode_exr) -- the format the reference reads its
environment texture from (loadEnvTexture -> tinyexr LoadEXR, common/common_host.cpp:2674).

The files are written here, byte by byte from the OpenEXR file-layout document, with Python's zlib as the compressor -- an encoder that
shares no code with the reader and its own inflate: scanline files, compression NONE / RLE / ZIPS / ZIP, HALF / FLOAT / UINT channels,
RGBA / RGB / Y channel sets, data windows that do not start at the origin, both line orders, chunks that did not shrink and are stored
as they are; plus the round trip through the product's own EXR writer (gfxh_save_image_hdr)."""
import struct
import zlib

import numpy as np
import pytest

from gfxexp_amd import api

HALF, FLOAT, UINT = 1, 2, 0
NONE, RLE, ZIPS, ZIP, PIZ = 0, 1, 2, 3, 4


def _attr(name, kind, body):
    return name.encode() + b"\0" + kind.encode() + b"\0" + struct.pack("<i", len(body)) + body


def _rle(data):
    out, i, n = bytearray(), 0, len(data)
    while i < n:
        run = 1
        while i + run < n and run < 128 and data[i + run] == data[i]:
            run += 1
        if run >= 3:
            out += struct.pack("b", run - 1) + bytes([data[i]])
            i += run
            continue
        j = i
        while j < n and j - i < 127 and not (j + 2 < n and data[j] == data[j + 1] == data[j + 2]):
            j += 1
        out += struct.pack("b", -(j - i)) + bytes(data[i:j])
        i = j
    return bytes(out)


def _exr(channels, compression, window=None, line_order=0, level=6, flags=0):
    """channels: {name: (pixel type, (H, W) array)}.  Returns the file's bytes."""
    names = sorted(channels)
    h, w = channels[names[0]][1].shape
    x0, y0 = window if window else (0, 0)
    chlist = b""
    for n in names:
        chlist += n.encode() + b"\0" + struct.pack("<iBBBBii", channels[n][0], 0, 0, 0, 0, 1, 1)
    chlist += b"\0"
    head = struct.pack("<II", 20000630, 2 | flags)
    head += _attr("channels", "chlist", chlist) + _attr("compression", "compression", bytes([compression]))
    head += _attr("dataWindow", "box2i", struct.pack("<iiii", x0, y0, x0 + w - 1, y0 + h - 1))
    head += _attr("displayWindow", "box2i", struct.pack("<iiii", 0, 0, x0 + w - 1, y0 + h - 1))
    head += _attr("lineOrder", "lineOrder", bytes([line_order])) + _attr("pixelAspectRatio", "float", struct.pack("<f", 1.0))
    head += _attr("screenWindowCenter", "v2f", struct.pack("<ff", 0, 0)) + _attr("screenWindowWidth", "float", struct.pack("<f", 1.0)) + b"\0"
    per = 16 if compression in (ZIP,) else (32 if compression == PIZ else 1)
    dtypes = {HALF: "<f2", FLOAT: "<f4", UINT: "<u4"}
    chunks = []
    for c0 in range(0, h, per):
        raw = b"".join(np.ascontiguousarray(channels[n][1][y].astype(dtypes[channels[n][0]])).tobytes() for y in range(c0, min(h, c0 + per)) for n in names)
        body = raw
        if compression in (RLE, ZIPS, ZIP):
            a = np.frombuffer(raw, np.uint8)
            split = np.concatenate([a[0::2], a[1::2]]).astype(np.int32)
            pred = split.copy()
            pred[1:] = (split[1:] - split[:-1] + 128) & 255
            packed = _rle(bytes(pred.astype(np.uint8))) if compression == RLE else zlib.compress(bytes(pred.astype(np.uint8)), level)
            if len(packed) < len(raw):
                body = packed
        chunks.append(struct.pack("<ii", y0 + c0, len(body)) + body)
    order = list(range(len(chunks)))
    if line_order == 1:
        order.reverse()                                     # DECREASING_Y: the chunks are stored bottom first; the table stays in y order
    at = len(head) + 8 * len(chunks)
    offsets = [0] * len(chunks)
    blob = b""
    for k in order:
        offsets[k] = at + len(blob)
        blob += chunks[k]
    return head + b"".join(struct.pack("<Q", o) for o in offsets) + blob


def _load(tmp_path, data, name="t.exr"):
    p = tmp_path / name
    p.write_bytes(data)
    s = api.HostScene()
    slot = s.load_texture(str(p))
    (_, w, h, fmt, texels), = [t for t in s.textures() if t[0] == slot]
    assert fmt == api.TEX_RGBA32F
    return texels.view(np.float32).reshape(h, w, 4)


def _same(a, b):
    return np.array_equal(np.ascontiguousarray(a, np.float32).view(np.uint32), np.ascontiguousarray(b, np.float32).view(np.uint32))


def test_uncompressed_float_rgb_with_an_offset_data_window(built_lib, tmp_path):
    rng = np.random.default_rng(1)
    r, g, b = (rng.standard_normal((5, 7)).astype(np.float32) * 100 for _ in range(3))
    img = _load(tmp_path, _exr({"R": (FLOAT, r), "G": (FLOAT, g), "B": (FLOAT, b)}, NONE, window=(3, 2)))
    assert img.shape == (5, 7, 4) and _same(img[..., 0], r) and _same(img[..., 1], g) and _same(img[..., 2], b) and np.all(img[..., 3] == 1.0)


@pytest.mark.parametrize("line_order", [0, 1])
@pytest.mark.parametrize("level", [1, 6, 9])
def test_zip_half_rgba(built_lib, tmp_path, line_order, level):
    """ZIP: 16 scanlines per chunk (37 rows: 16 + 16 + 5), HALF samples incl. denormals, infinities and a NaN, smooth data that deflate
    codes with long matches (dynamic Huffman blocks) next to noise."""
    rng = np.random.default_rng(2)
    h, w = 37, 19
    yy, xx = np.mgrid[0:h, 0:w]
    planes = {"R": (np.sin(xx * 0.3) * 8 + yy).astype(np.float16), "G": rng.standard_normal((h, w)).astype(np.float16),
              "B": np.full((h, w), 0.25, np.float16), "A": (xx % 3 == 0).astype(np.float16)}
    planes["G"][0, :6] = np.array([np.inf, -np.inf, 6e-8, -6e-8, 65504, 0], np.float16).view(np.float16)
    planes["G"][1, 0] = np.float16(np.nan)
    img = _load(tmp_path, _exr({k: (HALF, v) for k, v in planes.items()}, ZIP, line_order=line_order, level=level))
    for k, name in enumerate("RGBA"):
        want = planes[name].astype(np.float32)
        ok = (img[..., k] == want) | (np.isnan(img[..., k]) & np.isnan(want))
        assert ok.all(), name
        assert _same(np.where(np.isnan(want), 0, img[..., k]), np.where(np.isnan(want), 0, want)), name      # bit patterns incl. -0 / denormals


def test_zips_luminance_only_and_a_long_scanline(built_lib, tmp_path):
    """ZIPS (one scanline per chunk) with a single Y channel: R = G = B = Y.  A 4096-pixel line of structured FLOAT data gives deflate
    matches at distances up to the 16 KB of a line; a noise line next to it does not shrink and is stored as it is."""
    rng = np.random.default_rng(3)
    w = 4096
    y = np.stack([np.tile(np.arange(64, dtype=np.float32), w // 64), rng.standard_normal(w).astype(np.float32), np.zeros(w, np.float32),
                  (np.arange(w) // 512).astype(np.float32)])
    img = _load(tmp_path, _exr({"Y": (FLOAT, y)}, ZIPS, level=9))
    for k in range(3):
        assert _same(img[..., k], y)
    assert np.all(img[..., 3] == 1.0)


def test_rle_and_uint(built_lib, tmp_path):
    rng = np.random.default_rng(4)
    h, w = 9, 33
    r = np.repeat(rng.integers(0, 4, (h, 3)), 11, axis=1).astype(np.float16)            # flat runs
    g = rng.standard_normal((h, w)).astype(np.float16)
    ids = rng.integers(0, 1 << 20, (h, w)).astype(np.uint32)
    img = _load(tmp_path, _exr({"R": (HALF, r), "G": (HALF, g), "B": (UINT, ids)}, RLE))
    assert _same(img[..., 0], r.astype(np.float32)) and _same(img[..., 1], g.astype(np.float32)) and _same(img[..., 2], ids.astype(np.float32))


def test_tiny_file_and_other_channels_are_ignored(built_lib, tmp_path):
    """2 x 1 pixels (deflate codes so little with its fixed Huffman tables); a depth channel Z next to RGB is skipped."""
    r = np.array([[1.5, -2.0]], np.float16)
    z = np.array([[10.0, 20.0]], np.float32)
    for comp in (ZIPS, ZIP):
        img = _load(tmp_path, _exr({"R": (HALF, r), "G": (HALF, r * 2), "B": (HALF, r * 4), "Z": (FLOAT, z)}, comp, level=9), name="tiny%d.exr" % comp)
        assert _same(img[0, :, 0], r[0].astype(np.float32)) and _same(img[0, :, 2], (r[0] * 4).astype(np.float32))
        assert not np.array_equal(img[0, :, 2], z[0])


def test_all_three_deflate_block_types(built_lib, tmp_path):
    """The reader's own inflate meets stored, fixed-Huffman and dynamic-Huffman blocks: the first block type of each stream is read back
    from the file (bits 1-2 of the byte after the zlib header)."""
    rng = np.random.default_rng(6)
    seen = set()
    cases = [("flat", np.full((1, 64), 3.0, np.float32), 9), ("skewed", rng.integers(0, 7, (1, 4096)).astype(np.float32) * 0.37, 9),
             ("stored", np.tile(np.arange(256, dtype=np.float32), (1, 8)), 0)]
    for name, y, level in cases:
        data = _exr({"Y": (FLOAT, y)}, ZIPS, level=level)
        head_end = data.index(b"screenWindowWidth") + len("screenWindowWidth") + 1 + len("float") + 1 + 4 + 4 + 1
        chunk = head_end + 8                                  # one chunk: its offset table entry, then y, size, zlib header (2 bytes)
        size = struct.unpack_from("<i", data, chunk + 4)[0]
        if size < y.size * 4:                                 # compressed (a level-0 stream is larger than the data and is not used)
            seen.add((data[chunk + 8 + 2] >> 1) & 3)
        img = _load(tmp_path, data, name=name + ".exr")
        assert _same(img[..., 0], y), name
    assert {1, 2} <= seen, seen                               # fixed and dynamic Huffman blocks were both decoded
    # a stored block inside a stream that is still smaller than the data: a zlib stream assembled by hand (stored block of the predicted bytes
    # of a short constant line is longer than the line, so the reader would take it as raw; instead check stored blocks through a two-block stream)
    y = np.tile(np.arange(128, dtype=np.float32), (1, 16))
    raw = y.astype("<f4").tobytes()
    a = np.frombuffer(raw, np.uint8)
    split = np.concatenate([a[0::2], a[1::2]]).astype(np.int32)
    pred = split.copy()
    pred[1:] = (split[1:] - split[:-1] + 128) & 255
    pred = bytes(pred.astype(np.uint8))
    co = zlib.compressobj(9, zlib.DEFLATED, 15)
    stream = co.compress(pred[:100]) + co.flush(zlib.Z_FULL_FLUSH)          # Z_FULL_FLUSH ends the block and appends an EMPTY STORED block
    stream += co.compress(pred[100:]) + co.flush()
    assert len(stream) < len(raw)
    data = bytearray(_exr({"Y": (FLOAT, y)}, NONE))
    table = bytes(data).index(b"screenWindowWidth") + len("screenWindowWidth") + 1 + len("float") + 1 + 4 + 4 + 1
    data = bytes(data[:table + 8]) + struct.pack("<ii", 0, len(stream)) + stream
    data = data.replace(b"compression\0compression\0" + struct.pack("<i", 1) + bytes([NONE]), b"compression\0compression\0" + struct.pack("<i", 1) + bytes([ZIPS]))
    img = _load(tmp_path, data, name="flushed.exr")
    assert _same(img[..., 0], y)


def test_what_the_reader_refuses(built_lib, tmp_path):
    r = np.zeros((4, 4), np.float16)
    s = api.HostScene()
    for name, data, what in (("piz.exr", _exr({"R": (HALF, r)}, PIZ), "PIZ"),
                             ("tiled.exr", _exr({"R": (HALF, r)}, NONE, flags=0x200), "tiled"),
                             ("cut.exr", _exr({"R": (HALF, r)}, ZIP)[:-3], "chunk"),
                             ("nochan.exr", _exr({"Z": (FLOAT, r.astype(np.float32))}, NONE), "no R, G, B or Y")):
        p = tmp_path / name
        p.write_bytes(data)
        with pytest.raises(api.GfxError, match=what):
            s.load_texture(str(p))
    wild = bytearray(_exr({"R": (HALF, r)}, NONE))
    table = bytes(wild).index(b"screenWindowWidth") + len("screenWindowWidth") + 1 + len("float") + 1 + 4 + 4 + 1
    struct.pack_into("<Q", wild, table, 0xFFFFFFFFFFFFFFFC)      # a chunk offset that wraps when eight is added to it
    p = tmp_path / "wild.exr"
    p.write_bytes(bytes(wild))
    with pytest.raises(api.GfxError, match="offset"):
        s.load_texture(str(p))
    good = bytearray(_exr({"R": (HALF, np.tile(np.arange(16, dtype=np.float16), (16, 1)))}, ZIP))
    good[-20] ^= 0x5A                                         # a damaged deflate stream
    p = tmp_path / "damaged.exr"
    p.write_bytes(bytes(good))
    try:
        s.load_texture(str(p))                                # either refused or decoded to the right size: never a crash
    except api.GfxError:
        pass


def test_round_trip_through_the_products_exr_writer(built_lib, tmp_path):
    """gfxh_save_image_hdr(.exr) -> gfxh_scene_load_texture: the HALF values the writer stored (brightness scale applied), channel for
    channel -- the same file a renderer's frame is saved to with `-out frame.exr`."""
    rng = np.random.default_rng(5)
    w, h = 21, 13
    rgba = (rng.random((h, w, 4)) * 50).astype(np.float32)
    path = str(tmp_path / "frame.exr")
    api.save_image_hdr(path, rgba.reshape(-1, 4), w, h, 2.0)
    s = api.HostScene()
    slot = s.load_texture(path)
    (_, tw, th, fmt, texels), = [t for t in s.textures() if t[0] == slot]
    got = texels.view(np.float32).reshape(th, tw, 4)
    assert (tw, th) == (w, h) and _same(got, (np.float32(2.0) * rgba).astype(np.float16).astype(np.float32))
