"""Output chain.  CPU: the EXR writer of gfxh_save_image_hdr (saveImageHDR, common_host.cpp:2762-2857:
channels A B G R stored as HALF) parsed back by a reader written here -- header attributes, channel order, fp32 -> fp16
round-to-nearest-even, brightness scale, flipY.  GPU: copyToLinearBuffers / visualizeToOutputBuffer
(restir_di/gpu_kernels/copy_buffers.cu:6-80) against numpy after a rendered frame, bit for bit."""
import struct

import numpy as np
import pytest

from gfxexp_amd import api
from tests import util


def _read_exr(path):
    d = open(path, "rb").read()
    assert struct.unpack_from("<II", d, 0) == (20000630, 2)
    at = 8
    attrs = {}
    while d[at] != 0:
        e = d.index(b"\0", at); name = d[at:e].decode(); at = e + 1
        e = d.index(b"\0", at); typ = d[at:e].decode(); at = e + 1
        size, = struct.unpack_from("<i", d, at); at += 4
        attrs[name] = (typ, d[at:at + size]); at += size
    at += 1
    chans = []
    c = attrs["channels"][1]
    k = 0
    while c[k] != 0:
        e = c.index(b"\0", k); chans.append((c[k:e].decode(), struct.unpack_from("<i", c, e + 1)[0])); k = e + 1 + 16
    x0, y0, x1, y1 = struct.unpack("<iiii", attrs["dataWindow"][1])
    w, h = x1 - x0 + 1, y1 - y0 + 1
    assert attrs["compression"][1] == b"\0" and attrs["lineOrder"][1] == b"\0"
    offsets = struct.unpack_from("<%dQ" % h, d, at)
    img = {}
    for y in range(h):
        yy, size = struct.unpack_from("<ii", d, offsets[y])
        assert yy == y and size == 2 * len(chans) * w
        row = np.frombuffer(d, np.float16, len(chans) * w, offsets[y] + 8).reshape(len(chans), w)
        for ci, (name, typ) in enumerate(chans):
            assert typ == 1      # HALF
            img.setdefault(name, np.zeros((h, w), np.float16))[y] = row[ci]
    return chans, img


def test_exr_writer_round_trip(tmp_path, built_lib):
    rng = np.random.default_rng(2)
    w, h = 13, 7
    rgba = (rng.standard_normal((h, w, 4)) * np.array([1e-7, 1.0, 300.0, 7e4])).astype(np.float32)
    rgba[0, 0] = (0.0, -0.0, 65504.0, 65519.9)          # largest half, and the last value that still rounds to it
    rgba[0, 1] = (65520.0, 1e30, -1e30, 6e-8)           # round to inf; just above the smallest subnormal
    rgba[0, 2] = (2.0 ** -24, 2.0 ** -25, 2.0 ** -25 * 1.0001, 1.0009765625 + 2.0 ** -11)   # ties and near-ties
    path = str(tmp_path / "o.exr")
    api.save_image_hdr(path, rgba, w, h, brightness=0.5, flip_y=True)
    chans, img = _read_exr(path)
    assert [c[0] for c in chans] == ["A", "B", "G", "R"]
    want = (np.float32(0.5) * rgba[::-1]).astype(np.float16)          # numpy's conversion rounds to nearest even
    with np.errstate(over="ignore"):
        for ci, name in enumerate(("R", "G", "B", "A")):
            assert np.array_equal(img[name].view(np.uint16), want[..., ci].view(np.uint16)), name


@pytest.mark.gpu
def test_copy_to_linear_and_visualize(built_lib):
    import torch
    w, h = 96, 64
    ctx = api.Context(0)
    hs = util.small_street()
    hs.upload(ctx)
    cfg = api.RestirRenderer.default_config(w, h, api.RENDERER_BIASED)
    cfg.camera = api.make_camera(w, h, pos=(2.0, 5.0, 26.0), pitch=4.0, yaw=180.0)
    cfg.enableAccumulation = 1
    r = api.RestirRenderer(ctx, cfg)
    stream = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        r.render_frame(stream)
    torch.cuda.synchronize()
    s, f, cur, base, _ = r.params()
    ctx.restir_set_params(s, f, cur, base)
    n = w * h
    col, alb, nrm = (torch.zeros((n, 4), dtype=torch.float32, device="cuda") for _ in range(3))
    mot = torch.zeros((n, 2), dtype=torch.float32, device="cuda")
    ctx.restir_copy_to_linear(col.data_ptr(), alb.data_ptr(), nrm.data_ptr(), mot.data_ptr(), stream)
    torch.cuda.synchronize()
    rd = lambda ptr, k: ctx.read_device(ptr, n * 4 * k).view(np.float32).reshape(n, k)
    beauty, albedo, normal = rd(s.beautyAccumBuffer, 4), rd(s.albedoAccumBuffer, 4), rd(s.normalAccumBuffer, 4)
    motion = rd(s.gbuffer1[f.bufferIndex], 2)
    util.assert_same_bits("color", col.cpu().numpy(), beauty)
    util.assert_same_bits("albedo", alb.cpu().numpy(), albedo)
    util.assert_same_bits("motion", mot.cpu().numpy(), motion)
    got_n = nrm.cpu().numpy()
    nz = np.any(normal[:, :3] != 0, axis=1)
    assert nz.mean() > 0.5
    assert np.allclose(np.linalg.norm(got_n[nz, :3], axis=1), 1.0, atol=2e-6) and np.all(got_n[~nz, :3] == 0) and np.all(got_n[:, 3] == 1)
    assert np.allclose(got_n[nz, :3] * np.linalg.norm(normal[nz, :3], axis=1, keepdims=True), normal[nz, :3], atol=2e-6)
    out = torch.zeros((n, 4), dtype=torch.float32, device="cuda")
    ctx.visualize(nrm.data_ptr(), 2, w, h, out.data_ptr(), stream=stream)
    torch.cuda.synchronize()
    want = got_n.copy(); want[:, :3] = np.float32(0.5) + np.float32(0.5) * got_n[:, :3]
    util.assert_same_bits("normal display", out.cpu().numpy(), want)
    ctx.visualize(mot.data_ptr(), 3, w, h, out.data_ptr(), mv_offset=0.5, mv_scale=0.02, stream=stream)
    torch.cuda.synchronize()
    m = mot.cpu().numpy()
    wantf = np.stack([np.clip(np.float32(0.02) * m[:, 0] + np.float32(0.5), 0, 1), np.clip(np.float32(0.02) * m[:, 1] + np.float32(0.5), 0, 1),
                      np.full(n, 0.5, np.float32), np.ones(n, np.float32)], 1).astype(np.float32)
    util.assert_same_bits("flow display", out.cpu().numpy(), wantf)
    ctx.visualize(col.data_ptr(), 0, w, h, out.data_ptr(), stream=stream)
    torch.cuda.synchronize()
    util.assert_same_bits("beauty display", out.cpu().numpy(), beauty)
