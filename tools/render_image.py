"""Render the bench frame (or the bunny) with accumulation and write it out through the output chain
(gfxh_save_image_sdr / _hdr); also a PNG (stdlib zlib) for quick inspection.

    python tools/render_image.py [--scene street|bunny] [--renderer 0..5] [--frames N] [--width W --height H] [--out gpurun_out/frame]
"""
import argparse
import os
import struct
import sys
import zlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gfxexp_amd import api, scenes  # noqa: E402

BUNNY_OBJ = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "assets", "stanford_bunny_309_faces.obj")


def write_png(path, px, width, height):
    """px: uint32 R | G << 8 | B << 16 per pixel, top row first."""
    rgb = np.stack([px & 255, (px >> 8) & 255, (px >> 16) & 255], axis=-1).astype(np.uint8).reshape(height, width * 3)
    raw = b"".join(b"\x00" + rgb[y].tobytes() for y in range(height))

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", width, height, 8, 2, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="street")
    ap.add_argument("--renderer", type=int, default=api.RENDERER_BIASED, help="0-5 = gfxh_renderer, 6 = NRC path tracer")
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--width", type=int, default=960)
    ap.add_argument("--height", type=int, default=540)
    ap.add_argument("--brightness", type=float, default=1.0)
    ap.add_argument("--out", default="gpurun_out/frame")
    args = ap.parse_args()
    import torch
    W, H = args.width, args.height
    ctx = api.Context(0)
    if args.scene == "street":
        scenes.bench_street().upload(ctx)
        cam = api.make_camera(W, H, pos=(1.5, 2.2, 52.0), pitch=4.0, yaw=181.5)
    else:
        scenes.bunny_scene(BUNNY_OBJ).upload(ctx)
        cam = api.make_camera(W, H, pos=(1.5, 5.0, 14.0), pitch=12.0, yaw=186.0)
    if args.renderer == 6:     # neural radiance cache: --warm frames of training without accumulation, then accumulate
        hs_bounds = (scenes.bench_street() if args.scene == "street" else scenes.bunny_scene(BUNNY_OBJ)).bounds()
        cfg = api.NrcRenderer.default_config(W, H, hs_bounds)
        cfg.camera = cam
        cfg.enableAccumulation = 1
        r = api.NrcRenderer(ctx, cfg)
        losses = [r.render_frame(want_loss=(k % 32 == 31)) for k in range(args.frames)]
        print({"loss_every_32_frames": [round(x, 5) for x in losses if x is not None]})
    else:
        cfg = api.RestirRenderer.default_config(W, H, args.renderer)
        cfg.camera = cam
        cfg.enableAccumulation = 1
        r = api.RestirRenderer(ctx, cfg)
        for _ in range(args.frames):
            r.render_frame()
    torch.cuda.synchronize()
    img = ctx.read_device(r.beauty_ptr(), W * H * 16).view(np.float32).reshape(H, W, 4)
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    sdr = api.sdr_config(brightness=args.brightness, tone_map=True, gamma=True)
    api.save_image_sdr(args.out + ".bmp", img, W, H, sdr)
    api.save_image_hdr(args.out + ".pfm", img, W, H)
    write_png(args.out + ".png", api.tonemap_sdr(img, W, H, sdr), W, H)
    rgb = img[..., :3]
    print({"out": args.out, "frames": args.frames, "mean_rgb": [float(x) for x in rgb.reshape(-1, 3).mean(0)],
           "finite": bool(np.isfinite(rgb).all()), "max": float(rgb.max())})


if __name__ == "__main__":
    main()
