"""The lanes of a band renderer change WHEN an exchange runs, never what a frame computes (run as a script by
tests/test_gpu_strip_exchange.py: the product loads one librccl per process and this check names tests/native/librccl_mirror.so).

Band 2 of 4 and band 4 of 8 of a 640x360 frame render the same sequence three times through the production C++ callback
(gfxh_rccl_exchange) over the mirror transport, whose strips are a deterministic function of the sender's rows at the moment the
transfer executes on its stream:
    serial    GFX_SERIAL_FRAMES=1: every pass and every exchange on ONE stream, in program order -- the reference schedule (stripMode 1; a second
              one in stripMode 3 for the `recompute` schedule)
    round5    the G-buffer pass pipelined, the G-buffer strips on the frame's stream ahead of the candidate pass, gather synchronous
    noseam    the G-buffer strips on the G-buffer lane behind the pipelined pass, the band gather on the gather lane underneath the
              next frame, with 40 / 150 microseconds of injected latency per strip exchange / gather (so that a missing wait reads rows
              that have not arrived)
    lanes     + the seam rows of a spatial pass that another pass follows first, their exchange on the seam lane underneath the interior
              rows (stripMode 2, GFX_SEAM_FIRST=1)
    recompute the lanes with stripMode 3: the spatial pass that another pass follows recomputed on its halo, ONE reservoir exchange per frame
              (reservoirs + pixel RNG states, radius x passes rows) -- the schedule gfxh_restir_render_frame runs by default
Every buffer a later pass or frame reads (G-buffer halves, reservoirs, infos, the RNG states, the HDR frame) must come out bit for bit
the same: a missing or misplaced event between the lanes shows up as a difference."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
MIRROR = os.path.join(ROOT, "tests", "native", "librccl_mirror.so")


def render(api, ctx, cfg, rank, world, H, frames, schedule, mirror, ids, moving):
    import torch
    os.environ.pop("GFX_SERIAL_FRAMES", None)
    os.environ["GFX_GB_STRIPS_ON_MAIN"] = "0"
    os.environ["GFX_SEAM_FIRST"] = "1" if schedule == "lanes" else "0"
    os.environ["GFX_STRIP_MODE"] = "3" if schedule in ("recompute", "serial3") else "1"        # (GFX_SEAM_FIRST=1 makes it 2)
    if schedule in ("serial", "serial3"):
        os.environ["GFX_SERIAL_FRAMES"] = "1"
    if schedule == "round5":
        os.environ["GFX_GB_STRIPS_ON_MAIN"] = "1"
    injected = schedule in ("lanes", "noseam", "recompute")
    mirror.rccl_mirror_set_latency_us(C.c_float(40.0 if injected else 0.0), C.c_float(150.0 if injected else 0.0))
    r = api.RestirRenderer(ctx, cfg)
    ex = api.RcclExchange(ids, rank, world, H)
    ex.install(r, 8 if moving else 0)
    r.set_async_gather(injected)
    stream = torch.cuda.current_stream().cuda_stream
    for f in range(frames):
        if moving:
            r.set_camera(api.make_camera(cfg.width, cfg.height, pos=(1.5 + 0.02 * f, 5.0, 14.0), pitch=12.0, yaw=186.0 + 0.1 * f))
        r.render_frame(stream)
    r.finish_gather(stream)
    torch.cuda.synchronize()
    s, _, last_res, _, _ = r.params()
    n = cfg.width * cfg.height
    out = {"beauty": ctx.read_device(s.beautyAccumBuffer, 16 * n), "rng": ctx.read_device(s.rngBuffer, 8 * n)}
    for i in range(2):
        out["res%d" % i] = ctx.read_device(s.reservoirBuffer[i], 48 * n)
        out["info%d" % i] = ctx.read_device(s.reservoirInfoBuffer[i], 8 * n)
        for k, (ptr, b) in enumerate(((s.gbuffer0[i], 16), (s.gbuffer2[i], 16), (s.gbuffer3[i], 16))):
            out["gb%d_%d" % (k, i)] = ctx.read_device(ptr, b * n)
    r.close()
    ex.close()
    return out


def main():
    os.environ["GFX_RCCL_LIBRARY"] = MIRROR
    import torch  # noqa: F401
    from gfxexp_amd import api
    from tests import util
    mirror = C.CDLL(MIRROR)
    mirror.rccl_mirror_set_latency_us.argtypes = [C.c_float, C.c_float]
    W, H, frames = 640, 360, 5
    ctx = api.Context(0)
    util.bunny_scene().upload(ctx)
    checked = 0
    for renderer, world, moving in ((api.RENDERER_BIASED, 4, False), (api.RENDERER_UNBIASED, 8, False), (api.RENDERER_BIASED, 4, True),
                                    (api.RENDERER_REARCH_BIASED, 4, False), (api.RENDERER_PATH_TRACE, 4, False)):
        rank = world // 2
        cfg = api.RestirRenderer.default_config(W, H, renderer)
        cfg.camera = api.make_camera(W, H, pos=(1.5, 5.0, 14.0), pitch=12.0, yaw=186.0)
        cfg.spatialNeighborRadius = 12.0
        cfg.rowBegin, cfg.rowEnd = api.band_rows(H, world, rank)
        ids = api.RcclExchange.unique_ids(api.NUM_LANES)
        # (stripMode 3 moves other rows than modes 1 / 2 -- the candidate pass's reservoirs and RNG states instead of the first spatial pass's
        # results -- and the mirror transport is not a consistent neighbour: its frames are compared with ITS one-stream schedule)
        references = {"serial": render(api, ctx, cfg, rank, world, H, frames, "serial", mirror, ids, moving),
                      "serial3": render(api, ctx, cfg, rank, world, H, frames, "serial3", mirror, ids, moving)}
        for schedule in ("round5", "noseam", "lanes", "recompute"):
            want = references["serial3" if schedule == "recompute" else "serial"]
            got = render(api, ctx, cfg, rank, world, H, frames, schedule, mirror, ids, moving)
            for name in want:
                if not np.array_equal(want[name], got[name]):
                    bad = np.flatnonzero(want[name] != got[name])
                    raise SystemExit("renderer %d, %d bands, moving=%s, schedule %s: %s differs from the one-stream schedule at %d bytes (first at byte %d)"
                                     % (renderer, world, moving, schedule, name, bad.size, bad[0]))
                checked += 1
    print("ok: %d buffers bit-identical across the serial, round-5 and lane schedules" % checked)


if __name__ == "__main__":
    main()
