"""Experiment: would running the two halves of a frame as concurrent streams fill the tails of the VALU-bound kernels?
Two band renderers (rows 0-540, 540-1080, strip mode with a callback that moves nothing) are driven from two host threads on
two HIP streams; the wall time per frame pair is compared with the whole-frame renderer.  (Exchange copies are left out: an
upper bound on what intra-frame pipelining over two streams could give.)"""
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gfxexp_amd import api, scenes  # noqa: E402


def main():
    import torch
    W, H, K = 1920, 1080, 40
    hs = scenes.bench_street(textured=True)
    cam = api.make_camera(W, H, pos=(1.5, 2.2, 52.0), pitch=4.0, yaw=181.5)

    def make(band):
        ctx = api.Context(0)
        hs.upload(ctx)
        cfg = api.RestirRenderer.default_config(W, H, api.RENDERER_BIASED)
        cfg.camera = cam
        cfg.enableBumpMapping = 1
        cfg.rowBegin, cfg.rowEnd = band
        r = api.RestirRenderer(ctx, cfg)
        if band != (0, 0):
            r.set_exchange(lambda stream, d: None, 0)
        return ctx, r

    _, full = make((0, 0))
    for _ in range(5):
        full.render_frame()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        full.render_frame()
    torch.cuda.synchronize()
    t_full = (time.perf_counter() - t0) / K * 1e3
    out = {"full_frame_ms": round(t_full, 4)}
    for split in (2, 3, 4):
        rows = [(H * k // split) // 8 * 8 for k in range(split)] + [H]
        made = [make((rows[k], rows[k + 1])) for k in range(split)]
        streams = [torch.cuda.Stream() for _ in made]

        def run(k, n):
            for _ in range(n):
                made[k][1].render_frame(streams[k].cuda_stream)
        for n in (5, K):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            th = [threading.Thread(target=run, args=(k, n)) for k in range(split)]
            for t in th:
                t.start()
            for t in th:
                t.join()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n * 1e3
        out[f"{split}_concurrent_bands_ms"] = round(dt, 4)
        del made
    print(json.dumps(out))


if __name__ == "__main__":
    main()
