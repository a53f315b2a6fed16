"""Experiment (verdict r03, lever a): would cutting one rank's band into sub-bands on separate streams fill the drain tails of its
small launches?  Band 4 of an 8-way split of the 1920x1080 bench frame (rows 544-680, the slowest) is rendered (i) by one band
renderer and (ii) as S sub-bands, each its own context (own scratch, ray queues, ticket areas) and band renderer on its own HIP
stream, driven from S host threads, strip mode with a callback that moves nothing and NO dependency between the sub-bands -- an
upper bound on what any intra-rank pipelining of sub-bands could give (the real schedule has to order a sub-band's spatial pass
behind its neighbour's temporal pass).  Also: the same band with S independent copies of itself side by side (S frames' worth of
work per wall-clock frame), to see how much of the GPU one band's frame leaves idle.  One JSON line."""
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gfxexp_amd import api, scenes, tilesplit  # noqa: E402


def main():
    import torch
    W, H, K = 1920, 1080, 60
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
        r.set_exchange(lambda stream, d: None, 0)
        return ctx, r

    def timed(made):
        streams = [torch.cuda.Stream() for _ in made]

        def run(k, n):
            for _ in range(n):
                made[k][1].render_frame(streams[k].cuda_stream)
        dt = 0.0
        for n in (8, K):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            th = [threading.Thread(target=run, args=(k, n)) for k in range(len(made))]
            for t in th:
                t.start()
            for t in th:
                t.join()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n * 1e3
        return round(dt, 4)

    band = tilesplit.band_rows(H, 8)[4]
    out = {"band": list(band)}
    one = [make(band)]
    out["one_renderer_ms"] = timed(one)
    for copies in (2, 3):
        made = one + [make(band) for _ in range(copies - 1)]
        out[f"{copies}_independent_copies_ms"] = timed(made)
        del made
    del one
    rows = band[1] - band[0]
    for split in (2, 3):
        cuts = [band[0] + (rows * k // split) // 8 * 8 for k in range(split)] + [band[1]]
        made = [make((cuts[k], cuts[k + 1])) for k in range(split)]
        out[f"{split}_sub_bands"] = {"rows": [[cuts[k], cuts[k + 1]] for k in range(split)], "concurrent_ms": timed(made),
                                      "one_after_the_other_ms": round(sum(timed([m]) for m in made), 4)}
        del made
    print(json.dumps(out))


if __name__ == "__main__":
    main()
