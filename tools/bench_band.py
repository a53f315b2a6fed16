"""Frame time of ONE rank's row band on one GPU (no communication) when the 1920x1080 bench frame is split N ways
(tilesplit.band_rows), as a compute-only bound on strong scaling, in both band modes:
  strips          gfxh_restir_set_exchange with a callback that moves nothing: every pass on the band only (what bench.py --gpus N
                  runs), the next frame's G-buffer pass pipelined underneath the reuse passes (round 3)
  strips_serial   the same with GFX_SERIAL_FRAMES=1: every pass on one stream (round 2)
  strips_balanced the pipelined strip mode with cost-balanced bands: the equal partition's band times go through
                  gfxh_balance_bands (what bench.py --gpus N does during its warm-up), twice
  halo            no callback: the band plus the halo rows the reuse passes read are recomputed (round-1 scheme)
The seam rows' state is not refreshed here (no neighbour rank), which does not change the amount of work.  One JSON line."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gfxexp_amd import api, scenes, tilesplit  # noqa: E402


def band_ms(ctx, cam, W, H, band, steps=30, strips=False, serial=False):
    import torch
    config4 = "--config4" in sys.argv       # BASELINE configs[4]: unbiased estimator + 2048 x 1024 environment map (the 8-GPU configuration)
    cfg = api.RestirRenderer.default_config(W, H, api.RENDERER_UNBIASED if config4 else api.RENDERER_BIASED)
    cfg.camera = cam
    cfg.rowBegin, cfg.rowEnd = band
    cfg.enableBumpMapping = int("--plain" not in sys.argv)
    if serial:
        os.environ["GFX_SERIAL_FRAMES"] = "1"
    r = api.RestirRenderer(ctx, cfg)
    os.environ.pop("GFX_SERIAL_FRAMES", None)
    if config4:
        r.set_env(api.env_make_sky(2048, 1024), 2048, 1024, 0.6, 0.4)
    if strips:
        r.set_exchange(lambda stream, d: None, 0)
    for _ in range(5):
        r.render_frame()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        r.render_frame()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps * 1e3
    r.close()
    return dt


def main():
    W, H = 1920, 1080
    ctx = api.Context(0)
    textured = "--plain" not in sys.argv
    scenes.bench_street(textured=textured).upload(ctx)
    cam = api.make_camera(W, H, pos=(1.5, 2.2, 52.0), pitch=4.0, yaw=181.5)
    full = band_ms(ctx, cam, W, H, (0, 0))
    out = {"workload": "bench frame (%s street%s), one rank's band rendered alone on one GPU (compute only)" % ("textured" if textured else "plain", ", configs[4]: unbiased + environment map" if "--config4" in sys.argv else ""), "full_frame_ms": round(full, 4), "bands": {}}
    for n in (2, 4, 8):
        bands = tilesplit.band_rows(H, n)
        entry = {}
        modes = sys.argv[sys.argv.index("--modes") + 1].split(",") if "--modes" in sys.argv else ["strips", "strips_serial", "halo"]
        for mode in modes:
            ms = [band_ms(ctx, cam, W, H, b, strips=mode != "halo", serial=mode == "strips_serial") for b in bands]
            worst = max(ms)
            entry[mode] = {"band_ms": [round(m, 4) for m in ms], "compute_bound_speedup": round(full / worst, 2),
                           "compute_bound_efficiency": round(full / worst / n, 3)}
        # cost-balanced bands: two rounds of measuring and re-cutting, strips no shorter than the exchange strip (radius 20 -> 24 rows)
        cur, ms = bands, [band_ms(ctx, cam, W, H, b, strips=True) for b in bands]
        for _ in range(2):
            cur = api.balance_bands(H, cur, ms, min_rows=24)
            ms = [band_ms(ctx, cam, W, H, b, strips=True) for b in cur]
        worst = max(ms)
        entry["strips_balanced"] = {"bands": cur, "band_ms": [round(m, 4) for m in ms], "compute_bound_speedup": round(full / worst, 2),
                                    "compute_bound_efficiency": round(full / worst / n, 3)}
        out["bands"][str(n)] = entry
    print(json.dumps(out))


if __name__ == "__main__":
    main()
