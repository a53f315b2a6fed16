"""Why a band's trace launches are slower per ray than the full frame's: scheduling diagnostics of the counting k_trace
(wave iterations, lane occupancy, drain share, mean wave lifetime) next to the serial per-kernel times, for the full 1080p
bench frame and one band of an 8-way and a 4-way split.  JSON lines."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gfxexp_amd import api, scenes, tilesplit  # noqa: E402


def measure(ctx, cam, W, H, band, bpc=4):
    import torch
    cfg = api.RestirRenderer.default_config(W, H, api.RENDERER_BIASED)
    cfg.camera = cam
    cfg.rowBegin, cfg.rowEnd = band
    cfg.enableBumpMapping = 1
    os.environ["GFX_SERIAL_FRAMES"] = "1"
    r = api.RestirRenderer(ctx, cfg)
    del os.environ["GFX_SERIAL_FRAMES"]
    if band != (0, 0):
        r.set_exchange(lambda stream, d: None, 0)
    for _ in range(5):
        r.render_frame()
    torch.cuda.synchronize()
    ctx.timing_enable(True)
    for _ in range(8):
        r.render_frame()
    torch.cuda.synchronize()
    t = ctx.timing_collect()
    ctx.timing_enable(False)
    per = {k: round(ms / 8, 4) for k, (ms, calls) in t.items() if k.startswith("trace")}
    ctx.counters_enable(True)
    ctx.counters_read(reset=True); ctx.trace_diag_read(reset=True)
    ctx.timing_enable(True)
    r.render_frame()
    torch.cuda.synchronize()
    tc = {k: round(ms, 4) for k, (ms, calls) in ctx.timing_collect().items() if k.startswith("trace")}
    ctx.timing_enable(False)
    c = ctx.counters_read(reset=True)
    d = ctx.trace_diag_read(reset=True)
    ctx.counters_enable(False)
    r.close()
    waves = 256 * bpc * 4      # CUs x blocks per CU x waves per block, per launch
    launches = 3
    clock_mhz = 2400.0
    return {"band": list(band), "trace_ms_serial": per, "trace_ms_counting_launches": tc, "rays": c["rays"], "items_per_ray": round((c["nodeFetches"] + c["triFetches"]) / max(1, c["rays"]), 2),
            "wave_iterations": d["iterations"], "iterations_per_wave_launch": round(d["iterations"] / (waves * launches), 1),
            "lane_occupancy": round(d["itemLanes"] / max(1, 64 * d["iterations"]), 3),
            "drain_share": round(d["drainIterations"] / max(1, d["iterations"]), 3),
            "drain_lane_occupancy": round(d["drainItemLanes"] / max(1, 64 * d["drainIterations"]), 3),
            "cycles_per_iteration": round(d["waveCycles"] / max(1, d["iterations"]), 1),
            "mean_wave_lifetime_us": round(d["waveCycles"] / (waves * launches) / clock_mhz, 1),
            "sum_counting_launch_us": round(1e3 * sum(tc.values()), 1),
            "shares": {"refill": round(d["refillCycles"] / max(1, d["waveCycles"]), 3), "fetch": round(d["fetchCycles"] / max(1, d["waveCycles"]), 3),
                       "process": round(d["processCycles"] / max(1, d["waveCycles"]), 3)}}


def main():
    W, H = 1920, 1080
    ctx = api.Context(0)
    scenes.bench_street(textured=True).upload(ctx)
    cam = api.make_camera(W, H, pos=(1.5, 2.2, 52.0), pitch=4.0, yaw=181.5)
    print(json.dumps(measure(ctx, cam, W, H, (0, 0))), flush=True)
    print(json.dumps(measure(ctx, cam, W, H, tilesplit.band_rows(H, 8)[4])), flush=True)
    print(json.dumps(measure(ctx, cam, W, H, tilesplit.band_rows(H, 4)[1])), flush=True)
    for bpc in (1, 2):
        ctx.tunable_set("trace_blocks_per_cu", bpc)
        out = measure(ctx, cam, W, H, tilesplit.band_rows(H, 8)[4], bpc)
        out["trace_blocks_per_cu"] = bpc
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
