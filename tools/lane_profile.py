#!/usr/bin/env python
"""Lane-utilisation profile of k_initial_candidates on the bench frame: an experiment build (GFX_LANE_PROFILE, gm_math.hip.h
GFX_PROF) counts, per code section, how often a wave enters it and with how many active lanes.

    python gfxexp_amd/build.py --variant laneprof GFX_LANE_PROFILE
    GFX_LIB=$PWD/gfxexp_amd/variants/libgfxexp_laneprof.so python tools/lane_profile.py [--plain]
"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

SECTIONS = {0: "candidate loop body (RNG, light selection, fetch, shadow-ray geometry, reservoir update)",
            1: "light_fetch (record + normal matrix gathers, point on the triangle)",
            2: "emitter faces the shading point (lpCos > 0): BSDF evaluate is called",
            3: "BSDF evaluate past the horizon test (GGX D / G / Fresnel, diffuse lobe)",
            4: "deferred emittance-texture read",
            5: "reservoir accepts the candidate (sample copy)",
            7: "smooth-emitter / fallback extra record",
            8: "after the loop (finalise, shadow ray)"}


def main():
    import torch
    from gfxexp_amd import api, scenes
    from tests import util
    W, H = 1920, 1080
    textured = "--plain" not in sys.argv
    hs = scenes.bench_street(textured=textured)
    ctx = api.Context(0)
    hs.upload(ctx)
    accel = ctx.accel_build()
    ctx.lights_build_static()
    pb = util.PixelBuffers(W, H)
    env = "--env" in sys.argv                  # configs[4]: 2048 x 1024 sky + sun environment map, a quarter of the candidates sample it
    if env:
        pb.set_env(api.env_make_sky(2048, 1024), 2048, 1024)
        pb.use_env_row_table = "--no-row-table" not in sys.argv
    dev = util.DeviceBuffers(pb)
    cam = api.make_camera(W, H, pos=(1.5, 2.2, 52.0), pitch=4.0, yaw=181.5)
    stream = torch.cuda.current_stream().cuda_stream
    f = util.frame_params(api.GfxRestirFrameParams, api.GfxCamera, W, H, cam, travHandle=accel, frameIndex=0, bufferIndex=0, resetFlowBuffer=1,
                          enableBumpMapping=int(textured), enableEnvLight=int(env), envLightPowerCoeff=0.6, envLightRotation=0.4)
    ctx.lights_build_instances(stream)
    ctx.restir_set_params(dev.static_params(), f, 0, 0, stream)
    ctx.restir_launch(api.PASS_SETUP_GBUFFERS, W, H, stream)
    torch.cuda.synchronize()
    L = api.lib()
    out = (C.c_uint64 * 64)()
    assert L.gfx_debug_lane_profile(out, 1) == 0
    ctx.restir_launch(api.PASS_INITIAL_RIS, W, H, stream)
    torch.cuda.synchronize()
    assert L.gfx_debug_lane_profile(out, 1) == 0
    res = {}
    for k, name in SECTIONS.items():
        lanes, visits = int(out[2 * k]), int(out[2 * k + 1])
        res[k] = {"section": name, "wave_visits": visits, "lanes": lanes, "lanes_per_visit": round(lanes / max(1, visits), 2)}
    # where the waves' clock cycles went (GFX_CYC marks in k_initial_candidates; wave-level s_memtime)
    names = {0: "random numbers, light type, table lookup", 1: "cooperative record fetch (issue, wait, read back), matrix loads issued",
             2: "point on the emitter (waits for the matrix)", 3: "shadow-ray geometry, BSDF evaluation, emittance texture", 4: "reservoir update",
             5: "after the loop", 7: "before the loop"}
    cyc = {k: int(out[32 + k]) for k in names}
    total = max(1, sum(cyc.values()))
    print(json.dumps({"workload": "textured" if textured else "plain", "sections": res,
                      "wave_cycle_shares": {names[k]: round(cyc[k] / total, 4) for k in names}, "wave_cycles_total": total}, indent=1))


if __name__ == "__main__":
    main()
