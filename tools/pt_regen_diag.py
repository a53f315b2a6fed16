"""configs[1] (path tracing on the bunny scene, 512x512, max path length 5) in the one-kernel form with and without path regeneration:
frame time and the scheduling counters of gfx_pt_diag_read -- bounce iterations summed over the waves, lanes that held a ray in them,
traversal steps, waves, refills.  JSON lines (profiles/r06_pt_regen.jsonl).
usage: pt_regen_diag.py [--steps 200]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from gfxexp_amd import api, scenes
    steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 200
    W = H = 512
    obj = os.path.join(ROOT, "tests", "golden", "assets", "stanford_bunny_309_faces.obj")
    for regen, min_refill in ((0, 16), (3, 1), (3, 16), (3, 32), (2, 16), (1, 16)):
        ctx = api.Context(0)
        scenes.bunny_scene(obj).upload(ctx)
        ctx.tunable_set("pt_regen", regen)
        ctx.tunable_set("pt_regen_min", min_refill)
        cfg = api.RestirRenderer.default_config(W, H, api.RENDERER_PATH_TRACE)
        cfg.camera = api.make_camera(W, H, pos=(1.5, 5.0, 14.0), pitch=12.0, yaw=186.0)
        r = api.RestirRenderer(ctx, cfg)
        stream = torch.cuda.current_stream().cuda_stream
        for _ in range(20):
            r.render_frame(stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            r.render_frame(stream)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        ctx.tunable_set("pt_diag", 1)
        r.render_frame(stream)
        torch.cuda.synchronize()
        ctx.pt_diag_read(reset=True)
        frames = 8
        for _ in range(frames):
            r.render_frame(stream)
        d = ctx.pt_diag_read(reset=True)
        ctx.tunable_set("pt_diag", 0)
        print(json.dumps({"kernel": "k_pt_regen" if regen else "k_pt_fused", "pt_regen_blocks_per_cu": regen, "pt_regen_min": min_refill if regen else None,
                          "ms_per_frame": round(ms, 4), "waves_per_frame": d["waves"] // frames, "wave_iterations_per_frame": d["iterations"] // frames,
                          "lane_fraction": round(d["lanes"] / max(1, 64 * d["iterations"]), 4),
                          "traversal_steps_per_frame": d["steps"] // frames, "steps_per_iteration": round(d["steps"] / max(1, d["iterations"]), 2),
                          "refills_per_frame": d["refills"] // frames}), flush=True)
        r.close()
        ctx.close()


if __name__ == "__main__":
    main()
