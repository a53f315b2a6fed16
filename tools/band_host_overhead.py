"""What the exchanges of a band renderer cost on ONE GPU, as far as one GPU can tell (profiles/r06_band_host_overhead.json).

  host      wall time the host spends inside gfxh_restir_render_frame per band frame (band 4 of 8 of the 1920x1080 bench frame), with
            the production C++ callback gfxh_rccl_exchange over the mirror librccl stand-in at zero latency (tests/native/librccl_mirror.so:
            a hipMemcpyAsync per received strip, about what a grouped ncclSend / ncclRecv costs the host) against the torch.distributed callback (tilesplit.StripExchange) over a `dist` whose collectives
            return at once -- the Python a frame executes between its passes, without any transport behind it.  The GPU is slower than
            either host, so the launch queue never pushes back: this is enqueue time.
  latency   frame time of that band with a transport of the right shape: gfxh_rccl_exchange over tests/native/librccl_mirror.so
            (the strips a rank would send across a seam come back as the strips it receives, behind a spin kernel of L microseconds
            per exchange point on the exchange's stream; the band gather behind a spin kernel of G microseconds), for L in
            0 / 30 / 60 / 120 and four schedules: `round5` = every exchange on the frame's stream in program order (G-buffer strips
            ahead of the candidate pass, the gather synchronous), `gb_lane` = the G-buffer strips on the G-buffer stream behind the
            pipelined pass, `lanes_noseam` = that + the gather on its own stream underneath the next frame, `lanes` = that + the seam
            rows of the first biased spatial pass ahead of its interior, their exchange on the seam lane, `recompute` = lanes_noseam with
            stripMode 3 (the first biased spatial pass recomputed on its halo: one reservoir exchange per frame; what bench.py --gpus N runs).
usage: band_host_overhead.py host|latency [--config4] [--bands 8] [--steps 60]"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def arg(name, default):
    return sys.argv[sys.argv.index(name) + 1] if name in sys.argv else default


class NullDist:
    """torch.distributed's surface as StripExchange uses it, every call returning at once."""
    class ReduceOp:
        SUM = 0

    class _Done:
        def wait(self):
            pass

    def isend(self, *a, **k):
        pass

    def irecv(self, *a, **k):
        pass

    class _Op:
        def __init__(self, op, tensor, peer):
            self.op, self.tensor, self.peer = op, tensor, peer

    def P2POp(self, op, tensor, peer, group=None):
        return NullDist._Op(op, tensor, peer)

    def batch_isend_irecv(self, ops):
        return [NullDist._Done()]

    def all_reduce(self, t, op=None, group=None):
        return NullDist._Done()

    def all_gather_into_tensor(self, out, inp, async_op=False, group=None):
        return NullDist._Done()

    def broadcast(self, t, src=0, group=None):
        pass


def make_renderer(api, scenes, ctx, W, H, band, config4):
    cam = api.make_camera(W, H, pos=(1.5, 2.2, 52.0), pitch=4.0, yaw=181.5)
    cfg = api.RestirRenderer.default_config(W, H, api.RENDERER_UNBIASED if config4 else api.RENDERER_BIASED)
    cfg.camera = cam
    cfg.rowBegin, cfg.rowEnd = band
    cfg.enableBumpMapping = 1
    r = api.RestirRenderer(ctx, cfg)
    if config4:
        r.set_env(api.env_make_sky(2048, 1024), 2048, 1024, 0.6, 0.4)
    return r, cfg


def run(r, stream, steps, warm=8):
    import torch
    for _ in range(warm):
        r.render_frame(stream)
    r.finish_gather(stream)
    torch.cuda.synchronize()
    host = 0.0
    t0 = time.perf_counter()
    for _ in range(steps):
        h0 = time.perf_counter()
        r.render_frame(stream)
        host += time.perf_counter() - h0
    r.finish_gather(stream)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    return wall / steps * 1e3, host / steps * 1e3


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "host"
    config4 = "--config4" in sys.argv
    nb, steps = int(arg("--bands", "8")), int(arg("--steps", "60"))
    W, H = 1920, 1080
    stub = os.path.join(ROOT, "tests", "native", "librccl_mirror.so")   # (the recording stub moves its all-gather on the host: device pointers over PCIe)
    os.environ["GFX_RCCL_LIBRARY"] = stub          # before libgfxexp loads librccl
    if mode == "latency":
        sched = arg("--schedule", "recompute")
        os.environ["GFX_GB_STRIPS_ON_MAIN"] = "1" if sched == "round5" else "0"
        os.environ["GFX_SEAM_FIRST"] = "1" if sched == "lanes" else "0"
        os.environ["GFX_STRIP_MODE"] = "3" if sched == "recompute" else "1"
    import torch
    from gfxexp_amd import api, scenes, tilesplit
    ctx = api.Context(0)
    scenes.bench_street(textured=True).upload(ctx)
    bands = tilesplit.band_rows(H, nb)
    rank = nb // 2
    band = bands[rank]
    stream = torch.cuda.current_stream().cuda_stream
    out = {"workload": "band %d of %d (rows %d-%d) of the 1920x1080 bench frame%s, one GPU" % (rank, nb, band[0], band[1], ", configs[4]" if config4 else ""),
           "mode": mode, "steps": steps}
    ids = api.RcclExchange.unique_ids(api.NUM_LANES)
    if mode == "host":
        res = {}
        # C++ callback: with the mirror's device copies behind it (a hipMemcpyAsync per received strip -- more host work than RCCL's one
        # launch per group), and with the stand-in's calls returning at once (the callback's own code)
        mirror = C.CDLL(stub)
        for label, copy in (("gfxh_rccl_exchange (C++), mirror transport at zero latency", 1), ("gfxh_rccl_exchange (C++), null transport", 0)):
            mirror.rccl_mirror_set_copy(C.c_int(copy))
            r, cfg = make_renderer(api, scenes, ctx, W, H, band, config4)
            ex = api.RcclExchange(ids, rank, nb, H)
            ex.install(r, 0)
            r.set_async_gather(True)
            wall, host = run(r, stream, steps)
            res[label] = {"host_ms_per_frame": round(host, 4), "frame_ms": round(wall, 4)}
            r.close()
        mirror.rccl_mirror_set_copy(C.c_int(1))
        # torch callback over a dist whose collectives return at once
        r, cfg = make_renderer(api, scenes, ctx, W, H, band, config4)
        sx = tilesplit.StripExchange(NullDist(), rank, nb, H, tilesplit.device_bytes, device="cuda")
        r.set_exchange(sx, 0)
        r.set_async_gather(True)
        wall, host = run(r, stream, steps)
        res["tilesplit.StripExchange (Python callback), null transport"] = {"host_ms_per_frame": round(host, 4), "frame_ms": round(wall, 4)}
        r.close()
        # no exchange at all: the launches alone
        r, cfg = make_renderer(api, scenes, ctx, W, H, band, config4)
        r.set_exchange(lambda s, d: None, 0)
        wall, host = run(r, stream, steps)
        res["no-op callback"] = {"host_ms_per_frame": round(host, 4), "frame_ms": round(wall, 4)}
        r.close()
        out["host"] = res
    else:
        mirror = C.CDLL(stub)
        mirror.rccl_mirror_set_latency_us.argtypes = [C.c_float, C.c_float]
        schedule = arg("--schedule", "recompute")
        gather_us = float(arg("--gather-us", "300"))
        r, cfg = make_renderer(api, scenes, ctx, W, H, band, config4)
        ex = api.RcclExchange(ids, rank, nb, H)
        ex.install(r, 0)
        r.set_async_gather(schedule in ("lanes", "lanes_noseam", "recompute"))
        rows = {}
        for lat in (0.0, 30.0, 60.0, 120.0):
            mirror.rccl_mirror_set_latency_us(C.c_float(lat), C.c_float(gather_us if lat > 0 else 0.0))
            best = min(run(r, stream, steps)[0] for _ in range(3))
            rows["%g" % lat] = round(best, 4)
        out["schedule"] = schedule
        out["gather_us_when_latency_nonzero"] = gather_us
        out["frame_ms_by_strip_latency_us"] = rows
        out["slowdown_at_60us"] = round(rows["60"] / rows["0"], 4)
        r.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
