#!/usr/bin/env python
"""bench.py -- ReSTIR DI (original, biased) on the Bistro-Exterior stand-in, 1920x1080, 1 spp.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one steady-state frame (frameIndex >= 1: temporal reuse active, static camera, no
accumulation): light-instance distribution, G-buffer, initial+temporal RIS, 2 spatial passes x 5
neighbours, shading -- exactly the span of the reference's GPUTimer.frame minus denoise/display
(restir_di/restir_di_main.cpp:2245-2422).  1 path = 1 pixel sample, so
    Mpaths/s = W*H*K / t / 1e6.
Inputs (scene, BVH, per-pixel state) are resident in HBM before the timed region starts.

N > 1 splits the frame into N row bands (multiples of 8 rows), one process per GPU; every rank
runs every pass on its band only, the strips of G-buffer / reservoir rows the reuse passes read
across the seams are exchanged between the passes, and the HDR bands are all-gathered over
RCCL/xGMI once per frame ("strong" scaling: total work fixed).

--config selects another BASELINE.json configuration (0-based index; the default, 2, is the one the metric is quoted on):
    1  path tracing on the bunny scene, 512x512, max path length 5 (1 GPU)
    3  NRC frame (path trace + inference + 4 training steps), hash grid, 1920x1080, street stand-in for Zero-Day (1 GPU)
    4  ReSTIR DI unbiased + 2048x1024 environment map, 1920x1080 (1..N GPUs, same band split as config 2)
--animate adds the reference command line's moving rectangle light (restir_di_main.cpp:7-12) and a slowly orbiting camera to
config 2 / 4: every frame updates the light's transform (in-place rebuild of the animated BVH subtree) and the camera.

Rank 0 prints ONE JSON line (see DESIGN.md "Measurement").
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); the streaming-copy ceiling of the box is measured live (hbm_stream_peak)


def _profile_value(name, key):
    """A figure from a file committed under profiles/ (rocprofv3 / the CPU tree comparison cannot run inside this
    process); None when the file or the key is missing."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", name)
    try:
        with open(path) as f:
            return json.load(f)[key]
    except (OSError, KeyError, ValueError):
        return None


def _profile_lines(name):
    """Non-empty lines of a JSON-lines file committed under profiles/ ([] when missing)."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", name)
    try:
        with open(path) as f:
            return [l for l in f if l.strip().startswith("{")]
    except OSError:
        return []


# committed counter passes, newest round first (tools/profile_round.sh pmc / pmc1 / pmc4 / pmca -> profiles/make_pmc_json.py), per BASELINE configuration
PMC_FILES = {2: ("r06_pmc.json", "r05_pmc.json", "r04_pmc.json"), 1: ("r06_pmc_config1.json", "r05_pmc_config1.json"), 4: ("r06_pmc_config4.json", "r05_pmc_config4.json"),
             "animate": ("r06_pmc_animate.json", "r05_pmc_animate.json")}


def sources_sha16():
    """A hash of the kernel sources this run's library was built from (gfxexp_amd/csrc/*.hip and *.h, sorted by name): the committed
    counter files carry the same hash of the tree they were measured on (profiles/make_pmc_json.py), so a counter file that no longer
    belongs to the kernels shows in the line (`pmc_matches_sources`)."""
    import hashlib
    h = hashlib.sha256()
    root = os.path.join(ROOT, "gfxexp_amd", "csrc")
    for fn in sorted(os.listdir(root)):          # the kernels and the headers they include; not capi.cpp / scene.cpp / host/ (host code)
        if fn.endswith((".hip", ".h")):
            h.update(fn.encode())
            with open(os.path.join(root, fn), "rb") as f:
                h.update(f.read())
    return h.hexdigest()[:16]


def _profiled_traffic(config=2):
    """HBM bytes per traversal launch from the committed PMC passes (rocprofv3 --pmc cannot run inside this process): value + file."""
    for name in PMC_FILES.get(config, ()) + (("r02_traffic.json", "r01_traffic.json") if config == 2 else ()):
        v = _profile_value(name, "hbm_bytes_per_launch")
        if v is not None:
            return v, "profiles/" + name
    return None, None


def _profiled_kernels(config=2):
    for name in PMC_FILES.get(config, ()):
        k = _profile_value(name, "kernels")
        if k:
            return k, "profiles/" + name
    return {}, None


def _pmc_provenance(pmc_file):
    """Which commit / which sources the counter file was measured on, and whether those are this run's sources."""
    if not pmc_file:
        return {"pmc_head": None, "pmc_sources_sha16": None, "pmc_matches_sources": None}
    name = os.path.basename(pmc_file)
    sha = _profile_value(name, "sources_sha16")
    return {"pmc_head": _profile_value(name, "git_head"), "pmc_sources_sha16": sha, "run_sources_sha16": sources_sha16(),
            "pmc_matches_sources": (sha == sources_sha16()) if sha else None}


def hbm_stream_peak(ctx):
    """Streaming rates of THIS box ("measure peak with a streaming-copy microbenchmark on the box, don't quote the
    datasheet"): gfx_stream_copy -- per-block contiguous chunks, 16-byte non-temporal loads and stores per lane, the best of the shapes
    of tools/microbench/stream_copy.hip -- over 1 GiB (4x the 256-MiB Infinity Cache), read + write bytes over the HIP-event time of
    10 copies; and the same bytes read only.  Returns (copy GB/s, read-only GB/s).  The torch byte copy of rounds 1-3 read 4.7-5.2
    TB/s; the microarchitecture guide quotes 6.29 TB/s for a float4 copy."""
    import torch
    n = 1 << 30
    a = torch.empty(n, dtype=torch.uint8, device="cuda")
    b = torch.empty(n, dtype=torch.uint8, device="cuda")
    a.fill_(1)
    stream = torch.cuda.current_stream().cuda_stream

    def timed(dst):
        for _ in range(2):
            ctx.stream_copy(dst, a.data_ptr(), n, stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10
        e0.record()
        for _ in range(reps):
            ctx.stream_copy(dst, a.data_ptr(), n, stream)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e-3

    copy_s = timed(b.data_ptr())
    ok = bool((b[::4097] == 1).all().item()) and bool((b[-64:] == 1).all().item())
    read_s = timed(0)
    del a, b
    torch.cuda.empty_cache()
    return (round(2 * n / copy_s / 1e9, 1) if ok else None), round(n / read_s / 1e9, 1)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--mse-ref-spp", type=int, default=65536, help="frames of plain-NEE reference accumulated in fp64 for the MSE figure; the metric names 64k (0 = skip)")
    ap.add_argument("--cpu-sample", type=str, default="480x270", help="resolution of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--balance-bands", type=int, default=2,
                    help="N > 1: rounds of cost-balancing the row bands during the warm-up (each round: 4 frames of every rank's band alone, "
                         "all-gather of the times, gfxh_balance_bands, band renderers re-created); 0 = equal bands")
    ap.add_argument("--plain", action="store_true", help="constant-colour materials (the round-1 workload) instead of the textured street")
    ap.add_argument("--exchange", default="auto", choices=["auto", "torch", "rccl"],
                    help="N > 1: strip-exchange transport -- rccl: the C++ gfxh_rccl_exchange (no Python between the passes, one communicator per lane); "
                         "torch: tilesplit.StripExchange over torch.distributed; auto (default): rccl when librccl loads on every rank, else torch")
    ap.add_argument("--sync-gather", action="store_true", help="N > 1: the HDR band gather on the frame's own stream instead of the gather lane underneath the next frame")
    ap.add_argument("--cluttered", action="store_true", help="secondary workload: + 70 trees of 6 000 leaf cards, cables, railings (depth complexity)")
    ap.add_argument("--bump", type=int, default=1, help="enableBumpMapping (normal maps) for the textured workload")
    ap.add_argument("--config", type=int, default=2, choices=[1, 2, 3, 4], help="BASELINE.json configs[] index (0-based); 2 = the metric's configuration")
    ap.add_argument("--animate", action="store_true", help="configs 2 / 4: moving rectangle light (restir_di_main.cpp:7-12) + slowly orbiting camera")
    ap.add_argument("--other-configs", type=int, default=1,
                    help="default run only (configs[2], static, textured, one GPU): afterwards measure configs[1], [3], [4] and --animate for --other-steps "
                         "steps each (5 warm-up frames) and report them inside the one JSON line as `other_configs` (0 = skip)")
    ap.add_argument("--spawn-timeout", type=float, default=420.0, help="N > 1 without a launcher: seconds the ranks this command starts may take before they are killed")
    ap.add_argument("--rank-timeout", type=float, default=300.0, help="N > 1: seconds a rank may spend between the start of its warm-up and the end of its timed frames before it reports where it is and exits (0 = no limit)")
    ap.add_argument("--other-steps", type=int, default=20)
    ap.add_argument("--other-timeout", type=float, default=240.0, help="seconds each other-configs child process may take before it is killed (the headline stands on its own)")
    return ap.parse_args()


BUNNY_OBJ = os.path.join(ROOT, "tests", "golden", "assets", "stanford_bunny_309_faces.obj")   # data fixture (a mesh the reference's harness names)


def light_transform(api, t_seconds):
    """The reference command line's moving rectangle light (restir_di_main.cpp:7-12: -begin-pos / -end-pos, InstanceController::
    updateBody's cosine ease, 5 s period), placed over the street of the stand-in."""
    import math
    t = 0.5 - 0.5 * math.cos(2 * math.pi * (t_seconds % 5.0) / 5.0)
    pos = ((1 - t) * -6.0 + t * 7.0, 4.5, (1 - t) * 30.0 + t * 12.0)
    return api.make_transform(pitch=-90.0, yaw=(1 - t) * 150.0 + t * 30.0, pos=pos)


def orbit_camera(api, W, H, frame):
    """A slow orbit around the bench camera's pose: +-0.75 m sideways, +-0.3 m up, +-2 degrees of yaw over 600 frames -- a few
    pixels of screen-space motion per frame, enough for the temporal hint and the temporal reuse to miss where geometry is close."""
    import math
    ph = 2 * math.pi * frame / 600.0
    return api.make_camera(W, H, pos=(1.5 + 0.75 * math.sin(ph), 2.2 + 0.3 * math.sin(2 * ph), 52.0 - 0.5 * (1 - math.cos(ph))),
                           pitch=4.0 + 0.5 * math.sin(ph), yaw=181.5 + 2.0 * math.sin(ph))


# rows a pixel's motion vector may span per frame under --animate (orbit_camera: a few pixels; the moving light: ~6 at the bench camera's
# distance): what a band renderer exchanges across its seams besides the reuse radius (gfxh_restir_set_exchange maxMotionRows)
ANIMATE_MAX_MOTION_ROWS = 24


def _respawn_under_torchrun(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks ourselves (one process per GPU, rendezvous on
    127.0.0.1), so the command works in either shape.  The ranks run in a process group of their own with a time limit; when they fail
    or outlast it with the default transport (gfxh_rccl_exchange on its lanes), ONE more attempt is made over the most conservative
    path -- torch.distributed's own nccl backend, every exchange and the band gather on the frame's stream -- and its line says so."""
    import signal
    import socket
    import subprocess
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: the only mode the host driver supports (RCCL across processes)

    def attempt(extra, limit):
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:] + extra
        child = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, start_new_session=True)
        try:
            out, _ = child.communicate(timeout=limit)
            return child.returncode, out
        except subprocess.TimeoutExpired:
            try:
                os.killpg(child.pid, signal.SIGKILL)            # the group this call started, nothing else
            except ProcessLookupError:
                pass
            out, _ = child.communicate()
            return -9, out

    sys.stdout.flush()
    rc, out = attempt([], args.spawn_timeout)
    line = next((l for l in reversed((out or "").splitlines()) if l.startswith("{")), None)
    if (rc != 0 or line is None) and args.exchange == "auto" and not args.sync_gather:
        sys.stderr.write("bench: the %d ranks %s with the default transport; one more attempt with --exchange torch --sync-gather\n"
                         % (args.gpus, "were killed after %.0f s" % args.spawn_timeout if rc == -9 else "exited with code %d" % rc))
        rc, out = attempt(["--exchange", "torch", "--sync-gather"], args.spawn_timeout)
        line = next((l for l in reversed((out or "").splitlines()) if l.startswith("{")), None)
        if rc == 0 and line is not None:
            r = json.loads(line)
            r["config"]["fallback"] = "the default transport (gfxh_rccl_exchange on its lanes) did not finish; this line is the second attempt: --exchange torch --sync-gather"
            line = json.dumps(r)
    if line is not None:
        print(line)
    else:
        sys.stdout.write(out or "")
    sys.stdout.flush()
    raise SystemExit(0 if (rc == 0 and line is not None) else (rc if rc > 0 else 1))


OTHER_CONFIGS = (("configs[1]", ["--config", "1"]), ("configs[3]", ["--config", "3"]), ("configs[4]", ["--config", "4"]), ("configs[2] --animate", ["--animate"]),
                 ("configs[2] --cluttered", ["--cluttered"]))


def _other_configs(args):
    """Every other BASELINE configuration (and the secondary workloads of configs[2]) under the same clock as the headline: a short run
    of each in a CHILD process with a time limit, after the headline measurement is complete -- a hang or a crash in one of them costs
    that entry, never the line.  Each child is this file with --other-configs 0."""
    import subprocess
    others = {}
    for name, flags in OTHER_CONFIGS:
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", str(args.other_steps), "--warmup", "5", "--mse-ref-spp", "0",
               "--cpu-sample", "0", "--other-configs", "0"] + flags
        t0 = time.perf_counter()
        try:
            out = subprocess.run(cmd, capture_output=True, text=True, timeout=args.other_timeout)
            line = next((l for l in reversed(out.stdout.splitlines()) if l.startswith("{")), None)
            if out.returncode != 0 or line is None:
                others[name] = {"error": "exit code %d: %s" % (out.returncode, (out.stderr or out.stdout)[-300:])}
                continue
            r = json.loads(line)
            roof = r.get("roofline") or {}
            others[name] = {"metric": r["metric"], "value": r["value"], "unit": r["unit"], "ms_per_step": r["ms_per_step"], "steps": r["steps"], "warmup": r["warmup"],
                            "workload": r["config"]["workload"], "width": r["config"]["width"], "height": r["config"]["height"],
                            "roofline": {k: roof.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "frac_nominal_hbm", "frac_hbm_counter", "traffic", "avg_launch_ms", "valu",
                                                                "pmc_head", "pmc_matches_sources", "infer_ms_per_frame") if k in roof},
                            "kernels_ms_per_frame": r.get("kernels_ms_per_frame"), "seconds": round(time.perf_counter() - t0, 1)}
        except subprocess.TimeoutExpired:
            others[name] = {"error": "killed after %.0f s (--other-timeout)" % args.other_timeout}
        except Exception as e:               # the headline stands on its own
            others[name] = {"error": repr(e)[:300]}
    return others


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        _respawn_under_torchrun(args)            # does not return
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if world > 1 and args.config in (1, 3):
        raise SystemExit(f"--config {args.config} is a single-GPU configuration in BASELINE.json")
    # GFX_BENCH_ONE_GPU=1: every rank on device 0, collectives staged through host memory over gloo (tilesplit.HostStaged) -- a functional
    # check of the multi-rank frame loop on a one-GPU box (RCCL refuses two ranks on a device); its numbers mean nothing
    one_gpu = world > 1 and os.environ.get("GFX_BENCH_ONE_GPU") == "1"
    if one_gpu:
        local_rank = 0
    elif world > 1 and torch.cuda.device_count() < world:
        raise SystemExit(f"--gpus {world} but this box has {torch.cuda.device_count()} GPU(s) (GFX_BENCH_ONE_GPU=1 runs the ranks on one device as a functional check)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_gpu:
            from gfxexp_amd import tilesplit
            if args.exchange == "rccl":
                raise SystemExit("GFX_BENCH_ONE_GPU=1 runs over the torch.distributed exchange (--exchange torch)")
            args.exchange = "torch"
            dist.init_process_group("gloo", rank=rank, world_size=world)
            dist = tilesplit.HostStaged(dist)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    result = run_config(args, rank, local_rank, world, dist)
    headline_only = args.config == 2 and not (args.animate or args.plain or args.cluttered)
    if rank == 0 and world == 1 and args.other_configs and headline_only:
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        result["other_configs"] = _other_configs(args)
    if world > 1 and args.other_configs and headline_only:
        # N > 1: BASELINE's configs[4] IS the split across the GPUs of the node (unbiased estimator + environment map), so the same ranks
        # run a short measurement of it after the headline is complete.  The headline must not depend on it: when the ranks do not finish it
        # in time, every rank's watchdog ends its process with code 0 and rank 0 prints the headline with the reason first.
        import copy
        extra = copy.copy(args)
        extra.config, extra.steps, extra.warmup, extra.mse_ref_spp, extra.cpu_sample = 4, args.other_steps, 5, 0, "0"
        extra.rank_timeout = min(args.rank_timeout, 150.0) if args.rank_timeout > 0 else 150.0

        def give_up(why):
            if rank == 0:
                result["other_configs"] = {"configs[4]": {"error": why}}
                print(json.dumps(result))
                sys.stdout.flush()
            os._exit(0)
        try:
            r4 = run_config(extra, rank, local_rank, world, dist, on_timeout=give_up)
            if rank == 0:
                result["other_configs"] = {"configs[4]": {"metric": r4["metric"], "value": r4["value"], "unit": r4["unit"], "n_gpus": world, "ms_per_step": r4["ms_per_step"],
                                                          "steps": r4["steps"], "warmup": r4["warmup"], "workload": r4["config"]["workload"],
                                                          "bands": r4["config"]["bands"], "band_balancing": r4["config"]["band_balancing"], "exchange": r4["config"]["exchange"],
                                                          "gathered_frame_matches_bands": r4.get("gathered_frame_matches_bands")}}
        except BaseException as e:          # this rank alone may have failed: the others end at their watchdogs; no collective after this point
            give_up("rank %d: %r" % (rank, e))
    if rank == 0:
        print(json.dumps(result))
        sys.stdout.flush()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def run_config(args, rank, local_rank, world, dist, on_timeout=None):
    """One measurement: scene, renderer, warm-up, the timed steps, and (rank 0, one GPU) the roofline / MSE / CPU legs.  Returns the JSON object.
    on_timeout(reason): what a rank's watchdog does instead of leaving with code 3 (the secondary measurement of an N > 1 run)."""
    import torch
    from gfxexp_amd import api
    from gfxexp_amd import tilesplit
    from gfxexp_amd import scenes            # the measured path imports nothing from tests/ or oracle/

    W, H = args.width, args.height
    if args.config == 1:
        W = H = 512
    t0 = time.time()
    textured = not args.plain
    ctx = api.Context(local_rank)
    light_slot = None
    if args.config == 1:
        hs = scenes.bunny_scene(BUNNY_OBJ)
    else:
        hs = scenes.bench_street(textured=textured, cluttered=args.cluttered)
        if args.animate:
            light_slot = hs.add_instance(hs.add_rectangle(1.5, 1.5, (60, 60, 60)), light_transform(api, 0.0))
    counts = hs.counts()
    hs.upload(ctx)
    if light_slot is not None:
        ctx.instance_set_dynamic(light_slot)
    bands = tilesplit.band_rows(H, world) if world > 1 else None
    band = tilesplit.band_for_rank(H, world, rank)
    street = (f"procedural street stand-in ({counts['triangles']} instanced triangles, {counts['insts']} instances, 2100 emitter instances), "
              + ("textured materials (albedo / smoothness / normal maps with bump mapping, float emittance maps on the signs), " if textured else "constant-colour materials, ")
              + ("+ trees of leaf cards, cables and railings (depth-complexity variant), " if args.cluttered else ""))
    if args.config == 1:
        cam = api.make_camera(W, H, pos=(1.5, 5.0, 14.0), pitch=12.0, yaw=186.0)
        cfg = api.RestirRenderer.default_config(W, H, api.RENDERER_PATH_TRACE)
        cfg.camera = cam
        renderer = api.RestirRenderer(ctx, cfg)
        metric = "Mpaths/s, path tracing (BVH8 traversal + BSDF) 512x512 1 spp, stanford_bunny_309_faces"
        workload = ("configs[1] (0-based index into BASELINE.json): baseline path tracer, bunny (309 faces) + ground + 2 rectangle lights, "
                    f"{counts['triangles']} triangles, max path length 5, NEE + MIS")
    elif args.config == 3:
        cam = api.make_camera(W, H, pos=(2.0, 5.0, 26.0), pitch=6.0, yaw=184.0)
        cfg = api.NrcRenderer.default_config(W, H, hs.bounds())
        cfg.camera = cam
        renderer = api.NrcRenderer(ctx, cfg)
        metric = "Mpaths/s, NRC frame (path trace + cache inference + 4 training steps) 1920x1080 1 spp, Zero-Day stand-in"
        workload = ("configs[3] (0-based index into BASELINE.json): neural radiance caching, hash-grid encoding, fully fused 64-wide MLP (2 hidden layers) "
                    "on bf16 MFMA, training on; " + street + "Zero-Day is not in the reference checkout")
    else:
        unbiased = args.config == 4
        cam = api.make_camera(W, H, pos=(1.5, 2.2, 52.0), pitch=4.0, yaw=181.5)
        cfg = api.RestirRenderer.default_config(W, H, api.RENDERER_UNBIASED if unbiased else api.RENDERER_BIASED)
        cfg.camera = cam
        cfg.enableBumpMapping = int(textured and args.bump)
        cfg.rowBegin, cfg.rowEnd = band
        renderer = api.RestirRenderer(ctx, cfg)
        sky = api.env_make_sky(2048, 1024) if unbiased else None
        if unbiased:
            renderer.set_env(sky, 2048, 1024, 0.6, 0.4)
        metric = ("Mpaths/s, ReSTIR DI (original, unbiased) + environment light 1920x1080 1 spp, Bistro-Exterior stand-in" if unbiased
                  else "Mpaths/s, ReSTIR DI (original, biased) 1920x1080 1 spp, Bistro-Exterior stand-in")
        workload = (f"configs[{args.config}] (0-based index into BASELINE.json): ReSTIR DI " + ("unbiased, 2048x1024 sky + sun environment map, " if unbiased else "biased, ")
                    + street.replace("procedural street stand-in", "procedural street stand-in for Bistro Exterior")
                    + ("32 candidates, temporal + 1x3 spatial reuse with MIS rays, radius 20, visibility reuse" if unbiased
                       else "32 candidates, temporal + 2x5 spatial reuse, radius 20, visibility reuse")
                    + (", moving rectangle light (animated BVH subtree rebuilt per frame) + orbiting camera" if args.animate else ""))
    accel_stats = ctx.accel_stats(renderer.accel()) if hasattr(renderer, "accel") else None
    setup_s = time.time() - t0
    stream = torch.cuda.current_stream().cuda_stream

    exchange = None
    transport = None
    stage = ["setup"]
    measured = [None]          # the line once the timed frames are in: what a failure of a LATER leg must not cost

    def leave_with_the_measurement(why):
        """A leg after the timed frames (the gathered-frame check, the MSE reference) failed or hung on this rank: no collective is safe
        any more, so rank 0 prints the measurement with the reason and every rank ends its process with code 0."""
        if on_timeout is not None:
            on_timeout(why)                  # the secondary measurement of an N > 1 run: the caller prints the headline
        if rank == 0 and measured[0] is not None:
            measured[0]["after_the_timed_frames"] = {"error": why}
            print(json.dumps(measured[0]))
            sys.stdout.flush()
        os._exit(0 if measured[0] is not None else 3)

    def arm_watchdog(limit):
        """A collective that never completes (a transport that deadlocks on hardware nobody could test on) must cost the run minutes and
        leave a reason, not hang the node: the rank says where it is and exits, and the launcher ends the others."""
        if world == 1 or args.rank_timeout <= 0:
            return None
        import threading

        def expired():
            why = "rank %d did not finish within %.0f s (--rank-timeout); it is in: %s; transport: %s" % (rank, limit, stage[0], transport)
            sys.stderr.write("bench: %s\n" % why)
            sys.stderr.flush()
            if measured[0] is not None:
                leave_with_the_measurement(why)
            if on_timeout is not None:
                on_timeout(why)
            os._exit(3)
        timer = threading.Timer(limit, expired)
        timer.daemon = True
        timer.start()
        return timer
    watchdog = arm_watchdog(args.rank_timeout)
    if world > 1:
        # strip exchange: the rows the reuse passes read across the seams travel between the passes -- the G-buffer strips on the G-buffer
        # lane behind the pipelined G-buffer pass, the reservoir strips once per frame on the frame's stream behind the candidate pass
        # (the first spatial pass of two is recomputed on its halo: stripMode 3) -- and the HDR bands are all-gathered on the gather lane
        # underneath the next frame (gfxexp_host.h gfxh_lane)
        motion_rows = ANIMATE_MAX_MOTION_ROWS if args.animate else 0       # static camera and scene: no motion rows
        L = api.lib()
        L.gfxh_rccl_last_error.restype = C.c_char_p
        use_rccl = args.exchange in ("auto", "rccl")
        ids = None
        stage[0] = "the transport's setup (ncclUniqueId broadcast, ncclCommInitRank per lane)"
        if use_rccl:
            # rank 0 draws one ncclUniqueId per lane (this also answers whether librccl loads here); the verdict and the ids go to every
            # rank over the torch process group, so the ranks choose the same transport before anyone enters ncclCommInitRank
            ok = torch.ones(1, dtype=torch.int32, device="cuda")
            buf = torch.zeros(128 * api.NUM_LANES, dtype=torch.uint8, device="cuda")
            try:
                mine = api.RcclExchange.unique_ids(api.NUM_LANES)
                if rank == 0:
                    buf.copy_(torch.frombuffer(bytearray(mine), dtype=torch.uint8))
            except api.GfxError as e:
                ok.zero_()
                if rank == 0:
                    sys.stderr.write("bench: librccl does not load (%s)\n" % e)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            dist.broadcast(buf, src=0)
            use_rccl = bool(ok.item())
            ids = bytes(buf.cpu().numpy().tobytes())
            if not use_rccl and args.exchange == "rccl":
                raise SystemExit("--exchange rccl: librccl does not load on every rank")

        one_gpu = os.environ.get("GFX_BENCH_ONE_GPU") == "1"
        rccl_ex = None
        if use_rccl:
            # the communicators: made once (an ncclUniqueId serves one ncclCommInitRank).  A failure that every rank sees (a librccl that
            # refuses the call) sends all of them to the torch.distributed transport together; --exchange rccl makes it an error instead
            made = torch.ones(1, dtype=torch.int32, device="cuda")
            try:
                rccl_ex = api.RcclExchange(ids, rank, world, H)
            except api.GfxError as e:
                made.zero_()
                sys.stderr.write("bench: rank %d: %s\n" % (rank, e))
            dist.all_reduce(made, op=dist.ReduceOp.MIN)
            if not bool(made.item()):
                if args.exchange == "rccl":
                    raise SystemExit("--exchange rccl: gfxh_rccl_create_lanes failed")
                if rccl_ex is not None:
                    rccl_ex.close()
                rccl_ex, use_rccl = None, False

        def install(r, bands_now):
            """The transport for one band renderer (re-installed when the bands are re-cut)."""
            if use_rccl:
                api.check_bands(cfg, bands_now, motion_rows)                      # the same verdict on every rank, before the first collective
                rccl_ex.set_bands(bands_now)
                rccl_ex.install(r, motion_rows)                                   # no Python between the passes
                ex = rccl_ex
            else:
                groups = {} if one_gpu else tilesplit.make_lane_groups(dist)
                ex = tilesplit.StripExchange(dist, rank, world, H, tilesplit.device_bytes, device="cuda", bands=bands_now, lane_groups=groups)
                r.set_exchange(ex, motion_rows)
            r.set_async_gather(not args.sync_gather)
            return ex
        transport = "gfxh_rccl_exchange (C++, %d communicators)" % api.NUM_LANES if use_rccl else "tilesplit.StripExchange (torch.distributed)"
        # Equal rows are not equal work (sky rows are cheap).  Each round: time this rank's band ALONE over a few frames -- behind a callback
        # that moves nothing, because a frame with its exchanges takes as long as the slowest neighbour's and says nothing about this
        # band's cost (the seam rows then hold stale data: the same amount of work) --, all-gather the times, cut the frame where
        # gfxh_balance_bands says (the same call with the same numbers on every rank) and start over with a band renderer for the new rows.
        # All inside the untimed warm-up; the timed frames use the final partition and a renderer that has seen none of this.
        stage[0] = "band balancing (untimed)"
        strip_rows_needed = 8 * ((int(cfg.spatialNeighborRadius) * max(1, int(cfg.numSpatialReusePasses)) + motion_rows + 7) // 8)
        balancing = []

        def nothing(stream_, d):
            return None
        for _ in range(max(0, args.balance_bands)):
            renderer.set_exchange(nothing, motion_rows)
            renderer.set_async_gather(False)
            for _ in range(2):
                renderer.render_frame(stream)
            torch.cuda.synchronize()
            batches = []
            for _ in range(3):                       # the least disturbed of three batches of four frames
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(4):
                    renderer.render_frame(stream)
                e1.record()
                torch.cuda.synchronize()
                batches.append(e0.elapsed_time(e1) / 4)
            mine = torch.tensor([min(batches)], dtype=torch.float32, device="cuda")
            times = torch.zeros(world, dtype=torch.float32, device="cuda")
            dist.all_gather_into_tensor(times, mine)
            band_ms = [float(t) for t in times.cpu()]
            balancing.append({"bands": [list(b) for b in bands], "band_ms_alone": [round(t, 4) for t in band_ms]})
            new_bands = api.balance_bands(H, bands, band_ms, min_rows=max(24, strip_rows_needed))
            # a cut that leaves a band below 0.7x or above 1.4x its equal share is a measurement gone wrong, not a scene (the sky-heavy
            # first band of the bench frame is 1.24x, its cheapest 0.82x: profiles/r06_band_compute_bound*.json): the partition stays
            share = H / world
            if any(not (0.7 * share <= e - b <= 1.4 * share) for b, e in new_bands):
                balancing[-1]["rejected"] = [list(b) for b in new_bands]
                new_bands = bands
            changed = new_bands != bands
            bands = new_bands
            renderer.close()                         # (also when nothing moved: this one's seams have seen no neighbour)
            cfg.rowBegin, cfg.rowEnd = bands[rank]
            renderer = api.RestirRenderer(ctx, cfg)
            if args.config == 4:
                renderer.set_env(sky, 2048, 1024, 0.6, 0.4)
            if not changed:
                break
        exchange = install(renderer, bands)

    frame_no = [0]

    def frame():
        if args.animate:
            # InstanceController::update + updateASs of the reference's frame loop (restir_di_main.cpp:2258-2264), at 60 frames per second
            ctx.instance_set_transform(light_slot, light_transform(api, frame_no[0] / 60.0))
            renderer.rebuild_accel(stream)
            renderer.set_camera(orbit_camera(api, W, H, frame_no[0]))
        renderer.render_frame(stream)
        frame_no[0] += 1

    def barrier():
        if args.config == 3:
            renderer.network()                            # joins the training stream of the last frame
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # frame 0 starts the sequence (no temporal reuse); it is part of the untimed warm-up
    stage[0] = "warm-up frames"
    for _ in range(max(1, args.warmup)):
        frame()
    barrier()
    stage[0] = "timed frames"
    t_start = time.perf_counter()
    for _ in range(args.steps):
        frame()
    if exchange is not None:
        renderer.finish_gather(stream)                # the last frame's bands are in place on every rank
    barrier()
    elapsed = time.perf_counter() - t_start
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if watchdog is not None:
        watchdog.cancel()
    ms_per_step = 1e3 * elapsed / args.steps
    mpaths = W * H * args.steps / elapsed / 1e6

    result = {
        "metric": metric,
        "value": round(mpaths, 3), "unit": "Mpaths/s", "n_gpus": world, "steps": args.steps, "warmup": max(1, args.warmup),
        # the frame is fixed and split across the GPUs: total work does not grow with N (also the label of the N = 1 point of that curve)
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f32" if args.config != 3 else "f32 path tracing + bf16 MFMA network (fp32 accumulate, fp32 master weights)", "data": "synthetic",
        "config": {"workload": workload,
                   "width": W, "height": H, "spp": 1, "parallelism": (f"row-bands x{world}" + (" on ONE device, host-staged gloo (GFX_BENCH_ONE_GPU: a functional check, not a measurement)" if os.environ.get("GFX_BENCH_ONE_GPU") == "1" else "")) if world > 1 else "single GPU",
                   "bands": bands, "band_balancing": (balancing if world > 1 else None), "exchange": transport, "band_gather": (None if world == 1 else "caller's stream" if args.sync_gather else "gather lane, underneath the next frame"),
                   "bvh": {"nodes": accel_stats["nodes"], "triangles": accel_stats["triRecords"], "levels": accel_stats["maxDepth"]} if accel_stats else None,
                   "light_table": ctx.lights_table_info(),
                   # scene.setupLightInstDistribution runs every frame in the reference (restir_di_main.cpp:2303-2309); here the call is made
                   # every frame and returns early while no instance has moved (lights.hip: instDistValid) -- same values, about 0.04 ms not spent
                   "light_inst_distribution": "rebuilt every frame (an emitter instance moves)" if args.animate else "cached (static scene): built once, the per-frame call returns early"},
        "setup_s": round(setup_s, 2),
    }

    if rank == 0 and world == 1:
        result["gpu_bvh_build_ms"] = gpu_bvh_build_ms(ctx, stream)
        if not args.no_roofline:
            ctx.tunable_set("pt_overlap", 0)          # per-kernel durations: kernels that share the GPU with another stream read longer than they are
            if args.config == 3:
                result["roofline"], result["kernels_ms_per_frame"], result["nrc"] = roofline_nrc(ctx, renderer, stream, W, H)
            else:
                # per-kernel durations come from a second renderer that runs every pass on ONE stream: with the frame
                # pipelining of the timed renderer the next frame's G-buffer pass overlaps the spatial / shading passes
                # and both read longer than they are
                os.environ["GFX_SERIAL_FRAMES"] = "1"
                serial = api.RestirRenderer(ctx, cfg)
                del os.environ["GFX_SERIAL_FRAMES"]
                if args.config == 4:
                    serial.set_env(sky, 2048, 1024, 0.6, 0.4)
                for _ in range(3):
                    serial.render_frame(stream)
                result["roofline"], result["kernels_ms_per_frame"] = roofline(ctx, serial, stream, args.steps, W, H, args.config, animate=args.animate)
                serial.close()
            ctx.tunable_set("pt_overlap", 1)
        if args.mse_ref_spp > 0 and args.config == 2 and not args.animate:
            result["mse"] = mse_vs_reference(ctx, hs, renderer, cam, W, H, args.mse_ref_spp)
        if args.cpu_sample not in ("0", ""):
            if args.config in (2, 4) and not args.animate:
                result["cpu_baseline"] = cpu_baseline(hs, cam, args.cpu_sample, W, H, unbiased=args.config == 4, env=(sky, 2048, 1024, 0.6, 0.4) if args.config == 4 else None)
                result["cpu_baseline"]["gpu_bvh_build_ms"] = result["gpu_bvh_build_ms"]      # beside builds.*.bvh_build_s (the CPU SAH build, seconds)
            elif args.config == 1:
                result["cpu_baseline"] = cpu_baseline_path_tracer(hs, cam, W, H)
            else:
                result["cpu_baseline"] = {"value": None, "unit": "Mpaths/s", "cores": 0, "kind": "port",
                                          "sample": "not timed for this line: see the default line (configs[2], static) for the CPU restatement on this host"}
    if world > 1:
        # what the ranks can say about the frame together: rank 0's gathered frame holds every rank's band bit for bit, and the MSE leg.
        # From here on a failure costs these legs, not the measurement (leave_with_the_measurement).
        import torch
        measured[0] = result
        stage[0] = "the gathered frame against the ranks' bands"
        watchdog = arm_watchdog(args.rank_timeout)
        try:
            if os.environ.get("GFX_BENCH_TEST_FAIL_RANK") == str(rank):          # tests/test_gpu_strip_exchange.py: a rank that dies here
                raise RuntimeError("injected by GFX_BENCH_TEST_FAIL_RANK")
            n_words = W * H * 4
            frame_words = _device_view(renderer.beauty_ptr(), n_words).view(torch.int32)

            def band_sum(b):
                return frame_words[b[0] * W * 4: b[1] * W * 4].to(torch.int64).sum().reshape(1)
            mine = band_sum(bands[rank])
            every = torch.zeros(world, dtype=torch.int64, device="cuda")
            dist.all_gather_into_tensor(every, mine)
            if rank == 0:
                held = torch.cat([band_sum(b) for b in bands])
                result["gathered_frame_matches_bands"] = bool(torch.equal(held, every))
            if watchdog is not None:
                watchdog.cancel()
            if args.mse_ref_spp > 0 and args.config == 2 and not args.animate:
                stage[0] = "the MSE reference (every rank accumulates the reference of its rows)"
                # (about ref_spp x a band's plain-NEE frame: ~150 s / world at 64k frames; the limit scales with it)
                watchdog = arm_watchdog(args.rank_timeout + 0.01 * args.mse_ref_spp)
                m = mse_vs_reference(ctx, hs, renderer, cam, W, H, args.mse_ref_spp, band=tuple(bands[rank]), dist=dist)
                if rank == 0:
                    result["mse"] = m
                if watchdog is not None:
                    watchdog.cancel()
        except BaseException as e:           # (this rank alone, perhaps: the others would wait in a collective for ever)
            leave_with_the_measurement("rank %d: %r" % (rank, e))
        measured[0] = None
    renderer.close()
    ctx.close()
    return result


def gpu_bvh_build_ms(ctx, stream):
    """The HIP LBVH -> SAH-DP BVH8 build of the uploaded scene (csrc/lbvh.hip), the GPU side of SURVEY 8(d)'s "CPU SAH vs HIP LBVH build
    times side by side": wall time of gfx_accel_build into a NEW handle, host-paired (synchronise, build, synchronise) -- the first
    build of the run also allocates the builder's scratch, the following ones are what a rebuild costs."""
    import torch
    times = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx.accel_build(stream)
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) * 1e3)
    return {"first_ms": round(times[0], 3), "warm_ms": round(min(times[1:]), 3), "how": "gfx_accel_build into a new handle, host-paired wall time; warm = best of two with the builder's scratch allocated"}


def roofline(ctx, renderer, stream, steps, W, H, config=2, animate=False):
    """Dominant kernel family by GPU time; achieved = algorithmic bytes per launch / mean launch duration.
    Durations: HIP events on the launch stream around every kernel (gfx_timing_*), live in this run, passes
    serialised on one stream.  Algorithmic bytes of the traversal kernels: node fetches x (64 + 16) B + triangle
    fetches x 64 B + rays x (32 B in + result out), counted by the counting instantiation of the same kernel.
    k_initial_candidates gets its own entry: 32 candidates x (16-B interval entry + 64-B emitter record + 48-B
    normal matrix) + 64 B of pixel state in and 72 B out per pixel."""
    import torch
    ctx.timing_enable(True)
    n = max(4, min(steps, 16))
    for _ in range(n):
        renderer.render_frame(stream)
    torch.cuda.synchronize()
    timings = ctx.timing_collect()
    ctx.timing_enable(False)
    ctx.counters_enable(True)
    ctx.counters_read(reset=True)
    renderer.render_frame(stream)
    torch.cuda.synchronize()
    c = ctx.counters_read(reset=True)
    diag = ctx.trace_diag_read(reset=True)
    ctx.counters_enable(False)
    per_frame = {k: round(ms / n, 4) for k, (ms, calls) in sorted(timings.items(), key=lambda kv: -kv[1][0])}
    # a small launch (configs[1]: 512 x 512 is one round of waves) runs with the traversal INSIDE the per-pixel kernels (k_*_fused,
    # csrc/trace_local.hip.h): those kernels are then the ones priced; the counting launches below always take the k_trace form
    # The G-buffer pass of every renderer, and every ray pass of a small launch (configs[1]: 512 x 512 is one round of waves), run with
    # the traversal INSIDE the per-pixel kernel (k_*_fused, csrc/trace_local.hip.h).  configs[2] at full size keeps two k_trace<any>
    # launches per frame -- the visibility ray of the selected candidate and the final shadow ray, 0.9 of the frame's 1.25 ms of
    # traversal: that kernel is the one priced here, with its own bytes; the fused G-buffer kernel gets a line of its own below.
    # The other configurations price all their traversal kernels together.  (The counting frame always takes the k_trace form.)
    fused_form = sorted(k for k in timings if k.endswith("_fused"))
    any_only = config == 2 and "trace_any" in timings and "trace_closest" not in timings
    is_trav = (lambda k: k == "trace_any") if any_only else (lambda k: k.startswith("trace_") or k.endswith("_fused"))
    trav_ms = sum(ms for k, (ms, calls) in timings.items() if is_trav(k)) / n
    trav_launches = sum(calls for k, (ms, calls) in timings.items() if is_trav(k)) / n
    # (configs[2]: one frame = 1 closest-hit traversal (16 B out) + 2 any-hit launches (4 B out); the other configurations launch more)
    rays_closest, rays_any = c["closest"]["rays"], c["any"]["rays"]
    bytes_closest = c["closest"]["nodeFetches"] * (64 + 16) + c["closest"]["triFetches"] * 64 + rays_closest * (32 + 16)   # node = 64-B record + 16-B link
    bytes_any = c["any"]["nodeFetches"] * (64 + 16) + c["any"]["triFetches"] * 64 + rays_any * (32 + 4)
    bytes_frame = bytes_any if any_only else bytes_closest + bytes_any
    achieved = bytes_frame / (trav_ms * 1e-3) / 1e9
    nodes_per_primary = c["closest"]["nodeFetches"] / max(1, rays_closest)
    # the same launch time priced with the node / triangle fetches a reference-style SAH + spatial-split tree needs
    # for the same primary rays (tools/bvh_quality.py, committed): what the kernel achieves in units of useful work
    sah = _profile_value("r02_bvh_quality.json", "sah_tree") or {}
    frac_sah = None
    if sah.get("nodes_per_ray") and sah.get("tris_per_ray") and c["closest"]["triFetches"]:
        ours_bytes = nodes_per_primary * 80 + c["closest"]["triFetches"] / max(1, rays_closest) * 64
        sah_bytes = sah["nodes_per_ray"] * 80 + sah["tris_per_ray"] * 64
        frac_sah = round(achieved / HBM_PEAK_GBS * sah_bytes / ours_bytes, 4)
    peak_measured, peak_read_only = hbm_stream_peak(ctx)
    # what actually bounds the kernel (committed rocprofv3 PMC passes, profiles/make_pmc_json.py): VALU issue.
    # The BVH is served from L2 / Infinity Cache (`traffic` is 20-30x below the algorithmic bytes), so the SURVEY 8(d) byte
    # roof is nominal; the hardware-side figure is the share of VALU issue slots used and how many lanes each instruction carries.
    pmc_key = "animate" if (animate and config == 2) else config
    pmc, pmc_file = _profiled_kernels(pmc_key)
    traffic, traffic_file = _profiled_traffic(pmc_key)
    valu = None
    # which kernel's counters stand for the configuration's traversal: k_trace<any> where the frame's traversal is mostly any-hit launches
    # (configs[2]: 2 per frame, configs[4]: 3 incl. the MIS rays of the unbiased spatial pass), the one-kernel path tracer for configs[1]
    valu_kernel = "k_trace_any" if config in (2, 4) else "k_pt_fused" if config == 1 else None
    if valu_kernel in pmc:
        ka = pmc[valu_kernel]
        valu = {"source": pmc_file + " (rocprofv3 --pmc of this configuration's bench command; per-launch means of %s)" % valu_kernel.replace("k_trace_any", "k_trace<any>"),
                "busy": ka["valu_busy"], "lane_fraction": ka["lane_fraction"],
                # (the counting frame runs all three traversals as k_trace launches; the per-iteration figure needs the counters of all three)
                "insts_per_wave_iteration": round((2 * ka["valu_insts"] + pmc["k_trace_closest"]["valu_insts"]) / max(1, diag["iterations"]), 1) if "k_trace_closest" in pmc else None,
                "useful_fraction_of_valu_peak": None}
        for extra in ("k_gbuffer_fused", "k_initial_candidates", "k_spatial_unbiased"):
            if extra in pmc and extra != valu_kernel:
                valu[extra[2:]] = {k: pmc[extra].get(k) for k in ("valu_busy", "lane_fraction", "valu_insts", "l2_hit", "hbm_bytes")}
        # busy is SQ_ACTIVE_INST_VALU / (8 x SQ_BUSY_CYCLES); the r04 passes read 1.00-1.08 for the traversal kernels (the busy-cycle
        # normalisation is good to a few per cent): a SIMD cannot issue more than all the time, so the product is taken with min(busy, 1)
        valu["useful_fraction_of_valu_peak"] = round(min(valu["busy"], 1.0) * valu["lane_fraction"], 4)
    # The headline fraction is the one that binds: the share of the SIMDs' VALU issue slots x lanes that carry a ray (`valu`), when the
    # committed counter pass covers this configuration; the SURVEY 8(d) byte figure stays beside it as frac_nominal_hbm.
    useful = valu["useful_fraction_of_valu_peak"] if valu else None
    roof = {"bound": "valu" if useful is not None else "hbm (nominal, SURVEY 8d; no VALU counter pass committed for this configuration)",
            "kernel": "k_trace<any> (software BVH8 traversal; %g launches per frame: visibility of the selected candidate, final shadow ray)" % round(trav_launches, 1) if any_only
                      else ("%s (%g launches per frame; the k_*_fused kernels are per-pixel kernels with the software BVH8 traversal inside, csrc/trace_local.hip.h; "
                            "bytes = the traversal's, counted by a k_trace-form frame of the same rays)"
                            % (" + ".join(sorted("k_" + k for k in timings if is_trav(k))), round(trav_launches, 1))) if fused_form
                      else "k_trace<closest|any> (software BVH8 traversal; %g launches per frame)" % round(trav_launches, 1),
            "achieved": useful if useful is not None else round(achieved, 1),
            "peak": 1.0 if useful is not None else HBM_PEAK_GBS,
            "unit": "share of VALU issue slots x lanes carrying a ray (VALU busy x active lanes per instruction)" if useful is not None else "GB/s",
            "frac": useful if useful is not None else round(achieved / HBM_PEAK_GBS, 4),
            "achieved_nominal_hbm": round(achieved, 1), "peak_nominal_hbm": HBM_PEAK_GBS, "frac_nominal_hbm": round(achieved / HBM_PEAK_GBS, 4),
            "peak_measured": peak_measured, "peak_measured_read_only": peak_read_only,
            "peak_measured_how": "gfx_stream_copy (per-block contiguous chunks, 16-byte non-temporal loads / stores per lane) over 1 GiB: read + write bytes, and the same bytes read only; HIP events, live in this run",
            "frac_of_peak_measured": round(achieved / peak_measured, 4) if peak_measured else None,
            "frac_of_peak_measured_read_only": round(achieved / peak_read_only, 4) if peak_read_only else None,
            "valu": valu,
            "traffic": traffic,
            # the three fractions of the 8 TB/s HBM roof side by side: `frac_nominal_hbm` prices the algorithmic bytes (every node and triangle
            # fetch as if it came from HBM), `frac_hbm_counter` the bytes the memory-side counters saw (the BVH is served from L2 / Infinity
            # Cache), `frac` the figure that binds (VALU issue x lanes carrying a ray)
            "frac_hbm_counter": round(traffic / (trav_ms / max(trav_launches, 1) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic else None,
            "traffic_source": (traffic_file + ": HBM bytes per launch by PMC (FETCH_SIZE / WRITE_SIZE passes of this configuration's bench command), mean over the k_trace<any> launches") if traffic_file else None,
            "node_visits_per_ray": {"primary": round(nodes_per_primary, 3),
                                    "shadow": round(c["any"]["nodeFetches"] / max(1, rays_any), 3),
                                    "sah_tree_primary": sah.get("nodes_per_ray")},
            "tri_tests_per_ray": {"primary": round(c["closest"]["triFetches"] / max(1, rays_closest), 3),
                                  "shadow": round(c["any"]["triFetches"] / max(1, rays_any), 3),
                                  "sah_tree_primary": sah.get("tris_per_ray")},
            "frac_sah_normalised": frac_sah,
            "scheduling": {"wave_iterations": diag["iterations"], "lane_occupancy": round(diag["itemLanes"] / max(1, 64 * diag["iterations"]), 4),
                           "drain_iteration_share": round(diag["drainIterations"] / max(1, diag["iterations"]), 4),
                           "drain_lane_occupancy": round(diag["drainItemLanes"] / max(1, 64 * diag["drainIterations"]), 4),
                           # where the waves' clock cycles go in the counting launches (s_memtime around the sections of the loop)
                           "wave_cycle_shares": {"ray_refill": round(diag["refillCycles"] / max(1, diag["waveCycles"]), 4),
                                                 "item_fetch_wait": round(diag["fetchCycles"] / max(1, diag["waveCycles"]), 4),
                                                 "item_processing": round(diag["processCycles"] / max(1, diag["waveCycles"]), 4)},
                           "wave_cycles_per_iteration": round(diag["waveCycles"] / max(1, diag["iterations"]), 1)},
            "algorithmic_bytes_per_launch": int(bytes_frame / max(trav_launches, 1)),
            "avg_launch_ms": round(trav_ms / max(trav_launches, 1), 4),
            "per_frame": {"node_fetches": int(c["nodeFetches"]), "tri_fetches": int(c["triFetches"]), "rays": int(c["rays"]),
                          "stack_spills": int(c["spills"])}}
    roof.update(_pmc_provenance(pmc_file))
    gb_ms = timings.get("gbuffer_fused", (0.0, 0))[0] / n
    if gb_ms > 0 and any_only:
        roof["gbuffer_fused"] = {"kernel": "k_gbuffer_fused (primary ray -> closest hit with the temporal hint -> G-buffer resolve, one kernel)", "ms": round(gb_ms, 4),
                                 "algorithmic_bytes_per_launch": int(bytes_closest + W * H * 88),
                                 "achieved_nominal_hbm": round((bytes_closest + W * H * 88) / (gb_ms * 1e-3) / 1e9, 1),
                                 "frac_nominal_hbm": round((bytes_closest + W * H * 88) / (gb_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                 "bytes": "the traversal's (node / triangle fetches of the primary rays) + 88 B of G-buffer written per pixel (SURVEY 8d)"}
    init_ms = timings.get("initial_candidates", (0.0, 0))[0] / n
    if init_ms > 0:
        # The candidate pass gathers from ~3.4 MB of L2-resident tables: what bounds it is the rate at which the L2s hand scattered
        # 64-byte sectors to the CUs (tools/microbench/l2_gather.hip -> profiles/r03_l2_gather.jsonl: 218-243 G sectors/s out of a
        # 4-MB table, whatever the access shape), not HBM and not arithmetic (profiles/r03_experiments.txt: approximate division, -30 %
        # instructions, bought 1.5 %).  Per candidate: one guide cell (8 B), one emitter record (64 B), one normal matrix (48 B);
        # a lookup that lands in a boundary cell of the guide also reads one or two 16-byte spans.
        lt = ctx.lights_table_info()
        boundary = 1.0 - lt["interior_cells"] / max(1, lt["cells"])
        cand_bytes = W * H * (32 * (8 + 64 + 48 + boundary * 16) + 64 + 72)
        cand_sectors = W * H * (32 * (3 + boundary * 1.25) + 3)
        l2 = [json.loads(l) for l in _profile_lines("r03_l2_gather.jsonl")]
        l2_peak = [r["Gsectors_per_s"] for r in l2 if abs(r.get("table_MB", 0) - 4.0) < 1e-6]
        l2_peak = sum(l2_peak) / len(l2_peak) if l2_peak else None
        ach = cand_sectors / (init_ms * 1e-3) / 1e9
        roof["initial_candidates"] = {"bound": "l2-gather (64-byte sectors out of L2-resident tables)", "kernel": "k_initial_candidates (32 streaming-RIS candidates per pixel)",
                                      "ms": round(init_ms, 4), "algorithmic_bytes_per_launch": int(cand_bytes), "sectors_per_launch": int(cand_sectors),
                                      "guide_boundary_fraction": round(boundary, 4),
                                      "achieved": round(ach * 64, 1), "peak": round(l2_peak * 64, 1) if l2_peak else None, "unit": "GB/s",
                                      "frac": round(ach / l2_peak, 4) if l2_peak else None,
                                      "peak_source": "profiles/r03_l2_gather.jsonl: mean of the three access shapes at a 4-MB table, G sectors/s x 64 B",
                                      "frac_of_hbm_peak_nominal": round(cand_bytes / (init_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    return roof, per_frame


def roofline_nrc(ctx, renderer, stream, W, H):
    """configs[3]: the fully fused MLP is the one dense contraction of the path (north_star: "evidenced by rocprof MFMA utilisation").
    achieved = 18 432 FLOP per query (2 x (64x64 + 64x64 + 64x16): first layer, one hidden-to-hidden layer, padded output layer)
    x the frame's inference queries / the HIP-event duration of k_nrc_infer, live in this run, against the dense bf16 MFMA peak.
    The MFMA-busy counter and the TCP request rate of the same kernel come from the committed rocprofv3 --pmc pass."""
    import torch
    from gfxexp_amd import api
    os.environ["GFX_NRC_SERIAL_TRAINING"] = "1"          # one stream: kernels that overlap read longer than they are
    os.environ["GFX_SERIAL_FRAMES"] = "1"                # ... nor the next frame's G-buffer pass under this frame's inference
    serial = api.NrcRenderer(ctx, renderer.cfg)
    os.environ.pop("GFX_NRC_SERIAL_TRAINING", None)
    os.environ.pop("GFX_SERIAL_FRAMES", None)
    for _ in range(6):
        serial.render_frame(stream)
    serial.network()
    torch.cuda.synchronize()
    ctx.timing_enable(True)
    ctx.timing_collect()
    n = 10
    for _ in range(n):
        serial.render_frame(stream)
    serial.network()
    torch.cuda.synchronize()
    timings = ctx.timing_collect()
    ctx.timing_enable(False)
    stats = serial.stats()
    serial.close()
    per_frame = {k: round(ms / n, 4) for k, (ms, calls) in sorted(timings.items(), key=lambda kv: -kv[1][0])}
    infer_ms, infer_calls = timings.get("nrc_infer", (0.0, 0))
    queries = stats["numInferenceQueries"]
    flop_per_query = 2 * (64 * 64 + 64 * 64 + 64 * 16)
    launch_ms = infer_ms / max(1, infer_calls)
    # a frame infers the pixels' queries in one launch and the training tiles' suffix queries in a second, small one
    tflops = flop_per_query * queries / max(1e-9, infer_ms / n * 1e-3) / 1e12
    peak = 2500.0
    pmc_txt = None
    for name in ("r06_nrc_pmc.txt", "r05_nrc_pmc.txt", "r04_nrc_pmc.txt"):
        if os.path.exists(os.path.join(ROOT, "profiles", name)):
            pmc_txt = "profiles/" + name
            break
    roof = {"bound": "mfma (nominal: a wave of the kernel spends a third of its cycles on the level-table copies and their barriers, a fifth on the hash features: profiles/r06_nrc_infer_profile.json)",
            "kernel": "k_nrc_infer_staged (hash-grid levels staged through LDS + fused 64-wide MLP, v_mfma_f32_32x32x16_bf16)",
            "achieved": round(tflops, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(tflops / peak, 5),
            "flop_per_query": flop_per_query, "queries_per_frame": queries, "launches_per_frame": round(infer_calls / n, 2), "avg_launch_ms": round(launch_ms, 4),
            "infer_ms_per_frame": round(infer_ms / n, 4),
            "traffic": None,
            "counters": pmc_txt and (pmc_txt + ": SQ_VALU_MFMA_BUSY_CYCLES, TCP request rate and L2 hit rate of k_nrc_infer / k_nrc_train (rocprofv3 --pmc of tools/bench_nrc_frame.py)"),
            "training": {"records_per_frame": stats["numTrainingData"], "tile": list(stats["tileSize"]),
                         "fwd_bwd_ms_per_frame": per_frame.get("nrc_train_fwd_bwd"), "optimizer_ms_per_frame": per_frame.get("nrc_optimizer"), "pack_ms_per_frame": per_frame.get("nrc_pack")},
            "traversal_ms_per_frame": round(sum(v for k, v in per_frame.items() if k.startswith("trace_")), 4)}
    return roof, per_frame, stats


def mse_vs_reference(ctx, hs, renderer, cam, W, H, ref_spp, band=None, dist=None):
    """MSE / relMSE of one 1-spp ReSTIR frame against an fp64 accumulation of `ref_spp` frames of
    plain RIS/NEE (temporal + spatial reuse off), same scene and camera.
    N > 1 (`band`, `dist`): the reference estimator reads no other pixel, so every rank accumulates the reference of ITS rows with a band
    renderer that exchanges nothing -- the same per-pixel RNG streams, hence the same reference image as the one-GPU run's -- holds them
    against its rows of the timed renderer's last frame, and the sums of squared errors are added over the ranks."""
    import torch
    from gfxexp_amd import api
    b0, b1 = band if band is not None else (0, H)
    n = (b1 - b0) * W
    first = b0 * W                   # pixels are row-major: a band is one contiguous range of the HDR buffer
    test = torch.from_numpy(ctx.read_device(renderer.beauty_ptr() + first * 16, n * 16).view(np.float32).reshape(n, 4).copy())[:, :3].double()
    cfg = api.RestirRenderer.default_config(W, H, api.RENDERER_UNBIASED)
    cfg.camera = cam
    cfg.enableTemporalReuse = 0
    cfg.enableSpatialReuse = 0
    if band is not None:
        cfg.rowBegin, cfg.rowEnd = b0, b1
    ref_r = api.RestirRenderer(ctx, cfg)
    acc = torch.zeros(n * 4, dtype=torch.float64, device="cuda")
    view = _device_view(ref_r.beauty_ptr() + first * 16, n * 4)
    stream = torch.cuda.current_stream().cuda_stream
    t0 = time.perf_counter()
    for _ in range(ref_spp):
        ref_r.render_frame(stream)       # same stream as the accumulation below: in order, no host sync per frame
        acc += view
    torch.cuda.synchronize()
    seconds = time.perf_counter() - t0
    ref = (acc / ref_spp).view(n, 4)[:, :3].cpu()
    one_ref_frame = view.double().view(n, 4)[:, :3].cpu()      # the last frame of the reference estimator, for scale
    ref_r.close()
    err = (test - ref) ** 2
    err_plain = (one_ref_frame - ref) ** 2
    # sums over this rank's pixels (x 3 channels), then over the ranks: the means are over the whole frame
    sums = torch.tensor([float(err.sum()), float((err / (ref ** 2 + 1e-2)).sum()), float(err_plain.sum()), float((err_plain / (ref ** 2 + 1e-2)).sum()),
                         float(err.numel()), seconds], dtype=torch.float64)
    how = {}
    if dist is not None:
        sums = sums.cuda()
        longest = sums[5:6].clone()
        dist.all_reduce(sums, op=dist.ReduceOp.SUM)
        dist.all_reduce(longest, op=dist.ReduceOp.MAX)
        sums = sums.cpu()
        seconds = float(longest.item())
        how = {"how": "every rank accumulated the reference of its own rows (a band renderer without reuse reads no other rank's pixels) and held it against its rows "
                      "of the last timed frame; squared errors summed over the ranks.  Same per-pixel RNG streams as the one-GPU run: the same reference image"}
    count = float(sums[4])
    assert count == 3.0 * W * H, (count, W, H)
    out = {"mse": float(sums[0] / count), "rel_mse": float(sums[1] / count), "ref_spp": ref_spp,
           "mse_of_one_reference_frame": float(sums[2] / count), "rel_mse_of_one_reference_frame": float(sums[3] / count),
           "ref_estimator": "RIS/NEE 32 candidates + visibility, no reuse, fp64 accumulation", "ref_seconds": round(seconds, 1),
           "note": "the metric names a 64k-spp reference (the default); a smaller --mse-ref-spp reads higher by the reference's own noise"}
    out.update(how)
    return out


def _device_view(ptr, num_floats):
    """Wrap a raw device pointer as a torch tensor (no copy) through the CUDA array interface."""
    import torch

    class _Holder:
        pass
    h = _Holder()
    h.__cuda_array_interface__ = {"shape": (num_floats,), "typestr": "<f4", "data": (int(ptr), False), "version": 2}
    return torch.as_tensor(h, device="cuda")


def cpu_baseline(hs, cam, sample, W, H, unbiased=False, env=None):
    """The CPU restatement (oracle/, test infrastructure) timed on this host: same scene, same camera, same
    settings, steady-state frames on a reduced pixel count.  BASELINE.md section 2 names two builds of it -- the parity
    build (the checker: contraction off, x86-64-v3) and a speed build (-O3 -march=native, contraction allowed, compiled on
    this host) -- each once on one thread (the scalar CPU path) and once on every host thread (OpenMP over pixels inside
    each pass).  `value` is the faster of the two builds on one thread: the best the scalar CPU path does here."""
    from oracle import oracle as O
    from tests import util
    sw, sh = [int(x) for x in sample.lower().split("x")]
    ocam = util.copy_struct(O.GfxCamera, cam)
    ocam.aspect = float(sw) / float(sh)
    cores = int(O.lib().orc_max_threads())

    num_passes, num_nb = (1, 3) if unbiased else (2, 5)

    def run(osc, threads, frames):
        osc.set_threads(threads)
        pb = util.PixelBuffers(sw, sh)
        if env is not None:
            pb.set_env(env[0], env[1], env[2], oracle_side=True)
        s = pb.host_static_params()
        last_res, last_base = 1, 0
        times = []
        for frame in range(frames):
            kw = dict(frameIndex=frame, bufferIndex=frame % 2, resetFlowBuffer=int(frame == 0), numAccumFrames=0,
                      numSpatialNeighbors=num_nb, useUnbiasedEstimator=int(unbiased))
            if env is not None:
                kw.update(enableEnvLight=1, envLightPowerCoeff=env[3], envLightRotation=env[4])
            f = util.frame_params(O.GfxRestirFrameParams, O.GfxCamera, sw, sh, ocam, **kw)
            cur = (last_res + 1) % 2
            t0 = time.perf_counter()
            osc.restir_launch(s, f, cur, last_base, 0)
            osc.restir_launch(s, f, cur, last_base, 1 if frame == 0 else (3 if unbiased else 2))
            for i in range(num_passes):
                osc.restir_launch(s, f, cur, last_base + num_nb * i, 5 if unbiased else 4)
                cur = (cur + 1) % 2
            last_base += num_nb * num_passes
            osc.restir_launch(s, f, cur, last_base, 6)
            last_res = cur
            times.append(time.perf_counter() - t0)
        return float(np.median(times[1:]))

    def rows(library, flags):
        osc = util.feed_oracle(hs, threads=1, library=library)
        t1 = run(osc, 1, 4)
        tn = run(osc, cores, 8)
        out = {"flags": "g++ " + flags, "one_thread": {"value": round(sw * sh / t1 / 1e6, 5), "cores": 1, "seconds_per_sample_frame": round(t1, 3)},
               "all_threads": {"value": round(sw * sh / tn / 1e6, 5), "cores": cores, "seconds_per_sample_frame": round(tn, 4)},
               "bvh_build_s": round(osc.build_seconds, 2)}
        osc.close()
        return out

    parity = rows(None, O.PARITY_FLAGS)
    try:
        speed = rows(O.lib_fast(), O.FAST_FLAGS)
    except Exception as e:                       # no compiler on the host: the parity rows stand alone
        speed = {"error": repr(e)[:200]}
    best = speed if "one_thread" in speed and speed["one_thread"]["value"] >= parity["one_thread"]["value"] else parity
    return {"value": best["one_thread"]["value"], "unit": "Mpaths/s", "cores": 1, "kind": "port",
            "build": "speed" if best is speed else "parity",
            "sample": f"{sw}x{sh} pixels ({sw * sh / (W * H):.4f} of the frame), steady-state frames (median of 3 on one thread, of 7 on all), "
                      "same scene/camera/settings, scalar C++ restatement (oracle/), SAH BVH8 build untimed",
            "seconds_per_sample_frame": best["one_thread"]["seconds_per_sample_frame"],
            "all_cores": {"value": best["all_threads"]["value"], "unit": "Mpaths/s", "cores": cores,
                          "seconds_per_sample_frame": best["all_threads"]["seconds_per_sample_frame"], "how": "OpenMP over pixels inside every pass"},
            "builds": {"parity": parity, "speed": speed}}


def cpu_baseline_path_tracer(hs, cam, W, H):
    """configs[1] on the host cores: the oracle's restatement of the baseline path tracer (path_tracing/gpu_kernels/
    optix_pathtracing_kernels.cu:74-341 over the restated SAH BVH8), the whole 512x512 frame, one thread and all threads."""
    from oracle import oracle as O
    from tests import util
    ocam = util.copy_struct(O.GfxCamera, cam)
    cores = int(O.lib().orc_max_threads())

    def run(osc, threads, frames):
        osc.set_threads(threads)
        pb = util.PixelBuffers(W, H)
        s = pb.host_static_params()
        times = []
        for frame in range(frames):
            kw = dict(frameIndex=frame, bufferIndex=frame % 2, resetFlowBuffer=int(frame == 0), numAccumFrames=frame, enableJittering=1)
            f = util.frame_params(O.GfxRestirFrameParams, O.GfxCamera, W, H, ocam, **kw)
            t0 = time.perf_counter()
            osc.pt_launch(s, f, 0, 5)          # GFX_PT_SETUP_GBUFFERS
            osc.pt_launch(s, f, 1, 5)          # GFX_PT_PATH_TRACE_BASELINE
            times.append(time.perf_counter() - t0)
        return float(np.median(times[1:])) if len(times) > 1 else times[0]

    out = {}
    for label, library, flags in (("parity", None, O.PARITY_FLAGS), ("speed", "fast", O.FAST_FLAGS)):
        try:
            osc = util.feed_oracle(hs, threads=1, library=O.lib_fast() if library else None)
        except Exception as e:
            out[label] = {"error": repr(e)[:200]}
            continue
        t1, tn = run(osc, 1, 3), run(osc, cores, 6)
        out[label] = {"flags": "g++ " + flags, "one_thread": {"value": round(W * H / t1 / 1e6, 5), "cores": 1, "seconds_per_frame": round(t1, 3)},
                      "all_threads": {"value": round(W * H / tn / 1e6, 5), "cores": cores, "seconds_per_frame": round(tn, 4)}}
        osc.close()
    best = max((v for v in out.values() if "one_thread" in v), key=lambda v: v["one_thread"]["value"])
    return {"value": best["one_thread"]["value"], "unit": "Mpaths/s", "cores": 1, "kind": "port",
            "sample": f"the whole {W}x{H} frame, max path length 5, scalar C++ restatement (oracle/), median of 2 frames on one thread, of 5 on all; SAH BVH8 build untimed",
            "all_cores": {"value": best["all_threads"]["value"], "unit": "Mpaths/s", "cores": cores, "how": "OpenMP over pixels"},
            "builds": out}


if __name__ == "__main__":
    main()
