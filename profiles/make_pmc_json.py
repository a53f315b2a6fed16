"""profiles/make_pmc_json.py DIR [DIR2 label2 ...] > profiles/r03_pmc.json
   profiles/make_pmc_json.py --stamp FILE > profiles/FILE      (in the repository: adds `git_head` after checking that the file's
                                                                 `sources_sha16` is the hash of the checked-out kernel sources)

Condenses the per-kernel means of the rocprofv3 --pmc passes of tools/profile_round.sh (DIR/sq.txt, sq2.txt, tc.txt, ea.txt,
wr.txt: output of profiles/summarize_pmc.py) into the figures bench.py quotes in its `roofline` object:

  valu_busy        SQ_ACTIVE_INST_VALU / (8 x SQ_BUSY_CYCLES): quad-cycles a SIMD spent issuing VALU over the cycles the 1024
                   SIMDs were available (SQ_BUSY_CYCLES sums the 32 shader engines; 1024 SIMDs / 32 / 4 = 8).  Clock independent.
  lane_fraction    SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU): active lanes per issued VALU instruction
  valu_insts       SQ_INSTS_VALU per launch (wave-level instructions)
  l2_hit           TCC_HIT / (TCC_HIT + TCC_MISS)
  tcp_accesses_per_cu_clk   TCP_TOTAL_ACCESSES / 256 CUs / (SQ_BUSY_CYCLES / 32): cache-line requests of the vector-memory instructions
                   per CU and clock; a gather of 64 different lines is 64 requests, and the path takes about one per clock
  hbm_bytes        2 x FETCH_SIZE KiB (gfx950 tallies 128-B read requests at 64 B: MI355X_MICROARCH.md) + WRITE_SIZE KiB
"""
import json
import re
import sys

KERNELS = {"k_trace_any": "k_trace<true, false>", "k_trace_closest": "k_trace<false, false>",
           "k_initial_candidates": "k_initial_candidates<", "k_initial_candidates_pooled": "k_initial_candidates_pooled<", "k_spatial": "k_spatial<false>", "k_temporal": "k_temporal<1>",
           "k_shade_prepare": "k_shade_prepare(", "k_spatial_shade_prepare": "k_spatial_shade_prepare", "k_gbuffer_resolve": "k_gbuffer_resolve", "k_gbuffer_fused": "k_gbuffer_fused",
           "k_pt_fused": "k_pt_fused<", "k_spatial_unbiased": "k_spatial<true>", "k_spatial_mis_finish": "k_spatial_mis_finish", "k_temporal_unbiased": "k_temporal<2>",
           "k_initial_fused": "k_initial_fused<", "k_shading_fused": "k_shading_fused<"}


def sources_sha16():
    """bench.py sources_sha16: the hash of gfxexp_amd/csrc the counters belong to (the GPU box runs a snapshot of the tree without .git)."""
    import hashlib
    import os
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gfxexp_amd", "csrc")
    h = hashlib.sha256()
    for fn in sorted(os.listdir(root)):          # the kernels and the headers they include; not capi.cpp / scene.cpp / host/ (host code)
        if fn.endswith((".hip", ".h")):
            h.update(fn.encode())
            with open(os.path.join(root, fn), "rb") as f:
                h.update(f.read())
    return h.hexdigest()[:16]


def parse(path):
    out, cur = {}, None
    try:
        lines = open(path).read().splitlines()
    except OSError:
        return out
    for ln in lines:
        if not ln.startswith(" "):
            cur = ln.strip()
            out[cur] = {}
        else:
            m = re.match(r"\s+(\S+)\s+launches\s+(\d+)\s+mean\s+([\d.]+)", ln)
            if m and cur is not None:
                out[cur][m.group(1)] = float(m.group(3))
    return out


def condense(d):
    tables = {}
    for f in ("sq", "sq2", "tc", "ea", "wr"):
        for k, v in parse(f"{d}/{f}.txt").items():
            tables.setdefault(k, {}).update(v)
    res = {}
    for short, pat in KERNELS.items():
        c = next((v for k, v in tables.items() if pat in k), None)
        if not c:
            continue
        e = {}
        if c.get("SQ_BUSY_CYCLES"):
            e["valu_busy"] = round(c.get("SQ_ACTIVE_INST_VALU", 0) / (8 * c["SQ_BUSY_CYCLES"]), 4)
        if c.get("SQ_ACTIVE_INST_VALU"):
            e["lane_fraction"] = round(c.get("SQ_THREAD_CYCLES_VALU", 0) / (64 * c["SQ_ACTIVE_INST_VALU"]), 4)
        e["valu_insts"] = int(c.get("SQ_INSTS_VALU", 0))
        e["waves"] = int(c.get("SQ_WAVES", 0))
        if c.get("TCC_HIT_sum") is not None and (c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0)) > 0:
            e["l2_hit"] = round(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 4)
            e["l2_requests"] = int(c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
        if c.get("TCP_TOTAL_ACCESSES_sum") and c.get("SQ_BUSY_CYCLES"):
            # vector-memory cache-line requests per CU and clock (SQ_BUSY_CYCLES sums 32 shader engines; 256 CUs): the
            # texture-addresser / L1 path takes about one per clock
            e["tcp_accesses_per_cu_clk"] = round(c["TCP_TOTAL_ACCESSES_sum"] / 256.0 / (c["SQ_BUSY_CYCLES"] / 32.0), 3)
        if c.get("FETCH_SIZE") is not None:
            e["hbm_bytes"] = int((2 * c["FETCH_SIZE"] + c.get("WRITE_SIZE", 0)) * 1024)
        res[short] = e
    return res


def stamp(path):
    import subprocess
    d = json.load(open(path))
    if d.get("sources_sha16") != sources_sha16():
        sys.exit("make_pmc_json --stamp: %s was measured on other kernel sources (%s) than the checked-out ones (%s)" % (path, d.get("sources_sha16"), sources_sha16()))
    d["git_head"] = subprocess.run(["git", "rev-parse", "HEAD"], capture_output=True, text=True, check=True).stdout.strip()
    dirty = subprocess.run(["git", "status", "--porcelain", "--", "gfxexp_amd/csrc"], capture_output=True, text=True).stdout.strip()
    if dirty:
        sys.exit("make_pmc_json --stamp: uncommitted changes under gfxexp_amd/csrc -- commit first, HEAD would not name these sources")
    print(json.dumps(d, indent=1))


def main(argv):
    if argv and argv[0] == "--stamp":
        return stamp(argv[1])
    command = "python bench.py --steps 6 --warmup 2 --mse-ref-spp 0 --cpu-sample 0 --no-roofline"
    if argv and argv[0] == "--command":
        command, argv = argv[1], argv[2:]
    out = {"sources_sha16": sources_sha16(), "git_head": None, "command": command, "source": "rocprofv3 --pmc passes of tools/profile_round.sh (step pmc / pmc0) over `python bench.py --steps 6 --warmup 2 --mse-ref-spp 0 "
                     "--cpu-sample 0 --no-roofline`, MI355X; formulas in profiles/make_pmc_json.py", "kernels": condense(argv[0])}
    t = out["kernels"]
    if "k_trace_any" in t and "k_trace_closest" in t and "hbm_bytes" in t["k_trace_any"]:
        out["hbm_bytes_per_launch"] = int((2 * t["k_trace_any"]["hbm_bytes"] + t["k_trace_closest"]["hbm_bytes"]) / 3)
    elif "k_trace_any" in t and "hbm_bytes" in t["k_trace_any"]:     # round 4 on: the closest-hit traversal of the frame runs inside k_gbuffer_fused
        out["hbm_bytes_per_launch"] = int(t["k_trace_any"]["hbm_bytes"])
    for d, label in zip(argv[1::2], argv[2::2]):
        out[label] = condense(d)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1:])
