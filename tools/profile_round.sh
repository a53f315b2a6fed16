#!/bin/bash
# tools/profile_round.sh -- every measurement behind profiles/<round>_* (ROUND=r04 by default; r03 in the step list below reads "the round") as named steps, runnable from a clean checkout on a
# GPU box (through gpurun: `gpurun --timeout 1500 -- 'bash tools/profile_round.sh tests ab pmc'`).  Raw output goes to
# gpurun_out/r03/<step>/ (scratch); the summaries quoted in DESIGN.md are copied from there into profiles/ by hand-picked
# names (listed next to each step).  Replaces the one-off tools/sessions/r0*.sh scripts of rounds 1-2.
#
#   tests        python -m pytest tests -m gpu -x -q                                   -> gpurun_out/r03/tests.log
#   bench        default bench.py line (64k-spp MSE + CPU rows, ~3 min)                -> profiles/r03_bench_default.json
#   benchq       bench.py without the MSE / CPU legs, default + --plain + --cluttered  -> profiles/r03_bench_quick.jsonl
#   ab           tools/pixel_map_ab.py: scan-line / tiled / XCD-supertile pixel maps   -> profiles/r03_pixel_map_ab.jsonl
#   absweep      the same over supertile shapes                                        -> profiles/r03_pixel_map_supers.jsonl
#   stats        rocprofv3 --kernel-trace --stats of the short bench command           -> profiles/r03_kernel_stats.txt
#   pmc          PMC passes (SQ x2, TCP/TCC, FETCH, WRITE) of the same command, default pixel map
#   pmc0         the same with GFX_PIXEL_MAP=0 (rounds 1-2 mapping) for the before/after table -> profiles/r03_pixel_map_pmc.txt
#   nrcpmc       MFMA / VALU counters of k_nrc_infer, k_nrc_train (tools/bench_nrc.py)  -> profiles/r03_nrc_pmc.txt
#   nrc          tools/bench_nrc.py (network alone) + tools/bench_nrc_frame.py (NRC frame, training overlapped / serial)
#                                                                                       -> profiles/r03_nrc_frame.jsonl
#   pmcjson      profiles/make_pmc_json.py over the pmc (+ pmc0) outputs                -> profiles/r03_pmc.json
#   laneprof     lane-utilisation profile of k_initial_candidates (builds the GFX_LANE_PROFILE variant first, on this
#                host: python gfxexp_amd/build.py --variant laneprof GFX_LANE_PROFILE)    -> profiles/r03_initial_candidates.txt
#   renderers    tools/bench_renderers.py + tools/bench_config4.py                      -> profiles/r03_renderers.jsonl
#   bands        tools/bench_band.py (compute-only bound of N row bands)                -> profiles/r03_band_compute_bound.json
#   timeline     tools/band_timeline.py (kernel start / end times of one band frame)    -> profiles/r06_band_timeline.json
#   l2gather     tools/microbench/l2_gather.hip: scattered 64-byte sectors per second out of L2 / Infinity Cache / HBM by table size
#                                                                                       -> profiles/r03_l2_gather.jsonl
#   valurate     tools/microbench/valu_rate.hip: wave64 VALU issue rate per SIMD by instruction kind and resident waves
#   whatif       k_trace with extra VALU work / more resident waves (experiment variants)  -> profiles/r03_experiments.txt
#   hbm          tools/hbm_stream.py (streaming-copy ceiling of this box)               -> profiles/r03_hbm_stream.json
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
ROUND=${ROUND:-r06}
OUT=gpurun_out/$ROUND
mkdir -p $OUT
B="python bench.py --steps 6 --warmup 2 --mse-ref-spp 0 --cpu-sample 0 --other-configs 0"

pmc_passes() {   # $1 = output tag, rest = environment assignments; $BFLAGS = extra bench.py flags (--config N, --animate)
  local tag=$1; shift
  local B="$B ${BFLAGS:-}"
  local d=$OUT/$tag
  mkdir -p $d
  env "$@" timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --output-format csv -d $d/sq -- $B --no-roofline > /dev/null 2>&1
  env "$@" timeout 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_THREAD_CYCLES_VALU SQ_WAVES --output-format csv -d $d/sq2 -- $B --no-roofline > /dev/null 2>&1
  env "$@" timeout 300 rocprofv3 --pmc TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $d/tc -- $B --no-roofline > /dev/null 2>&1
  env "$@" timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $d/ea -- $B --no-roofline > /dev/null 2>&1
  env "$@" timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $d/wr -- $B --no-roofline > /dev/null 2>&1
  for p in sq sq2 tc ea wr; do
    python profiles/summarize_pmc.py $d/$p/*/*counter_collection.csv > $d/$p.txt 2>&1
  done
  rm -rf $d/sq $d/sq2 $d/tc $d/ea $d/wr
}

for step in "$@"; do
  echo "==== $step"
  case $step in
    tests)     timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $OUT/tests.log ;;
    bench)     timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; cat $OUT/bench_default.json | cut -c1-600 ;;
    benchq)    : > $OUT/bench_quick.jsonl
               for f in "" "--plain" "--cluttered"; do timeout 300 python bench.py --steps 30 --warmup 5 --mse-ref-spp 0 --cpu-sample 0 --other-configs 0 $f >> $OUT/bench_quick.jsonl 2>> $OUT/bench_quick.err; done
               cut -c1-400 $OUT/bench_quick.jsonl ;;
    ab)        timeout 600 python tools/pixel_map_ab.py > $OUT/pixel_map_ab.jsonl 2> $OUT/pixel_map_ab.err
               timeout 600 python tools/pixel_map_ab.py --plain >> $OUT/pixel_map_ab.jsonl 2>> $OUT/pixel_map_ab.err
               cat $OUT/pixel_map_ab.jsonl; tail -3 $OUT/pixel_map_ab.err ;;
    absweep)   timeout 900 python tools/pixel_map_ab.py --modes 2 --supers 0x0,1x1,2x1,2x2,3x2,3x3,4x3,4x4 > $OUT/pixel_map_supers.jsonl 2> $OUT/pixel_map_supers.err
               cat $OUT/pixel_map_supers.jsonl; tail -3 $OUT/pixel_map_supers.err ;;
    stats)     rm -rf $OUT/stats; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $B > /dev/null 2>&1
               cat $OUT/stats/*/*kernel_stats.csv | cut -c1-200 | head -40 | tee $OUT/kernel_stats.csv ;;
    pmc)       pmc_passes pmc_default GFX_NOOP=1; head -60 $OUT/pmc_default/sq.txt ;;
    pmc0)      pmc_passes pmc_map0 GFX_PIXEL_MAP=0 ;;
    pmcc1)     BFLAGS="--config 1" pmc_passes pmc_config1 GFX_NOOP=1      # round 5: counter passes of the other BASELINE configurations -> profiles/r05_pmc_config1.json ...
               python profiles/make_pmc_json.py --command "$B --config 1 --no-roofline" $OUT/pmc_config1 > $OUT/${ROUND}_pmc_config1.json; head -c 400 $OUT/${ROUND}_pmc_config1.json ;;
    pmcc4)     BFLAGS="--config 4" pmc_passes pmc_config4 GFX_NOOP=1
               python profiles/make_pmc_json.py --command "$B --config 4 --no-roofline" $OUT/pmc_config4 > $OUT/${ROUND}_pmc_config4.json; head -c 400 $OUT/${ROUND}_pmc_config4.json ;;
    pmcc4sk)   BFLAGS="--config 4" pmc_passes pmc_config4_sketch GFX_ENV_ROW_SKETCH=1     # round 6: the same with the rows' inverse-CDF sketches (no guide read in verified cells)
               python profiles/make_pmc_json.py --command "GFX_ENV_ROW_SKETCH=1 $B --config 4 --no-roofline" $OUT/pmc_config4_sketch > $OUT/${ROUND}_pmc_config4_sketch.json; head -c 400 $OUT/${ROUND}_pmc_config4_sketch.json ;;
    ptregen)   timeout 600 python tools/pt_regen_diag.py > $OUT/pt_regen.jsonl 2> $OUT/pt_regen.err; cat $OUT/pt_regen.jsonl ;;
    bandhost)  timeout 300 python tools/band_host_overhead.py host > $OUT/band_host.json 2> $OUT/band_host.err; cat $OUT/band_host.json
               : > $OUT/band_latency.jsonl
               for s in round5 gb_lane lanes_noseam lanes recompute; do timeout 300 python tools/band_host_overhead.py latency --schedule $s >> $OUT/band_latency.jsonl 2>> $OUT/band_host.err; done
               for s in round5 recompute; do timeout 300 python tools/band_host_overhead.py latency --schedule $s --config4 >> $OUT/band_latency.jsonl 2>> $OUT/band_host.err; done
               cat $OUT/band_latency.jsonl ;;
    pmca)      BFLAGS="--animate" pmc_passes pmc_animate GFX_NOOP=1
               python profiles/make_pmc_json.py --command "$B --animate --no-roofline" $OUT/pmc_animate > $OUT/${ROUND}_pmc_animate.json; head -c 400 $OUT/${ROUND}_pmc_animate.json ;;
    pmc1)      pmc_passes pmc_map1 GFX_PIXEL_MAP=1 ;;
    nrcpmc)    N="python tools/bench_nrc.py --steps 5"; mkdir -p $OUT/nrc_pmc
               timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/nrc_pmc/a -- $N > /dev/null 2>&1
               timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU SQ_WAVES --output-format csv -d $OUT/nrc_pmc/b -- $N > /dev/null 2>&1
               timeout 300 rocprofv3 --pmc TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/nrc_pmc/c -- $N > /dev/null 2>&1
               timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/nrc_pmc/stats -- $N > /dev/null 2>&1
               for p in a b c; do python profiles/summarize_pmc.py $OUT/nrc_pmc/$p/*/*counter_collection.csv > $OUT/nrc_pmc/$p.txt 2>&1; done
               cat $OUT/nrc_pmc/stats/*/*kernel_stats.csv | cut -c1-160 | head -12 > $OUT/nrc_pmc/kernel_stats.csv
               rm -rf $OUT/nrc_pmc/a $OUT/nrc_pmc/b $OUT/nrc_pmc/c $OUT/nrc_pmc/stats
               grep -A9 "k_nrc_infer\|k_nrc_train" $OUT/nrc_pmc/a.txt | head -40 ;;
    nrc)       timeout 300 python tools/bench_nrc.py --steps 20 2> $OUT/nrc_net.err | tail -1 > $OUT/nrc_net.json; cut -c1-700 $OUT/nrc_net.json
               timeout 600 python tools/bench_nrc_frame.py > $OUT/nrc_frame.jsonl 2> $OUT/nrc_frame.err; cat $OUT/nrc_frame.jsonl; tail -3 $OUT/nrc_frame.err ;;
    pmcjson)   if [ -d $OUT/pmc_map0 ]; then python profiles/make_pmc_json.py $OUT/pmc_default $OUT/pmc_map0 pixel_map_0_scan_lines > $OUT/${ROUND}_pmc.json
               else python profiles/make_pmc_json.py $OUT/pmc_default > $OUT/${ROUND}_pmc.json; fi; head -c 600 $OUT/${ROUND}_pmc.json ;;
    laneprof)  GFX_LIB=$PWD/gfxexp_amd/variants/libgfxexp_laneprof.so timeout 300 python tools/lane_profile.py > $OUT/lane_profile.json 2> $OUT/lane_profile.err; cat $OUT/lane_profile.json ;;
    renderers) timeout 900 python tools/bench_renderers.py > $OUT/renderers.jsonl 2> $OUT/renderers.err
               timeout 300 python tools/bench_config4.py >> $OUT/renderers.jsonl 2>> $OUT/renderers.err; cat $OUT/renderers.jsonl ;;
    bands)     timeout 900 python tools/bench_band.py > $OUT/band_compute_bound.json 2> $OUT/band.err; cat $OUT/band_compute_bound.json; tail -3 $OUT/band.err ;;
    bands4)    timeout 900 python tools/bench_band.py --config4 --modes strips > $OUT/band_compute_bound_config4.json 2>> $OUT/band.err; cat $OUT/band_compute_bound_config4.json ;;
    timeline)  rm -rf $OUT/tl; timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/tl/c2 -- python tools/band_timeline.py run > /dev/null 2>&1
               python tools/band_timeline.py read $OUT/tl/c2 > $OUT/band_timeline_c2.json
               timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/tl/c4 -- python tools/band_timeline.py run --config4 > /dev/null 2>&1
               python tools/band_timeline.py read $OUT/tl/c4 > $OUT/band_timeline_c4.json; rm -rf $OUT/tl; cat $OUT/band_timeline_c2.json $OUT/band_timeline_c4.json ;;
    nrcprof)   GFX_LIB=$PWD/gfxexp_amd/variants/libgfxexp_laneprof.so timeout 300 python tools/nrc_infer_profile.py > $OUT/nrc_infer_profile.json 2> $OUT/nrc_prof.err; cat $OUT/nrc_infer_profile.json ;;
    l2gather)  hipcc --offload-arch=gfx950 -O3 -w -o /tmp/l2_gather tools/microbench/l2_gather.hip && timeout 120 /tmp/l2_gather | tee $OUT/l2_gather.jsonl ;;
    valurate)  hipcc --offload-arch=gfx950 -O3 -w -o /tmp/valu_rate tools/microbench/valu_rate.hip && timeout 120 /tmp/valu_rate | tee $OUT/valu_rate.jsonl ;;
    whatif)    # (round 3 only: the GFX_WHATIF_* hooks these variants switched were removed from the kernels in round 4)
               # sensitivity of k_trace to extra VALU work and to more resident waves (variants built beforehand on this host:
               #   python gfxexp_amd/build.py --variant valu64 GFX_WHATIF_VALU=64 ; ... valu128 GFX_WHATIF_VALU=128 ;
               #   ... occ5 GFX_TRACE_LDS_STACK=6 GFX_TRACE_MIN_WAVES=5 ; ... occ6 GFX_TRACE_LDS_STACK=4 GFX_TRACE_MIN_WAVES=6 ;
               #   ... fastdiv -fno-hip-fp32-correctly-rounded-divide-sqrt   (approximate / and sqrtf: timing only, results differ))
               : > $OUT/whatif.jsonl
               Q="python bench.py --steps 20 --warmup 5 --mse-ref-spp 0 --cpu-sample 0 --other-configs 0"
               run_v() { tag=$1; shift; echo "{\"variant\": \"$tag\"}" >> $OUT/whatif.jsonl; env "$@" timeout 300 $Q >> $OUT/whatif.jsonl 2>> $OUT/whatif.err; }
               V=$PWD/gfxexp_amd/variants
               run_v shipped GFX_NOOP=1
               [ -f $V/libgfxexp_valu64.so ] && run_v valu64 GFX_LIB=$V/libgfxexp_valu64.so
               [ -f $V/libgfxexp_valu128.so ] && run_v valu128 GFX_LIB=$V/libgfxexp_valu128.so
               [ -f $V/libgfxexp_occ5.so ] && run_v occ5_stack6_4blocks GFX_LIB=$V/libgfxexp_occ5.so GFX_TRACE_BLOCKS_PER_CU=4
               [ -f $V/libgfxexp_occ5.so ] && run_v occ5_stack6_5blocks GFX_LIB=$V/libgfxexp_occ5.so GFX_TRACE_BLOCKS_PER_CU=5
               [ -f $V/libgfxexp_occ6.so ] && run_v occ6_stack4_4blocks GFX_LIB=$V/libgfxexp_occ6.so GFX_TRACE_BLOCKS_PER_CU=4
               [ -f $V/libgfxexp_occ6.so ] && run_v occ6_stack4_6blocks GFX_LIB=$V/libgfxexp_occ6.so GFX_TRACE_BLOCKS_PER_CU=6
               [ -f $V/libgfxexp_fastdiv.so ] && run_v fastdiv GFX_LIB=$V/libgfxexp_fastdiv.so
               [ -f $V/libgfxexp_sector.so ] && run_v trace_one_more_sector_per_node GFX_LIB=$V/libgfxexp_sector.so
               for n in 1 2 3 4 8 16 32 64 128 192; do [ -f $V/libgfxexp_init$n.so ] && run_v init_whatif_$n GFX_LIB=$V/libgfxexp_init$n.so; done   # GFX_WHATIF_INIT bit 0: no table search, coalesced records; bit 1: no BSDF evaluation; bit 2: real search, coalesced records; bit 3: no search, scattered records; bit 4: one guide load picks the record; bit 5: no material-texture reads in make_shading_point
               python - <<'PY'
import json
for l in open("gpurun_out/r03/whatif.jsonl"):
    d = json.loads(l)
    if "variant" in d: print(d["variant"], end=": ")
    else: print(d["ms_per_step"], d["kernels_ms_per_frame"])
PY
               ;;
    # (round 5's deferab step -- the candidate loop with the BSDF evaluation deferred, GFX_DEFER_CANDIDATES / _PARK / _BLOCKED sweeps -- went with the
    #  experiment: commit 5fd9a5c has it; profiles/r05_experiments.txt 1)
    hbm)       timeout 300 python tools/hbm_stream.py > $OUT/hbm_stream.json 2> $OUT/hbm.err; cat $OUT/hbm_stream.json ;;
    *)         echo "unknown step $step" ;;
  esac
done
