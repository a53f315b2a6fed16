#!/bin/bash
# GPU session r02c: software-pipelined candidate loop (4 waves/SIMD with spills vs 3 / 2 waves), PMC passes
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r02c
mkdir -p $OUT
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 ) > $OUT/pytest.log
( timeout 600 python bench.py --steps 20 --warmup 5 --mse-ref-spp 0 --cpu-sample 0 > $OUT/bench.json 2> $OUT/bench.err )
for v in p3 p3t p4t p2; do
  ( GFX_LIB=$PWD/gfxexp_amd/variants/libgfxexp_$v.so timeout 300 python bench.py --steps 20 --warmup 5 --mse-ref-spp 0 --cpu-sample 0 > $OUT/bench_$v.json 2> $OUT/bench_$v.err )
done
B="python bench.py --steps 6 --warmup 2 --mse-ref-spp 0 --cpu-sample 0 --no-roofline"
for lib in default p3; do
  if [ $lib != default ]; then export GFX_LIB=$PWD/gfxexp_amd/variants/libgfxexp_$lib.so; fi
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --output-format csv -d $OUT/pmc_sq_$lib -- $B > /dev/null 2>&1
  rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT --output-format csv -d $OUT/pmc_sq2_$lib -- $B > /dev/null 2>&1
  rocprofv3 --pmc TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_tc_$lib -- $B > /dev/null 2>&1
  rocprofv3 --pmc TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_LATENCY_sum --output-format csv -d $OUT/pmc_ta_$lib -- $B > /dev/null 2>&1
  for d in pmc_sq_$lib pmc_sq2_$lib pmc_tc_$lib pmc_ta_$lib; do
    python profiles/summarize_pmc.py $OUT/$d/*/*counter_collection.csv 2>&1 | grep -A12 "k_initial_candidates\|k_trace" > $OUT/$d.txt
    rm -rf $OUT/$d
  done
done
unset GFX_LIB
cat $OUT/pytest.log
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r02c/bench*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        k = d.get("kernels_ms_per_frame", {})
        print(f.split("/")[-1], d["value"], d["ms_per_step"], {n: k.get(n) for n in ("initial_candidates", "trace_any", "trace_closest")})
    except Exception as e:
        print(f, "ERR", e)
PY
