#!/bin/bash
set -u
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r02w
mkdir -p $OUT
cd /tmp
B="python $ROOT/bench.py --steps 6 --warmup 2 --mse-ref-spp 0 --cpu-sample 0 --no-roofline --plain"
for c in 1 0; do
  export GFX_INIT_COOP=$c
  ( timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS --output-format csv -d $OUT/pmc_$c -- $B > /dev/null 2> $OUT/pmc_$c.err )
  python $ROOT/profiles/summarize_pmc.py $OUT/pmc_$c/*/*counter_collection.csv 2>&1 | grep -A9 "k_initial_candidates" | head -12
  rm -rf $OUT/pmc_$c
  ( timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc2_$c -- $B > /dev/null 2> $OUT/pmc2_$c.err )
  python $ROOT/profiles/summarize_pmc.py $OUT/pmc2_$c/*/*counter_collection.csv 2>&1 | grep -A7 "k_initial_candidates" | head -9
  rm -rf $OUT/pmc2_$c
done
