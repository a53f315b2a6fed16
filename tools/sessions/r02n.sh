#!/bin/bash
# GPU session r02n: end-of-round profile of bench.py -- kernel trace + the PMC passes (separate runs, no trace domains with --pmc)
set -u
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r02n
mkdir -p $OUT
cd /tmp
B="python $ROOT/bench.py --steps 6 --warmup 2 --mse-ref-spp 0 --cpu-sample 0"
( timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/stats -o bench -- $B > $OUT/bench_under_rocprof.json 2> $OUT/stats.err )
( timeout 600 $B > $OUT/bench_plain_run.json 2> /dev/null )
pmc() { name=$1; shift; ( timeout 600 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- $B --no-roofline > /dev/null 2> $OUT/$name.err ); python $ROOT/profiles/summarize_pmc.py $OUT/$name/*/*counter_collection.csv > $OUT/$name.txt 2>&1; rm -rf $OUT/$name; }
pmc pmc_sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_THREAD_CYCLES_VALU
pmc pmc_sq2 SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_INSTS_SMEM
pmc pmc_tc TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum
pmc pmc_ta TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum
pmc pmc_fetch FETCH_SIZE
pmc pmc_write WRITE_SIZE
cd $ROOT
db=$(find $OUT/stats -name "*.db" | head -1)
if [ -n "$db" ]; then python profiles/summarize_rocpd.py $db > $OUT/kernel_stats.txt; else find $OUT/stats | head; fi
head -30 $OUT/kernel_stats.txt
for f in pmc_fetch pmc_write; do echo == $f; grep -A2 "k_trace\|k_initial" $OUT/$f.txt | head -20; done
rm -rf $OUT/stats
