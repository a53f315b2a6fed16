#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r02i
mkdir -p $OUT
( timeout 600 python tools/band_kernels.py > $OUT/band_kernels.jsonl 2> $OUT/band_kernels.err )
cat $OUT/band_kernels.jsonl; tail -3 $OUT/band_kernels.err
cd /tmp
( timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof_band -o band -- python $GRAFT_REPO_ROOT/tools/band_kernels.py > /dev/null 2> $GRAFT_REPO_ROOT/$OUT/prof.err )
cd $GRAFT_REPO_ROOT
ls -R $OUT/prof_band | head; 
f=$(find $OUT/prof_band -name "*kernel_stats.csv" | head -1); head -40 "$f"
