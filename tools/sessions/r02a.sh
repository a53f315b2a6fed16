#!/bin/bash
# GPU session r02a: parity after the emitter interval table, first bench line, tree-quality baseline, kernel stats.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r02a
mkdir -p $OUT
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $OUT/pytest.log
( timeout 600 python bench.py --steps 20 --warmup 5 --mse-ref-spp 2048 > $OUT/bench.json 2> $OUT/bench.err )
( GFX_LIGHT_TABLE=0 timeout 600 python bench.py --steps 20 --warmup 5 --mse-ref-spp 0 --cpu-sample 0 > $OUT/bench_notable.json 2> $OUT/bench_notable.err )
( timeout 600 python tools/bvh_quality.py bench $OUT/bvh_quality.json > $OUT/bvh_quality.log 2>&1 )
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --mse-ref-spp 0 --cpu-sample 0 --no-roofline > $GRAFT_REPO_ROOT/$OUT/prof_bench.log 2>&1 )
find $OUT/prof -name "*kernel_stats*" | head -3 | while read f; do head -40 "$f" > $OUT/kernel_stats.csv; done
find $OUT/prof -name "*.db" -delete 2>/dev/null
find $OUT/prof -size +2M -delete 2>/dev/null
cat $OUT/pytest.log
head -c 3000 $OUT/bench.json
