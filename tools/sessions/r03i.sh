#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r03i
mkdir -p $OUT
for v in iw3 iw2; do
  export GFX_LIB=$PWD/gfxexp_amd/variants/libgfxexp_$v.so
  for w in plain textured; do
    flag=""; if [ $w = plain ]; then flag="--plain"; fi
    ( timeout 300 python bench.py --steps 20 --warmup 5 --mse-ref-spp 0 --cpu-sample 0 $flag > $OUT/bench_${w}_$v.json 2> $OUT/bench_${w}_$v.err )
    python - <<PY
import json
try:
    d = json.loads(open("$OUT/bench_${w}_$v.json").read().strip().splitlines()[-1]); k = d["kernels_ms_per_frame"]
    print("$v $w:", d["value"], d["ms_per_step"], "initial", k["initial_candidates"])
except Exception as e:
    print("$v $w ERR", e, open("$OUT/bench_${w}_$v.err").read()[-600:])
PY
  done
done
