#!/bin/bash
# GPU session r02q: phased traversal, 2 and 3 postponed leaf sets per lane
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r02q
mkdir -p $OUT
for v in post2 post3; do
export GFX_LIB=$PWD/gfxexp_amd/variants/libgfxexp_$v.so
( GFX_TRACE_TRI_SHARE=24 timeout 900 python -m pytest tests/test_gpu_trace.py -m gpu -q -x 2>&1 | tail -3 ) > $OUT/pytest_$v.log
cat $OUT/pytest_$v.log
for t in 16 24 32 40; do
  ( GFX_TRACE_TRI_SHARE=$t timeout 300 python bench.py --steps 20 --warmup 5 --mse-ref-spp 0 --cpu-sample 0 > $OUT/bench_${v}_t$t.json 2> $OUT/bench_${v}_t$t.err )
  python - <<PY
import json
try:
    d = json.loads(open("$OUT/bench_${v}_t$t.json").read().strip().splitlines()[-1]); k = d["kernels_ms_per_frame"]; r = d["roofline"]
    print("$v triShare $t:", d["value"], d["ms_per_step"], "any", k["trace_any"], "closest", k["trace_closest"], "nodes/ray", r["node_visits_per_ray"]["primary"], r["node_visits_per_ray"]["shadow"], "sched", r["scheduling"])
except Exception as e:
    print("$v triShare $t ERR", e, open("$OUT/bench_${v}_t$t.err").read()[-500:])
PY
done
done
