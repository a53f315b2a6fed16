#!/bin/bash
# GPU session r02p: phased node / triangle iterations in k_trace (GFX_TRACE_TRI_SHARE sweep)
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r02p
mkdir -p $OUT
( GFX_TRACE_TRI_SHARE=16 timeout 900 python -m pytest tests/test_gpu_trace.py tests/test_gpu_restir.py -m gpu -q -x 2>&1 | tail -5 ) > $OUT/pytest16.log
cat $OUT/pytest16.log
for t in 0 1 8 16 24 32 40 48 64; do
  ( GFX_TRACE_TRI_SHARE=$t timeout 300 python bench.py --steps 20 --warmup 5 --mse-ref-spp 0 --cpu-sample 0 > $OUT/bench_t$t.json 2> $OUT/bench_t$t.err )
  python - <<PY
import json
try:
    d = json.loads(open("$OUT/bench_t$t.json").read().strip().splitlines()[-1]); k = d["kernels_ms_per_frame"]; r = d["roofline"]
    print("triShare $t:", d["value"], d["ms_per_step"], "any", k["trace_any"], "closest", k["trace_closest"], "nodes/ray", r["node_visits_per_ray"], "sched", r["scheduling"])
except Exception as e:
    print("triShare $t ERR", e, open("$OUT/bench_t$t.err").read()[-500:])
PY
done
