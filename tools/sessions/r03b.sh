#!/bin/bash
# GPU session r03b: compact normal-matrix array for the light candidates
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r03b
mkdir -p $OUT
( timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 ) > $OUT/pytest.log
cat $OUT/pytest.log
for w in plain textured; do
  flag=""; if [ $w = plain ]; then flag="--plain"; fi
  ( timeout 300 python bench.py --steps 20 --warmup 5 --mse-ref-spp 0 --cpu-sample 0 $flag > $OUT/bench_$w.json 2> $OUT/bench_$w.err )
  python - <<PY
import json
try:
    d = json.loads(open("$OUT/bench_$w.json").read().strip().splitlines()[-1]); k = d["kernels_ms_per_frame"]
    print("$w:", d["value"], d["ms_per_step"], k)
except Exception as e:
    print("$w ERR", e, open("$OUT/bench_$w.err").read()[-600:])
PY
done
( timeout 300 python tools/bench_renderers.py > $OUT/renderers.jsonl 2> $OUT/renderers.err )
python - <<'PY'
import json
for l in open("gpurun_out/r03b/renderers.jsonl"):
    d=json.loads(l); print(d["renderer"], d["ms_per_frame"])
PY
