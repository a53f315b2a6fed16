#!/bin/bash
# GPU session r03h: spatial reuse reading neighbours from a packed 80-byte mirror (experiment: separate pack kernel)
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r03h
mkdir -p $OUT
( GFX_SPATIAL_MIRROR=1 timeout 900 python -m pytest tests/test_gpu_restir.py -m gpu -q -x 2>&1 | tail -5 ) > $OUT/pytest.log
cat $OUT/pytest.log
for m in 0 1; do
  ( GFX_SPATIAL_MIRROR=$m timeout 300 python bench.py --steps 20 --warmup 5 --mse-ref-spp 0 --cpu-sample 0 > $OUT/bench_m$m.json 2> $OUT/bench_m$m.err )
  python - <<PY
import json
try:
    d = json.loads(open("$OUT/bench_m$m.json").read().strip().splitlines()[-1]); k = d["kernels_ms_per_frame"]
    print("mirror $m:", d["value"], d["ms_per_step"], "spatial", k.get("spatial_biased"), "pack", k.get("spatial_pack"), "temporal", k.get("temporal_biased"))
except Exception as e:
    print("mirror $m ERR", e, open("$OUT/bench_m$m.err").read()[-600:])
PY
done
