#!/bin/bash
# GPU session r02v: k_initial_candidates with pooled BSDF evaluations (GFX_INIT_COOP)
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r02v
mkdir -p $OUT
( timeout 900 python -m pytest tests/test_gpu_restir.py tests/test_gpu_textures.py tests/test_gpu_adversarial.py -m gpu -q -x 2>&1 | tail -15 ) > $OUT/pytest.log
cat $OUT/pytest.log
for c in 1 0; do
  for w in plain textured; do
    flag=""; if [ $w = plain ]; then flag="--plain"; fi
    ( GFX_INIT_COOP=$c timeout 300 python bench.py --steps 20 --warmup 5 --mse-ref-spp 0 --cpu-sample 0 $flag > $OUT/bench_${w}_coop$c.json 2> $OUT/bench_${w}_coop$c.err )
    python - <<PY
import json
try:
    d = json.loads(open("$OUT/bench_${w}_coop$c.json").read().strip().splitlines()[-1]); k = d["kernels_ms_per_frame"]
    print("coop $c $w:", d["value"], d["ms_per_step"], "initial", k["initial_candidates"])
except Exception as e:
    print("coop $c $w ERR", e, open("$OUT/bench_${w}_coop$c.err").read()[-600:])
PY
  done
done
