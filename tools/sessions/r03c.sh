#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r03c
mkdir -p $OUT
( timeout 900 python -m pytest tests/test_gpu_textures.py tests/test_gpu_restir.py tests/test_gpu_pathtrace.py -m gpu -q -x 2>&1 | tail -5 ) > $OUT/pytest.log
cat $OUT/pytest.log
( timeout 300 python bench.py --steps 20 --warmup 5 --mse-ref-spp 0 --cpu-sample 0 > $OUT/bench_textured.json 2> $OUT/bench_textured.err )
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03c/bench_textured.json").read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["kernels_ms_per_frame"])
PY
