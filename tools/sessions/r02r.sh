#!/bin/bash
# GPU session r02r: NRC band renderers (loopback), full suite
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r02r
mkdir -p $OUT
( timeout 900 python -m pytest tests/test_gpu_strip_exchange.py -m gpu -q -x 2>&1 | tail -30 ) > $OUT/pytest_strips.log
cat $OUT/pytest_strips.log
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15 ) > $OUT/pytest.log
cat $OUT/pytest.log
( timeout 300 python tools/bench_renderers.py > $OUT/renderers.jsonl 2> $OUT/renderers.err )
python - <<'PY'
import json
for l in open("gpurun_out/r02r/renderers.jsonl"):
    d=json.loads(l); print(d["renderer"], d["ms_per_frame"])
PY
