#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r03f
mkdir -p $OUT
( timeout 300 python tools/bench_renderers.py > $OUT/renderers.jsonl 2> $OUT/renderers.err )
( timeout 300 python tools/bench_config4.py >> $OUT/renderers.jsonl 2> $OUT/config4.err )
python - <<'PY'
import json
for l in open("gpurun_out/r03f/renderers.jsonl"):
    d=json.loads(l); print(d.get("renderer"), d.get("ms_per_frame"), d.get("kernels_ms_per_frame") if d.get("renderer")=="nrc" else "")
PY
