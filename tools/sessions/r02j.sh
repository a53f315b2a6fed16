#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r02j
mkdir -p $OUT
for b in 1 2 3; do
  ( GFX_TRACE_BLOCKS_PER_CU=$b timeout 600 python tools/band_kernels.py > $OUT/band_kernels_b$b.jsonl 2> $OUT/band_kernels_b$b.err )
  echo "blocks/CU $b"; python - <<PY
import json
for l in open("$OUT/band_kernels_b$b.jsonl"):
    d=json.loads(l); k=d["kernels_ms"]; print(d["band"], d["wall_ms"], k.get("trace_any"), k.get("trace_closest"), k.get("initial_candidates"))
PY
done
for r in 16 32; do
  ( GFX_TRACE_REFILL=$r timeout 600 python tools/band_kernels.py > $OUT/band_kernels_r$r.jsonl 2> $OUT/band_kernels_r$r.err )
  echo "refill $r"; python - <<PY
import json
for l in open("$OUT/band_kernels_r$r.jsonl"):
    d=json.loads(l); k=d["kernels_ms"]; print(d["band"], d["wall_ms"], k.get("trace_any"), k.get("trace_closest"), k.get("initial_candidates"))
PY
done
