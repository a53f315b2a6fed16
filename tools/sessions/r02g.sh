#!/bin/bash
# GPU session r02g: full parity (new: shim program, output chain, NRC fp32 / full-size, solid angle), benches
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r02g
mkdir -p $OUT
( timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -25 ) > $OUT/pytest.log
( timeout 600 python bench.py --steps 20 --warmup 5 --mse-ref-spp 0 --cpu-sample 0 --plain > $OUT/bench_plain.json 2> $OUT/bench_plain.err )
( timeout 600 python bench.py --steps 20 --warmup 5 --mse-ref-spp 0 --cpu-sample 0 > $OUT/bench_textured.json 2> $OUT/bench_textured.err )
cat $OUT/pytest.log
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r02g/bench*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], d["value"], d["ms_per_step"], d.get("kernels_ms_per_frame"))
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json", ".err")).read()[-800:])
PY
