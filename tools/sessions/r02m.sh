#!/bin/bash
# GPU session r02m: CLI parity, cluttered scene + per-ray histogram, full suite, default bench (64k-spp MSE, CPU rows)
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r02m
mkdir -p $OUT
( timeout 900 python -m pytest tests/test_headless_cli.py tests/test_gpu_trace.py -m gpu -q 2>&1 | tail -20 ) > $OUT/pytest_new.log
cat $OUT/pytest_new.log
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15 ) > $OUT/pytest.log
cat $OUT/pytest.log
( timeout 600 python tools/bvh_quality.py bench $OUT/bvh_quality_bench.json > $OUT/bvh_quality_bench.log 2>&1 )
( timeout 600 python tools/bvh_quality.py bench-cluttered $OUT/bvh_quality_cluttered.json > $OUT/bvh_quality_cluttered.log 2>&1 )
tail -c 1500 $OUT/bvh_quality_cluttered.log
( timeout 600 python bench.py --steps 20 --warmup 5 --mse-ref-spp 0 --cpu-sample 0 --cluttered > $OUT/bench_cluttered.json 2> $OUT/bench_cluttered.err )
( time timeout 1700 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2> $OUT/bench_default.time
cat $OUT/bench_default.time
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r02m/bench*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], d["value"], d["ms_per_step"], d.get("kernels_ms_per_frame"), d.get("mse"), d.get("cpu_baseline"))
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json", ".err")).read()[-800:])
PY
