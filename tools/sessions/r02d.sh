#!/bin/bash
# GPU session r02d: texture path parity + textured bench
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r02d
mkdir -p $OUT
( timeout 900 python -m pytest tests/test_gpu_textures.py -m gpu -x -q 2>&1 | tail -25 ) > $OUT/pytest_tex.log
( timeout 900 python -m pytest tests -m gpu -q --deselect tests/test_gpu_textures.py 2>&1 | tail -15 ) > $OUT/pytest.log
( timeout 600 python bench.py --steps 20 --warmup 5 --mse-ref-spp 512 --cpu-sample 0 > $OUT/bench_textured.json 2> $OUT/bench_textured.err )
( timeout 600 python bench.py --steps 20 --warmup 5 --mse-ref-spp 0 --cpu-sample 0 --bump 0 > $OUT/bench_textured_nobump.json 2> $OUT/bench_textured_nobump.err )
( timeout 600 python bench.py --steps 20 --warmup 5 --mse-ref-spp 0 --cpu-sample 0 --plain > $OUT/bench_plain.json 2> $OUT/bench_plain.err )
cat $OUT/pytest_tex.log $OUT/pytest.log
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r02d/bench*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], d["value"], d["ms_per_step"], d.get("kernels_ms_per_frame"))
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json", ".err")).read()[-1500:])
PY
