#!/bin/bash
# GPU session r02b: inline search fallback; variants (node test via v_perm + v_fma_mix, waves/SIMD of k_initial_candidates, table only)
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r02b
mkdir -p $OUT
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 ) > $OUT/pytest.log
( timeout 600 python bench.py --steps 20 --warmup 5 --mse-ref-spp 1024 --cpu-sample 0 > $OUT/bench.json 2> $OUT/bench.err )
for v in mix w5 w6 tonly tonly5; do
  ( GFX_LIB=$PWD/gfxexp_amd/variants/libgfxexp_$v.so timeout 300 python bench.py --steps 20 --warmup 5 --mse-ref-spp 0 --cpu-sample 0 > $OUT/bench_$v.json 2> $OUT/bench_$v.err )
done
( GFX_LIB=$PWD/gfxexp_amd/variants/libgfxexp_mix.so timeout 600 python -m pytest tests/test_gpu_trace.py tests/test_gpu_fullsize.py tests/test_gpu_adversarial.py -m gpu -x -q 2>&1 | tail -8 ) > $OUT/pytest_mix.log
( timeout 600 python tools/bvh_quality.py bench $OUT/bvh_quality.json > $OUT/bvh_quality.log 2>&1 )
cat $OUT/pytest.log $OUT/pytest_mix.log
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r02b/bench*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        k = d.get("kernels_ms_per_frame", {})
        print(f.split("/")[-1], d["value"], d["ms_per_step"], {n: k.get(n) for n in ("initial_candidates", "trace_any", "trace_closest")}, d.get("roofline", {}).get("frac"))
    except Exception as e:
        print(f, "ERR", e)
PY
