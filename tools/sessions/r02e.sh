#!/bin/bash
# GPU session r02e: templated emittance-texture path, embedded descriptors; -fno-slp-vectorize variant
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r02e
mkdir -p $OUT
( timeout 900 python -m pytest tests/test_gpu_textures.py tests/test_gpu_restir.py -m gpu -x -q 2>&1 | tail -8 ) > $OUT/pytest.log
for mode in "" "--plain"; do
  tag=${mode:-textured}; tag=${tag#--}
  ( timeout 600 python bench.py --steps 20 --warmup 5 --mse-ref-spp 0 --cpu-sample 0 $mode > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err )
  ( GFX_LIB=$PWD/gfxexp_amd/variants/libgfxexp_noslp.so timeout 600 python bench.py --steps 20 --warmup 5 --mse-ref-spp 0 --cpu-sample 0 $mode > $OUT/bench_${tag}_noslp.json 2> $OUT/bench_${tag}_noslp.err )
done
( GFX_LIB=$PWD/gfxexp_amd/variants/libgfxexp_noslp.so timeout 900 python -m pytest tests/test_gpu_restir.py tests/test_gpu_pathtrace.py tests/test_gpu_trace.py -m gpu -x -q 2>&1 | tail -5 ) > $OUT/pytest_noslp.log
( timeout 300 python tools/bench_renderers.py > $OUT/renderers.jsonl 2> $OUT/renderers.err )
( GFX_LIB=$PWD/gfxexp_amd/variants/libgfxexp_noslp.so timeout 300 python tools/bench_renderers.py > $OUT/renderers_noslp.jsonl 2> $OUT/renderers_noslp.err )
cat $OUT/pytest.log $OUT/pytest_noslp.log
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r02e/bench*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], d["value"], d["ms_per_step"], d.get("kernels_ms_per_frame"))
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json", ".err")).read()[-800:])
for f in sorted(glob.glob("gpurun_out/r02e/renderers*.jsonl")):
    for l in open(f):
        try:
            d = json.loads(l); print(f.split("/")[-1], d.get("renderer"), d.get("ms_per_frame"), d.get("mpaths_per_s"))
        except Exception: pass
PY
