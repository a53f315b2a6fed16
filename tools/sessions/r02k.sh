#!/bin/bash
# GPU session r02k: NRC network -- block gathers, occupancy, LDS-staged inputs, packed fp16 grid-gradient atomics
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r02k
mkdir -p $OUT
( timeout 900 python -m pytest tests/test_gpu_nrc_net.py tests/test_gpu_nrc_render.py tests/test_gpu_shim.py -m gpu -q 2>&1 | tail -15 ) > $OUT/pytest_nrc.log
cat $OUT/pytest_nrc.log
for v in default nrc_plain nrc_w2 nrc_w4; do
  if [ $v = default ]; then unset GFX_LIB; else export GFX_LIB=$PWD/gfxexp_amd/variants/libgfxexp_$v.so; fi
  ( timeout 300 python tools/bench_nrc.py --steps 10 > $OUT/nrc_$v.json 2> $OUT/nrc_$v.err )
  echo "$v: $(cat $OUT/nrc_$v.json | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print({k:v for k,v in d.items() if k not in ("metric","config")})')"
done
unset GFX_LIB
( GFX_NRC_GRID_GRAD=f32 timeout 300 python tools/bench_nrc.py --steps 10 > $OUT/nrc_f32atomics.json 2> $OUT/nrc_f32atomics.err )
echo "f32 atomics: $(cat $OUT/nrc_f32atomics.json)"
( timeout 300 python tools/bench_renderers.py > $OUT/renderers.jsonl 2> $OUT/renderers.err )
python - <<'PY'
import json
for l in open("gpurun_out/r02k/renderers.jsonl"):
    d=json.loads(l); print(d["renderer"], d["ms_per_frame"], d["kernels_ms_per_frame"] if d["renderer"]=="nrc" else "")
PY
