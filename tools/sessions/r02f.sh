#!/bin/bash
# GPU session r02f: full parity incl. solid-angle sampling; lookup variants on the plain workload
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r02f
mkdir -p $OUT
( timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -12 ) > $OUT/pytest.log
( timeout 600 python bench.py --steps 20 --warmup 5 --mse-ref-spp 0 --cpu-sample 0 --plain > $OUT/bench_plain.json 2> $OUT/bench_plain.err )
for v in search searchnoslp noslp; do
  ( GFX_LIB=$PWD/gfxexp_amd/variants/libgfxexp_$v.so timeout 600 python bench.py --steps 20 --warmup 5 --mse-ref-spp 0 --cpu-sample 0 --plain > $OUT/bench_plain_$v.json 2> $OUT/bench_plain_$v.err )
  ( GFX_LIB=$PWD/gfxexp_amd/variants/libgfxexp_$v.so timeout 600 python bench.py --steps 20 --warmup 5 --mse-ref-spp 0 --cpu-sample 0 > $OUT/bench_textured_$v.json 2> $OUT/bench_textured_$v.err )
done
cat $OUT/pytest.log
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r02f/bench*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        k = d.get("kernels_ms_per_frame", {})
        print(f.split("/")[-1], d["value"], d["ms_per_step"], {n: k.get(n) for n in ("initial_candidates", "trace_any", "trace_closest", "spatial_biased")})
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json", ".err")).read()[-800:])
PY
