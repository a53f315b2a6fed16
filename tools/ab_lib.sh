#!/bin/bash
# A/B of two builds of the library on ONE box, alternating (run-to-run spread on a box is ~0.5 %, box-to-box ~1 %): the committed sources
# built as a variant beforehand on the host --
#   git stash; python gfxexp_amd/build.py --variant base GFX_NOOP_DEFINE=1; git stash pop; python -c "import __graft_entry__ as g; g.build()"
# -- against the working tree's libgfxexp.so.  Prints: tag, frame ms, then per-kernel ms (candidates, trace_any, gbuffer, spatial+shade, pt_fused).
#   gpurun -- 'bash tools/ab_lib.sh [bench.py flags, e.g. --config 1]'
Q="python bench.py --steps 40 --mse-ref-spp 0 --cpu-sample 0 --other-configs 0"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d["kernels_ms_per_frame"]; print(sys.argv[1], d["ms_per_step"], k.get("initial_candidates"), k.get("trace_any"), k.get("gbuffer_fused"), k.get("spatial_shade_prepare"), k.get("pt_fused"))'
for i in 1 2 3 4; do
  GFX_LIB=$PWD/gfxexp_amd/variants/libgfxexp_base.so $Q $@ 2>/dev/null | python -c "$P" base
  $Q $@ 2>/dev/null | python -c "$P" new
done
