#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r02l
mkdir -p $OUT
for v in nrc_p2 nrc_p3 nrc_p4; do
  export GFX_LIB=$PWD/gfxexp_amd/variants/libgfxexp_$v.so
  ( timeout 300 python tools/bench_nrc.py --steps 10 > $OUT/nrc_$v.json 2> $OUT/nrc_$v.err )
  echo "$v: $(cat $OUT/nrc_$v.json | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["kernels_ms_per_frame"])')"
done
