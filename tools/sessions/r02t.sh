#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r02t
mkdir -p $OUT
( timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x -k eight_band 2>&1 | tail -30 ) > $OUT/pytest_8band.log
cat $OUT/pytest_8band.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1 ); tail -3 $OUT/smoke.log
