#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r03g
mkdir -p $OUT
( timeout 900 python -m pytest tests/test_gpu_adversarial.py -m gpu -q -x -k smooth 2>&1 | tail -25 ) > $OUT/pytest.log
cat $OUT/pytest.log
