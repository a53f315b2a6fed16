#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r02u
mkdir -p $OUT
( timeout 900 python -m pytest tests/test_headless_cli.py -m gpu -q 2>&1 | tail -12 ) > $OUT/pytest.log
cat $OUT/pytest.log
