#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r03j
mkdir -p $OUT
( timeout 600 python tools/two_stream_bands.py > $OUT/two_stream.json 2> $OUT/two_stream.err ); cat $OUT/two_stream.json; tail -3 $OUT/two_stream.err
