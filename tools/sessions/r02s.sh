#!/bin/bash
# GPU session r02s: secondary benches for the record (renderers, configs[4], animated instances, NRC network)
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r02s
mkdir -p $OUT
( timeout 300 python tools/bench_renderers.py > $OUT/renderers.jsonl 2> $OUT/renderers.err )
( timeout 300 python tools/bench_config4.py >> $OUT/renderers.jsonl 2> $OUT/config4.err )
( timeout 300 python tools/bench_animated.py > $OUT/animated.json 2> $OUT/animated.err )
( timeout 300 python tools/bench_nrc.py --steps 10 > $OUT/nrc.json 2> $OUT/nrc.err )
( timeout 300 python tools/bench_nrc.py --steps 10 --encoding tri > $OUT/nrc_tri.json 2> $OUT/nrc_tri.err )
( timeout 600 python tools/bench_band.py > $OUT/band.json 2> $OUT/band.err )
tail -n 3 $OUT/*.err | tail -30
cat $OUT/animated.json; cat $OUT/nrc.json | cut -c1-600; cat $OUT/nrc_tri.json | cut -c1-400
python - <<'PY'
import json
for l in open("gpurun_out/r02s/renderers.jsonl"):
    d=json.loads(l); print(d.get("renderer"), d.get("ms_per_frame"), d.get("Mpaths_per_s"))
PY
