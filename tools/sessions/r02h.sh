#!/bin/bash
# GPU session r02h: refactored frame loop (frame program) + strip exchange on one GPU (loopback), all parity tests, band bound, benches
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r02h
mkdir -p $OUT
( timeout 600 python -m pytest tests/test_gpu_strip_exchange.py -m gpu -q -x 2>&1 | tail -40 ) > $OUT/pytest_strips.log
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -25 ) > $OUT/pytest.log
( timeout 600 python bench.py --steps 20 --warmup 5 --mse-ref-spp 0 --cpu-sample 0 --plain > $OUT/bench_plain.json 2> $OUT/bench_plain.err )
( timeout 600 python bench.py --steps 20 --warmup 5 --mse-ref-spp 0 --cpu-sample 0 > $OUT/bench_textured.json 2> $OUT/bench_textured.err )
( timeout 900 python tools/bench_band.py > $OUT/band.json 2> $OUT/band.err )
( timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 2 --mse-ref-spp 0 --cpu-sample 0 --no-roofline > $OUT/bench_torchrun1.json 2> $OUT/bench_torchrun1.err )
cat $OUT/pytest_strips.log $OUT/pytest.log
cat $OUT/band.json; tail -3 $OUT/band.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r02h/bench*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], d["value"], d["ms_per_step"], d.get("kernels_ms_per_frame"))
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json", ".err")).read()[-800:])
PY
