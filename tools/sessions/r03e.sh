#!/bin/bash
# GPU session r03e: end-of-round record after the emitter-record compaction: full suite, default bench, profiles, secondary benches
set -u
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r03e
mkdir -p $OUT
cd $ROOT
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -8 ) > $OUT/pytest.log
cat $OUT/pytest.log
( time timeout 1700 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2> $OUT/bench_default.time
( timeout 600 python bench.py --steps 20 --warmup 5 --mse-ref-spp 0 --cpu-sample 0 --plain > $OUT/bench_plain.json 2> $OUT/bench_plain.err )
( timeout 600 python bench.py --steps 20 --warmup 5 --mse-ref-spp 0 --cpu-sample 0 --cluttered > $OUT/bench_cluttered.json 2> $OUT/bench_cluttered.err )
( timeout 300 python tools/bench_renderers.py > $OUT/renderers.jsonl 2> $OUT/renderers.err )
( timeout 300 python tools/bench_config4.py >> $OUT/renderers.jsonl 2> $OUT/config4.err )
( timeout 300 python tools/bench_animated.py > $OUT/animated.json 2> $OUT/animated.err )
( timeout 900 python tools/bench_band.py > $OUT/band.json 2> $OUT/band.err )
( timeout 600 python tools/band_kernels.py > $OUT/band_kernels.jsonl 2> $OUT/band_kernels.err )
cd /tmp
B="python $ROOT/bench.py --steps 6 --warmup 2 --mse-ref-spp 0 --cpu-sample 0"
( timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/stats -o bench -- $B > $OUT/bench_under_rocprof.json 2> $OUT/stats.err )
pmc() { name=$1; shift; ( timeout 600 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- $B --no-roofline > /dev/null 2> $OUT/$name.err ); python $ROOT/profiles/summarize_pmc.py $OUT/$name/*/*counter_collection.csv > $OUT/$name.txt 2>&1; rm -rf $OUT/$name; }
pmc pmc_sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_THREAD_CYCLES_VALU
pmc pmc_sq2 SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_INSTS_SMEM
pmc pmc_tc TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum
pmc pmc_ta TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum
pmc pmc_fetch FETCH_SIZE
pmc pmc_write WRITE_SIZE
cd $ROOT
db=$(find $OUT/stats -name "*.db" | head -1)
python profiles/summarize_rocpd.py $db > $OUT/kernel_stats.txt
rm -rf $OUT/stats
head -14 $OUT/kernel_stats.txt
cat $OUT/bench_default.time
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r03e/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], d["value"], d["ms_per_step"], d.get("kernels_ms_per_frame"), d.get("mse", {}).get("mse") if d.get("mse") else None)
    except Exception as e:
        print(f, "ERR", e)
for l in open("gpurun_out/r03e/renderers.jsonl"):
    d=json.loads(l); print(d.get("renderer"), d.get("ms_per_frame"))
print(open("gpurun_out/r03e/band.json").read()[:1500])
PY
