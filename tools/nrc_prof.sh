set -x
cd $GRAFT_REPO_ROOT
python tools/bench_nrc.py --steps 20 2>&1 | tail -1 > gpurun_out/nrc_bench.json
cat gpurun_out/nrc_bench.json
python tools/bench_nrc.py --steps 20 --hidden 5 2>&1 | tail -1
python tools/bench_nrc.py --steps 20 --encoding tri 2>&1 | tail -1
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/nrc_stats -- python tools/bench_nrc.py --steps 10 > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/nrc_pmc -- python tools/bench_nrc.py --steps 5 > /dev/null 2>&1
ls -R gpurun_out/nrc_stats gpurun_out/nrc_pmc | head -30
