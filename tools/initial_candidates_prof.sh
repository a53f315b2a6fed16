# rocprofv3 PMC passes used for profiles/r01g_initial_candidates_experiments.txt (TA / TCP / TCC view of k_initial_candidates).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
B="python bench.py --steps 6 --warmup 2 --mse-ref-spp 0 --cpu-sample 0"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/g_pmc_sq -- $B > /dev/null 2>&1
python profiles/summarize_pmc.py gpurun_out/g_pmc_sq/*/*counter_collection.csv 2>&1 | grep -A12 initial_cand | head -30
