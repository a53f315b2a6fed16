# rocprofv3 PMC passes over tools/bench_nrc.py for k_nrc_train (profiles/r01c_nrc_network.txt).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
B="python tools/bench_nrc.py --steps 5"
run() { n=$1; shift; timeout 200 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/nt_$n -- $B > /dev/null 2>&1; python profiles/summarize_pmc.py gpurun_out/nt_$n/*/*counter_collection.csv 2>&1 | grep -A8 "k_nrc_train" | head -9; }
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES
run b SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_FLAT SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16
run c TCC_ATOMIC_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCC_EA0_ATOMIC_sum WRITE_SIZE
timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/nt_stats -- $B > /dev/null 2>&1
head -8 gpurun_out/nt_stats/*/*kernel_stats.csv | cut -c1-120
