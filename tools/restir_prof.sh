# rocprofv3 passes over bench.py (ReSTIR DI, configs[2] stand-in).  Output under gpurun_out/.
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
B="python bench.py --steps 6 --warmup 2 --mse-ref-spp 0 --cpu-sample 0"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/restir_stats -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/restir_pmc_sq -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_THREAD_CYCLES_VALU SQ_WAVES --output-format csv -d gpurun_out/restir_pmc_sq2 -- $B > /dev/null 2>&1
rocprofv3 --pmc TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/restir_pmc_tc -- $B > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/restir_pmc_ea -- $B > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/restir_pmc_wr -- $B > /dev/null 2>&1
for d in restir_pmc_sq restir_pmc_sq2 restir_pmc_tc restir_pmc_ea restir_pmc_wr; do
  python profiles/summarize_pmc.py gpurun_out/$d/*/*counter_collection.csv > gpurun_out/$d.txt 2>&1
done
head -30 gpurun_out/restir_stats/*/*kernel_stats.csv | cut -c1-150
