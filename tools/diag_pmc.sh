# PMC passes for the bench workload (each pass in its own rocprofv3 run; --pmc never with sys/hip/hsa traces)
cd /tmp && export TMPDIR=/tmp
R=/root/repo
run() { name=$1; shift; rm -rf /tmp/pmc_$name; rocprofv3 --kernel-trace --pmc "$@" -d /tmp/pmc_$name -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /tmp/pmc_$name.json 2> $R/gpurun_out/pmc_$name.err; python $R/profiles/summarize_pmc.py /tmp/pmc_$name/*/*_results.db $R/gpurun_out/pmc_$name.txt > /dev/null; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES
run sq2 SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_VMEM SQ_WAVES
run fetch FETCH_SIZE
run write WRITE_SIZE
ls -la $R/gpurun_out/ | head
