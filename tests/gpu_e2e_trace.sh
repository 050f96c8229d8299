# kernel-trace summary of the end-to-end loop alone (1024 windows per batch): how busy is the GPU per batch
mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
B=1024 DEPTH=3 STEPS=10 python $R/tests/diag_e2e.py 2>&1 | tail -1
rm -rf /tmp/pe; B=1024 DEPTH=3 STEPS=10 rocprofv3 --kernel-trace --memory-copy-trace --stats -d /tmp/pe -- python $R/tests/diag_e2e.py > /tmp/pe.log 2>&1; tail -1 /tmp/pe.log
python $R/profiles/summarize_rocpd.py /tmp/pe/*/*_results.db $R/gpurun_out/r2_e2e_trace.txt | head -24
ls /tmp/pe/*/ | head
