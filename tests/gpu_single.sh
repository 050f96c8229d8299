mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
python $R/tests/diag_single.py 2>&1 | tail -3
rm -rf /tmp/prof_single; N=30 rocprofv3 --kernel-trace --stats -d /tmp/prof_single -- python $R/tests/diag_single.py > /tmp/prof_single.log 2>&1; tail -3 /tmp/prof_single.log
python $R/profiles/summarize_rocpd.py /tmp/prof_single/*/*_results.db $R/gpurun_out/single_trace.txt | head -40
