python -m pytest tests/test_gpu_parity.py tests/test_gpu_branches.py tests/test_gpu_plane.py tests/test_gpu_lio_joint.py -x -q 2>&1 | tail -5
python tests/diag_single.py 2>&1 | tail -3
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_single; N=30 rocprofv3 --kernel-trace --stats -d /tmp/prof_single -- python $R/tests/diag_single.py > /tmp/s.log 2>&1
python $R/profiles/summarize_rocpd.py /tmp/prof_single/*/*_results.db /tmp/single_trace.txt | head -16
