mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_branches.py -x -q 2>&1 | tail -5
python tests/diag_single.py 2>&1 | tail -3
python tests/diag_timing.py 2>&1 | tail -25
