cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=8
(timeout 600 python -m pytest tests/test_gpu_solve_kernels.py tests/test_gpu_parity.py tests/test_gpu_branches.py -x -q 2>&1 | tail -15) > gpurun_out/c2_tests.log 2>&1
(KERNELS=2,3 BIG=1 timeout 300 python tools/diag_scripts/chain_variants.py 2>&1 | tail -8) > gpurun_out/c2_chain.log 2>&1
(GFBE_LIB=$GRAFT_REPO_ROOT/ground-fusion2_amd/csrc/variants/libgfbe_chainstamp.so timeout 300 python tools/diag_scripts/chain_stamps.py 2>&1 | tail -10) > gpurun_out/c2_stamps.log 2>&1
tail -6 gpurun_out/c2_tests.log; cat gpurun_out/c2_chain.log gpurun_out/c2_stamps.log
