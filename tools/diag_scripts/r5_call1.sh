cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=8
(timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -25) > gpurun_out/c1_tests.log 2>&1
(timeout 300 python tools/diag_scripts/chain_variants.py 2>&1 | tail -30) > gpurun_out/c1_chain_default.log 2>&1
for v in rot0 rot1; do
  (KERNELS=2 GFBE_LIB=$GRAFT_REPO_ROOT/ground-fusion2_amd/csrc/variants/libgfbe_$v.so timeout 300 python tools/diag_scripts/chain_variants.py 2>&1 | tail -12) > gpurun_out/c1_chain_$v.log 2>&1
done
tail -5 gpurun_out/c1_tests.log; cat gpurun_out/c1_chain_default.log gpurun_out/c1_chain_rot0.log gpurun_out/c1_chain_rot1.log
