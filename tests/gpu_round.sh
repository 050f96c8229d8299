#!/bin/bash
# One GPU-box visit: parity tests, kernel variants side by side, the bench line. Outputs under gpurun_out/.
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 900 python -m pytest tests -m gpu -q --timeout 600 ) > gpurun_out/pytest_gpu.log 2>&1
tail -25 gpurun_out/pytest_gpu.log
for v in ${TEST_VARIANTS}; do
  ( time GFBE_LIB=$PWD/ground-fusion2_amd/csrc/variants/libgfbe_$v.so timeout 900 python -m pytest tests -m gpu -q --timeout 600 ) > gpurun_out/pytest_gpu_$v.log 2>&1
  echo "== pytest with variant $v"; tail -8 gpurun_out/pytest_gpu_$v.log
done
if [ -n "${VARIANTS}" ]; then ( time timeout 600 python tests/diag_variants.py run ${VARIANTS} ) > gpurun_out/variants.log 2>&1; cat gpurun_out/variants.log; fi
if [ -n "${STAMP}" ]; then GFBE_LIB=$PWD/ground-fusion2_amd/csrc/variants/libgfbe_stamp.so timeout 300 python tests/diag_kvis_stamp.py > gpurun_out/kvis_stamp.log 2>&1; cat gpurun_out/kvis_stamp.log; fi
( time GFBE_DEBUG_UPLOAD=${DEBUG_UPLOAD:-} timeout 900 python bench.py --steps 10 --warmup 3 --cpu-seconds 6 ${BENCH_ARGS} ) > gpurun_out/bench.json 2> gpurun_out/bench.err
tail -c 5000 gpurun_out/bench.json; tail -5 gpurun_out/bench.err
