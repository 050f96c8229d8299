mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT
for d in 2 3; do HWQ=8 DEPTH=$d python $R/tests/diag_e2e.py 2>&1 | tail -1; done
for d in 2 3; do HWQ=8 B=1024 DEPTH=$d python $R/tests/diag_e2e.py 2>&1 | tail -1; done
HWQ=8 B=1024 DEPTH=2 THREADS=64 python $R/tests/diag_e2e.py 2>&1 | tail -1
HWQ=16 B=1024 DEPTH=2 python $R/tests/diag_e2e.py 2>&1 | tail -1
