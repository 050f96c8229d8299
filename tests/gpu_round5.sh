mkdir -p gpurun_out
( time timeout 900 python -m pytest tests -m gpu -q --timeout 600 ) > gpurun_out/pytest_gpu.log 2>&1; tail -12 gpurun_out/pytest_gpu.log
echo "== reduced panel"; B=1024 SPLIT=1 python tests/diag_kvis.py 2>&1 | grep -E "k_vis_lin|k_schur |k_solve |solves_per_s|final_cost"
echo "== full panel"; GFBE_VIS_FULL=1 B=1024 SPLIT=1 python tests/diag_kvis.py 2>&1 | grep -E "k_vis_lin|solves_per_s|final_cost"
( time timeout 900 python bench.py --steps 10 --warmup 3 --cpu-seconds 6 ) > gpurun_out/bench.json 2> gpurun_out/bench.err
tail -c 4500 gpurun_out/bench.json; tail -5 gpurun_out/bench.err
