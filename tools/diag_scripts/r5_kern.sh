cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=8
(timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_uncleared.py tests/test_gpu_branches.py tests/test_gpu_lio_joint.py -x -q 2>&1 | tail -3)
B=2048 timeout 300 python tools/diag_scripts/schur_tile_time.py 2>&1 | grep "us per launch"
timeout 300 python bench.py --no-cpu-baseline --no-single --mixed 0 --no-other-configs --no-e2e 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('value', round(d['value']), 'r1024', round(d['resident_1024']['value']), 'single', d['single_window_ms'])"
