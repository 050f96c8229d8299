cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=8
(timeout 600 python -m pytest tests/test_gpu_ftab.py tests/test_gpu_stream.py tests/test_stream_loop_example.py -x -q 2>&1 | tail -3)
timeout 600 python bench.py --no-cpu-baseline --no-single --mixed 0 --no-other-configs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('value', round(d['value'])); print(json.dumps(d['end_to_end'], indent=0)[:1200])"
