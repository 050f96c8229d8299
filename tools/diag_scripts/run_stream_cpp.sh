set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_stream_loop_example.py tests/test_gpu_ftab.py tests/test_gpu_stream.py -x -q -m gpu 2>&1 | tail -5
python tools/dump_stream.py /tmp/stream.bin 3 36 250
g++ -O2 -std=c++17 -I include examples/stream_loop.cpp -L ground-fusion2_amd/csrc -lgfbe -Wl,-rpath,$PWD/ground-fusion2_amd/csrc -o /tmp/stream_loop
/tmp/stream_loop /tmp/stream.bin /tmp/traj.bin | tee gpurun_out/r4_stream_loop_cpp.txt
/tmp/stream_loop /tmp/stream.bin /tmp/traj.bin | tee -a gpurun_out/r4_stream_loop_cpp.txt
NEW=250 python tools/diag_stream_frame_time.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_stream_frame_time.txt
