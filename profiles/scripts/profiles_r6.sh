# Round-6 profile set (run through gpurun from the repo root: `gpurun -- 'bash profiles/scripts/profiles_r6.sh'`; writes gpurun_out/r6_*,
# copied to profiles/ afterwards), ALL from the build at HEAD: kernel stats of the default bench workload, the four PMC passes (each in
# its own rocprofv3 run; --pmc never with sys / hip / hsa traces), the bench line (reads the PMC summaries of THIS build), the
# single-window kernel trace, the chain kernels' phase stamps and the three factorisations side by side, the speculative linearisation on
# against off, the eigen square root of the prior, the GNSS window, the end-to-end loop under the tracer, the one-robot frame loop, the parity soaks.
mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; export GPU_MAX_HW_QUEUES=8
cd /tmp
rm -rf /tmp/ks; rocprofv3 --kernel-trace --stats -d /tmp/ks -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --no-single --mixed 0 --no-other-configs > $R/gpurun_out/r6_bench_under_rocprof.json 2> /tmp/ks.err
python $R/profiles/summarize_rocpd.py /tmp/ks/*/*_results.db $R/gpurun_out/r6_kernel_stats_b8192.txt | head -14
run() { name=$1; shift; rm -rf /tmp/pmc_$name; rocprofv3 --kernel-trace --pmc "$@" -d /tmp/pmc_$name -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-single --mixed 0 --no-other-configs > /tmp/pmc_$name.json 2> $R/gpurun_out/pmc_$name.err; python $R/profiles/summarize_pmc.py /tmp/pmc_$name/*/*_results.db $R/gpurun_out/r6_pmc_$name.txt > /dev/null; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES
run sq2 SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_VMEM SQ_WAVES
run fetch FETCH_SIZE
run write WRITE_SIZE
head -4 $R/gpurun_out/r6_pmc_fetch.txt; head -4 $R/gpurun_out/r6_pmc_write.txt
(cd $R && python -c "import bench; print(bench.kernel_source_digest())" > gpurun_out/r6_pmc_source_digest.txt; cat gpurun_out/r6_pmc_source_digest.txt)
# the bench line last: its roofline block reads the PMC summaries of THIS build from profiles/ (traffic, matrix-core busy fraction)
cp $R/gpurun_out/r6_pmc_sq1.txt $R/gpurun_out/r6_pmc_sq2.txt $R/gpurun_out/r6_pmc_fetch.txt $R/gpurun_out/r6_pmc_write.txt $R/gpurun_out/r6_pmc_source_digest.txt $R/profiles/
(cd $R && python bench.py > gpurun_out/r6_bench_default.json 2> gpurun_out/r6_bench_default.err; head -c 300 gpurun_out/r6_bench_default.json; echo)
rm -rf /tmp/prof_single; N=30 rocprofv3 --kernel-trace --stats -d /tmp/prof_single -- python $R/tools/diag_single.py > $R/gpurun_out/r6_single.log 2>&1
python $R/profiles/summarize_rocpd.py /tmp/prof_single/*/*_results.db $R/gpurun_out/r6_single_trace.txt | head -14
cd $R
python tools/diag_single.py 2>&1 | tail -3 | tee -a gpurun_out/r6_single.log
(KERNELS=2,3,1 BIG=1 python tools/diag_scripts/chain_variants.py 2>&1 | grep -v amdgpu.ids; GFBE_LIB=$R/ground-fusion2_amd/csrc/variants/libgfbe_chainstamp.so python tools/diag_scripts/chain_stamps.py 2>&1 | grep -v amdgpu.ids) > gpurun_out/r6_solve_chain_phases.txt; tail -9 gpurun_out/r6_solve_chain_phases.txt
# (round 6) the eigen square root by divide & conquer against the QL build and the oracle; the whole GPU suite of the tree the set is taken from
(GFBE_QL_LIB=$R/ground-fusion2_amd/csrc/variants/libgfbe_ql.so python tools/diag_scripts/eig_dc_check.py 2>&1 | grep -v amdgpu.ids) > gpurun_out/r6_eigen_dc_check.txt; tail -1 gpurun_out/r6_eigen_dc_check.txt
(python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|^FAILED|^ERROR") > gpurun_out/r6_gpu_tests.txt; cat gpurun_out/r6_gpu_tests.txt
# speculative linearisation on against off (every output bit for bit; single-window times; 8192 resident windows), the phases of the
# candidate's linearisation launch of a single window, and what a lone wave costs on this device (the model behind the latency path)
GFBE_LIB=$R/ground-fusion2_amd/csrc/libgfbe_diag.so python tools/diag_scripts/eig_time.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r6_eigen_prior.txt; cat gpurun_out/r6_eigen_prior.txt
python tools/diag_scripts/plane_kernels.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r6_plane_kernels.txt
python tools/diag_scripts/gnss_prof.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r6_gnss_window_profile.txt; head -3 gpurun_out/r6_gnss_window_profile.txt
# the pose graph (BASELINE configs[3]) under the tracer and beside the oracle; the soak's one diverging gimbal-lock window against the oracle's own sensitivity
(cd /tmp; rm -rf /tmp/pg; N=20 rocprofv3 --kernel-trace --stats -d /tmp/pg -- python $R/tools/diag_scripts/pg_trace.py 2>&1 | grep "pose graph"; python $R/profiles/summarize_rocpd.py /tmp/pg/*/*_results.db $R/gpurun_out/r6_posegraph_trace.txt | head -4)
python tools/diag_posegraph_bench.py 2>&1 | grep "poses:" | tee -a gpurun_out/r6_posegraph_trace.txt
python tools/diag_scripts/gimbal_divergence.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r6_gimbal_divergence.txt; tail -3 gpurun_out/r6_gimbal_divergence.txt
cd /tmp; rm -rf /tmp/pe; B=1024 DEPTH=3 STEPS=10 rocprofv3 --kernel-trace --memory-copy-trace --stats -d /tmp/pe -- python $R/tools/diag_e2e.py > /tmp/pe.log 2>&1; grep "e2e host-fed" /tmp/pe.log > $R/gpurun_out/r6_e2e.log
python $R/profiles/summarize_rocpd.py /tmp/pe/*/*_results.db $R/gpurun_out/r6_e2e_trace.txt | head -8; cat $R/gpurun_out/r6_e2e.log
cd $R
NEW=250 python tools/diag_stream_frame_time.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r6_stream_frame_time.txt; tail -12 gpurun_out/r6_stream_frame_time.txt
python tools/dump_stream.py /tmp/stream.bin 3 36 250 > /dev/null && g++ -O2 -std=c++17 -I include examples/stream_loop.cpp -L ground-fusion2_amd/csrc -lgfbe -Wl,-rpath,$PWD/ground-fusion2_amd/csrc -o /tmp/stream_loop && (/tmp/stream_loop /tmp/stream.bin /tmp/traj.bin; /tmp/stream_loop /tmp/stream.bin /tmp/traj.bin) | tee gpurun_out/r6_stream_loop_cpp.txt
N=${SOAK_N:-400} python tools/diag_soak.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r6_soak.txt; tail -7 gpurun_out/r6_soak.txt
python tools/diag_soak_batch.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r6_soak_batch.txt; tail -5 gpurun_out/r6_soak_batch.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
