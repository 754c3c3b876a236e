mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_failsafe_gpu.py -x -q > gpurun_out/r04c_failsafe.log 2>&1; echo "failsafe rc $?" 
tail -12 gpurun_out/r04c_failsafe.log
timeout 1200 python -m pytest tests/test_model_gpu.py tests/test_dp_rccl_gpu.py tests/test_seg_gpu.py tests/test_gemm_gpu.py -x -q > gpurun_out/r04c_model.log 2>&1; echo "model rc $?"
tail -5 gpurun_out/r04c_model.log
VAR=REFTR_FUSED_NORM VALS="0 1" timeout 600 bash benchmarks/ab_env.sh > gpurun_out/r04c_ab_norm.txt 2>&1; cat gpurun_out/r04c_ab_norm.txt
cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_r04c -o r04c -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-kernel-roofline --steps 30 > $GRAFT_REPO_ROOT/gpurun_out/r04c_rocprof.log 2>&1; cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py $(find /tmp/prof_r04c -name "*.db" | head -1) > gpurun_out/r04c_kernel_stats.md 2>&1; head -30 gpurun_out/r04c_kernel_stats.md
python tools/step_phases.py $(find /tmp/prof_r04c -name "*.db" | head -1) > gpurun_out/r04c_phases.txt 2>&1; cat gpurun_out/r04c_phases.txt
