mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -x -q -m gpu > gpurun_out/r04t_gpu_tests.log 2>&1; echo "gpu tests rc $?"; tail -8 gpurun_out/r04t_gpu_tests.log
