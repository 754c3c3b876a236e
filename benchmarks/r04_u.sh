mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_bottleneck_gpu.py -x -q > gpurun_out/r04u_bottleneck_test.log 2>&1; echo "bottleneck rc $?"; tail -5 gpurun_out/r04u_bottleneck_test.log
timeout 2000 python -m pytest tests/test_parity_fullsize_gpu.py tests/test_seg_gpu.py tests/test_fullsize_gpu.py -x -q -s > gpurun_out/r04u_parity.log 2>&1; echo "parity rc $?"; grep -a "cfg4 RES\|passed\|failed" gpurun_out/r04u_parity.log | tail -8
