mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gemm_gpu.py -x -q -k "dilated" > gpurun_out/r04ad_gemm.log 2>&1; echo "gemm rc $?"; tail -12 gpurun_out/r04ad_gemm.log
timeout 1500 python -m pytest tests/test_model_gpu.py -x -q -k "dilation or learned" > gpurun_out/r04ad_model.log 2>&1; echo "model rc $?"; tail -25 gpurun_out/r04ad_model.log
