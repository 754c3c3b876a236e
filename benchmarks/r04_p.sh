mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_bottleneck_gpu.py -x -q > gpurun_out/r04p_bottleneck_test.log 2>&1; echo "bottleneck rc $?"; tail -15 gpurun_out/r04p_bottleneck_test.log
timeout 600 python benchmarks/bottleneck_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04p_bottleneck_bench.txt; cat gpurun_out/r04p_bottleneck_bench.txt
timeout 1200 python -m pytest tests/test_gemm_gpu.py -x -q -k "linear_fwd" > gpurun_out/r04p_gemm.log 2>&1; echo "gemm rc $?"; tail -3 gpurun_out/r04p_gemm.log
STEPS=40 timeout 1200 bash benchmarks/ab_multi.sh "REFTR_L1_FUSE=0 REFTR_DEEP=0" "REFTR_L1_FUSE=1 REFTR_DEEP=0" "REFTR_L1_FUSE=0 REFTR_DEEP=1" "REFTR_L1_FUSE=1 REFTR_DEEP=1" > gpurun_out/r04p_ab.txt 2>&1; cat gpurun_out/r04p_ab.txt
