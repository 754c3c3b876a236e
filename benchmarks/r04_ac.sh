mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_bottleneck_gpu.py -x -q > gpurun_out/r04ac_bottleneck_test.log 2>&1; echo "bottleneck rc $?"; tail -5 gpurun_out/r04ac_bottleneck_test.log
timeout 600 python benchmarks/bottleneck_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04ac_bottleneck_bench.txt; cat gpurun_out/r04ac_bottleneck_bench.txt
STEPS=40 timeout 1200 bash benchmarks/ab_multi.sh "REFTR_BNK_V=1" "REFTR_BNK_V=3" > gpurun_out/r04ac_ab.txt 2>&1; cat gpurun_out/r04ac_ab.txt
