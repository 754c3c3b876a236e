mkdir -p gpurun_out
export TMPDIR=/tmp
STEPS=40 timeout 1500 bash benchmarks/ab_multi.sh "REFTR_ZERO_SIDE=0" "REFTR_ZERO_SIDE=2" > gpurun_out/r04ai_ab.txt 2>&1; cat gpurun_out/r04ai_ab.txt
STEPS=40 timeout 1500 bash benchmarks/ab_multi.sh "REFTR_ZERO_SIDE=0" "REFTR_ZERO_SIDE=2" >> gpurun_out/r04ai_ab.txt 2>&1; tail -4 gpurun_out/r04ai_ab.txt
timeout 2000 python -m pytest tests/test_model_gpu.py tests/test_failsafe_gpu.py tests/test_seg_gpu.py -x -q > gpurun_out/r04ai_tests.log 2>&1; echo "tests rc $?"; tail -3 gpurun_out/r04ai_tests.log
