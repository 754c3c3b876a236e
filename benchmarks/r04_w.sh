mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_failsafe_gpu.py -x -q > gpurun_out/r04w_failsafe.log 2>&1; echo "failsafe rc $?"; tail -15 gpurun_out/r04w_failsafe.log
STEPS=40 timeout 1200 bash benchmarks/ab_multi.sh "REFTR_OPT_SPARSE=0" "REFTR_OPT_SPARSE=1" > gpurun_out/r04w_ab.txt 2>&1; cat gpurun_out/r04w_ab.txt
