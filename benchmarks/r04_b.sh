mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_failsafe_gpu.py -x -q > gpurun_out/r04b_failsafe.log 2>&1; echo "failsafe rc $?" 
tail -8 gpurun_out/r04b_failsafe.log
timeout 900 python -m pytest tests/test_model_gpu.py -x -q > gpurun_out/r04b_model.log 2>&1; echo "model rc $?"
tail -5 gpurun_out/r04b_model.log
VAR=REFTR_OPT_EMIT VALS="0 1" timeout 600 bash benchmarks/ab_env.sh > gpurun_out/r04b_ab_emit.txt 2>&1; cat gpurun_out/r04b_ab_emit.txt
