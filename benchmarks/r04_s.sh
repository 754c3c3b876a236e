mkdir -p gpurun_out
export TMPDIR=/tmp
STEPS=40 timeout 1500 bash benchmarks/ab_multi.sh "REFTR_OPT_PIPE=0" "REFTR_OPT_PIPE=1" "REFTR_OPT_PIPE=1 REFTR_OPT_PIPE_LAYERS=4" "REFTR_OPT_PIPE=1 REFTR_OPT_PIPE_LAYERS=2" > gpurun_out/r04s_ab.txt 2>&1; cat gpurun_out/r04s_ab.txt
timeout 900 python tools/concurrent_timeline.py --reps 50 --out gpurun_out/r04s_concurrent_timeline.txt > gpurun_out/r04s_tl.log 2>&1; tail -3 gpurun_out/r04s_tl.log; head -48 gpurun_out/r04s_concurrent_timeline.txt
timeout 1500 python -m pytest tests/test_failsafe_gpu.py tests/test_model_gpu.py -x -q > gpurun_out/r04s_tests.log 2>&1; echo "tests rc $?"; tail -5 gpurun_out/r04s_tests.log
