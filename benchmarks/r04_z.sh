mkdir -p gpurun_out
export TMPDIR=/tmp
STEPS=40 timeout 1500 bash benchmarks/ab_multi.sh "REFTR_OPT_LATE=0" "REFTR_OPT_LATE=1" > gpurun_out/r04z_ab.txt 2>&1; cat gpurun_out/r04z_ab.txt
timeout 900 python tools/concurrent_timeline.py --reps 50 --out gpurun_out/r04z_concurrent_timeline.txt > gpurun_out/r04z_tl.log 2>&1; tail -3 gpurun_out/r04z_tl.log; head -24 gpurun_out/r04z_concurrent_timeline.txt | tail -20
timeout 2000 python -m pytest tests/test_model_gpu.py tests/test_failsafe_gpu.py tests/test_seg_gpu.py tests/test_dp_rccl_gpu.py -x -q > gpurun_out/r04z_tests.log 2>&1; echo "tests rc $?"; tail -4 gpurun_out/r04z_tests.log
