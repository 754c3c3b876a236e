#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_failsafe_gpu.py tests/test_model_gpu.py tests/test_decoder_coop_gpu.py -x -q -m gpu > gpurun_out/r04bj_tests.log 2>&1; echo "rc $?"; tail -3 gpurun_out/r04bj_tests.log
STEPS=50 bash benchmarks/ab_multi.sh "REFTR_TAIL_FUSE=0" "REFTR_TAIL_FUSE=1" > gpurun_out/r04bj_ab.txt 2>&1
STEPS=50 bash benchmarks/ab_multi.sh "REFTR_TAIL_FUSE=1" "REFTR_TAIL_FUSE=0" >> gpurun_out/r04bj_ab.txt 2>&1
cat gpurun_out/r04bj_ab.txt
