#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
REFTR_WG_DEFER4=1 timeout 600 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "golden or captured_step" > gpurun_out/r04bi_tests.log 2>&1; echo "rc $?"; tail -2 gpurun_out/r04bi_tests.log
STEPS=50 bash benchmarks/ab_multi.sh "REFTR_WG_DEFER4=0" "REFTR_WG_DEFER4=1" > gpurun_out/r04bi_ab.txt 2>&1
STEPS=50 bash benchmarks/ab_multi.sh "REFTR_WG_DEFER4=1" "REFTR_WG_DEFER4=0" >> gpurun_out/r04bi_ab.txt 2>&1
cat gpurun_out/r04bi_ab.txt
REFTR_WG_DEFER4=1 python tools/concurrent_timeline.py > gpurun_out/r04bi_timeline.txt 2>&1; sed -n 24,36p gpurun_out/r04bi_timeline.txt
