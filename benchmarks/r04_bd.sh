#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
REFTR_WG_MERGE_DEC=1 timeout 600 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "golden or captured_step" > gpurun_out/r04bd_tests.log 2>&1; echo "rc $?"; tail -3 gpurun_out/r04bd_tests.log
STEPS=50 bash benchmarks/ab_multi.sh "REFTR_WG_MERGE_DEC=0" "REFTR_WG_MERGE_DEC=1" > gpurun_out/r04bd_ab.txt 2>&1
STEPS=50 bash benchmarks/ab_multi.sh "REFTR_WG_MERGE_DEC=1" "REFTR_WG_MERGE_DEC=0" >> gpurun_out/r04bd_ab.txt 2>&1
cat gpurun_out/r04bd_ab.txt
