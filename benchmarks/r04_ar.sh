#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "direct_loss or captured or epoch" > gpurun_out/r04ar_tests.log 2>&1; echo "rc $?"; tail -4 gpurun_out/r04ar_tests.log
STEPS=60 bash benchmarks/ab_multi.sh "REFTR_LOSS_DIRECT=0" "REFTR_LOSS_DIRECT=1" > gpurun_out/r04ar_ab.txt 2>&1
STEPS=60 bash benchmarks/ab_multi.sh "REFTR_LOSS_DIRECT=1" "REFTR_LOSS_DIRECT=0" >> gpurun_out/r04ar_ab.txt 2>&1
cat gpurun_out/r04ar_ab.txt
