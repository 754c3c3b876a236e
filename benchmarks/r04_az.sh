#!/bin/bash
# backward of map_sentence / map_phrase at the head of the BERT-backward branch (language stream): tests + interleaved A/B
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "golden or captured_step or direct_loss" > gpurun_out/r04az_tests.log 2>&1; echo "rc $?"; tail -3 gpurun_out/r04az_tests.log
STEPS=50 bash benchmarks/ab_multi.sh "REFTR_LANG_TAIL=0" "REFTR_LANG_TAIL=1" > gpurun_out/r04az_ab.txt 2>&1
STEPS=50 bash benchmarks/ab_multi.sh "REFTR_LANG_TAIL=1" "REFTR_LANG_TAIL=0" >> gpurun_out/r04az_ab.txt 2>&1
cat gpurun_out/r04az_ab.txt
python tools/concurrent_timeline.py > gpurun_out/r04az_timeline.txt 2>&1; sed -n 3,36p gpurun_out/r04az_timeline.txt
