#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "golden or captured_step or direct_loss or dropout" > gpurun_out/r04ax_tests.log 2>&1; echo "rc $?"; tail -3 gpurun_out/r04ax_tests.log
STEPS=50 bash benchmarks/ab_multi.sh "REFTR_LANG_TAIL=0" "REFTR_LANG_TAIL=1" > gpurun_out/r04ax_ab.txt 2>&1
STEPS=50 bash benchmarks/ab_multi.sh "REFTR_LANG_TAIL=1" "REFTR_LANG_TAIL=0" >> gpurun_out/r04ax_ab.txt 2>&1
cat gpurun_out/r04ax_ab.txt
STEPS=50 bash benchmarks/ab_multi.sh "REFTR_ADAMW_LDS_KB=0" "REFTR_ADAMW_LDS_KB=20" "REFTR_ADAMW_LDS_KB=40" "REFTR_ADAMW_LDS_KB=80" > gpurun_out/r04aw_ab.txt 2>&1
cat gpurun_out/r04aw_ab.txt
for v in 40; do REFTR_ADAMW_LDS_KB=$v python tools/concurrent_timeline.py > gpurun_out/r04aw_timeline_$v.txt 2>&1; echo "== $v"; sed -n 3,22p gpurun_out/r04aw_timeline_$v.txt; done
