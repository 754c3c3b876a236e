#!/bin/bash
# bounded residency of the AdamW pass: step A/B + timeline of the best-looking arm
mkdir -p gpurun_out
STEPS=50 bash benchmarks/ab_multi.sh "REFTR_ADAMW_LDS_KB=0" "REFTR_ADAMW_LDS_KB=20" "REFTR_ADAMW_LDS_KB=40" "REFTR_ADAMW_LDS_KB=80" > gpurun_out/r04aw_ab.txt 2>&1
cat gpurun_out/r04aw_ab.txt
for v in 40 80; do REFTR_ADAMW_LDS_KB=$v python tools/concurrent_timeline.py > gpurun_out/r04aw_timeline_$v.txt 2>&1; echo "== $v"; sed -n 3,20p gpurun_out/r04aw_timeline_$v.txt; done
