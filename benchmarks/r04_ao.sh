#!/bin/bash
mkdir -p gpurun_out
REFTR_STEM_FIRST=2 python tools/concurrent_timeline.py > gpurun_out/r04ao_timeline_l1_first.txt 2>&1
STEPS=60 bash benchmarks/ab_multi.sh "REFTR_STEM_FIRST=1" "REFTR_STEM_FIRST=2" "REFTR_STEM_FIRST=0" > gpurun_out/r04ao_ab.txt 2>&1
STEPS=60 bash benchmarks/ab_multi.sh "REFTR_STEM_FIRST=2" "REFTR_STEM_FIRST=0" "REFTR_STEM_FIRST=1" >> gpurun_out/r04ao_ab.txt 2>&1
sed -n 3,36p gpurun_out/r04ao_timeline_l1_first.txt; cat gpurun_out/r04ao_ab.txt
