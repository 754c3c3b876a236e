#!/bin/bash
mkdir -p gpurun_out
STEPS=50 bash benchmarks/ab_multi.sh "REFTR_LANG_TAIL=1" "REFTR_LANG_TAIL=2" "REFTR_LANG_TAIL=0" > gpurun_out/r04ba_ab.txt 2>&1
STEPS=50 bash benchmarks/ab_multi.sh "REFTR_LANG_TAIL=2" "REFTR_LANG_TAIL=1" >> gpurun_out/r04ba_ab.txt 2>&1
cat gpurun_out/r04ba_ab.txt
