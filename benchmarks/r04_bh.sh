#!/bin/bash
mkdir -p gpurun_out
STEPS=50 bash benchmarks/ab_multi.sh "REFTR_PREP_IN_LANG=0" "REFTR_PREP_IN_LANG=1" > gpurun_out/r04bh_ab.txt 2>&1
STEPS=50 bash benchmarks/ab_multi.sh "REFTR_PREP_IN_LANG=1" "REFTR_PREP_IN_LANG=0" >> gpurun_out/r04bh_ab.txt 2>&1
cat gpurun_out/r04bh_ab.txt
