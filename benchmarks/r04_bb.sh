#!/bin/bash
mkdir -p gpurun_out
STEPS=50 bash benchmarks/ab_multi.sh "REFTR_WG_SIDE=4" "REFTR_WG_SIDE=0" "REFTR_WG_SIDE=4 REFTR_STEM_FIRST=2" > gpurun_out/r04bb_ab.txt 2>&1
cat gpurun_out/r04bb_ab.txt
