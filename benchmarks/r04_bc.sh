#!/bin/bash
mkdir -p gpurun_out
STEPS=50 bash benchmarks/ab_multi.sh "REFTR_WG_SIDE=4" "REFTR_WG_SIDE=5" "REFTR_WG_SIDE=7" "REFTR_WG_SIDE=6" > gpurun_out/r04bc_ab.txt 2>&1
cat gpurun_out/r04bc_ab.txt
