#!/bin/bash
mkdir -p gpurun_out
STEPS=40 bash benchmarks/ab_multi.sh "REFTR_LOSS_DIRECT=0" "REFTR_LOSS_DIRECT=1" "REFTR_LOSS_DIRECT=2" "REFTR_LOSS_DIRECT=3" "REFTR_LOSS_DIRECT=4" > gpurun_out/r04at_ab.txt 2>&1
cat gpurun_out/r04at_ab.txt
