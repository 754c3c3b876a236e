#!/bin/bash
mkdir -p gpurun_out
STEPS=50 bash benchmarks/ab_multi.sh "REFTR_HEAD_LEAF_SIDE=0" "REFTR_HEAD_LEAF_SIDE=1" > gpurun_out/r04bf_ab.txt 2>&1
STEPS=50 bash benchmarks/ab_multi.sh "REFTR_HEAD_LEAF_SIDE=1" "REFTR_HEAD_LEAF_SIDE=0" >> gpurun_out/r04bf_ab.txt 2>&1
cat gpurun_out/r04bf_ab.txt
