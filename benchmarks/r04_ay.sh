#!/bin/bash
mkdir -p gpurun_out
STEPS=50 bash benchmarks/ab_multi.sh "REFTR_OPT_SERIAL=0" "REFTR_OPT_SERIAL=1" "REFTR_OPT_SERIAL=1 REFTR_STEM_FIRST=0" > gpurun_out/r04ay_ab.txt 2>&1
cat gpurun_out/r04ay_ab.txt
REFTR_OPT_SERIAL=1 python tools/concurrent_timeline.py > gpurun_out/r04ay_timeline_serial.txt 2>&1; sed -n 3,22p gpurun_out/r04ay_timeline_serial.txt
