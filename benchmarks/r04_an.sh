#!/bin/bash
# fused stem, language branch forked behind it: timeline + interleaved A/B of the three arms
mkdir -p gpurun_out
REFTR_STEM_FIRST=1 python tools/concurrent_timeline.py > gpurun_out/r04an_timeline_stem_first.txt 2>&1
STEPS=60 bash benchmarks/ab_multi.sh "REFTR_STEM_FUSE=0" "REFTR_STEM_FUSE=1" "REFTR_STEM_FUSE=1 REFTR_STEM_FIRST=1" > gpurun_out/r04an_ab.txt 2>&1
STEPS=60 bash benchmarks/ab_multi.sh "REFTR_STEM_FUSE=1 REFTR_STEM_FIRST=1" "REFTR_STEM_FUSE=1" "REFTR_STEM_FUSE=0" >> gpurun_out/r04an_ab.txt 2>&1
sed -n 3,22p gpurun_out/r04an_timeline_stem_first.txt; cat gpurun_out/r04an_ab.txt
