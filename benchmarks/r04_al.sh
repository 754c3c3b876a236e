#!/bin/bash
# fused stem + max-pool: parity, cold/warm kernel time, step A/B
mkdir -p gpurun_out
python -m pytest tests/test_ops_gpu.py -q -k "stem" -x 2>&1 | tail -5 > gpurun_out/r04al_tests.log
timeout 300 python benchmarks/stem_bench.py > gpurun_out/r04al_stem_bench.txt 2>&1
VAR=REFTR_STEM_FUSE VALS="0 1" bash benchmarks/ab_env.sh > gpurun_out/r04al_ab.txt 2>&1
tail -3 gpurun_out/r04al_tests.log; cat gpurun_out/r04al_stem_bench.txt; tail -12 gpurun_out/r04al_ab.txt
