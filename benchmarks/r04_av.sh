#!/bin/bash
mkdir -p gpurun_out
timeout 900 python benchmarks/cfg_dc5.py > gpurun_out/r04av_dc5.txt 2>&1; echo "rc $?"; tail -5 gpurun_out/r04av_dc5.txt
