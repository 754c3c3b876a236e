#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "dilation" -s > gpurun_out/r04au_tests.log 2>&1; echo "rc $?"; grep -A8 "dilation 480" gpurun_out/r04au_tests.log; tail -5 gpurun_out/r04au_tests.log


