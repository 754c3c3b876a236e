#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_ops_gpu.py tests/test_parity_fullsize_gpu.py tests/test_failsafe_gpu.py tests/test_seg_gpu.py tests/test_fullsize_gpu.py tests/test_edge_cases_gpu.py -x -q -m gpu > gpurun_out/r04ap_tests.log 2>&1; echo "rc $?"; tail -5 gpurun_out/r04ap_tests.log
