#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_seg_gpu.py -x -q -m gpu -k "visualize or evaluate" > gpurun_out/r04be_tests.log 2>&1; echo "rc $?"; tail -4 gpurun_out/r04be_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py 2>&1 | tail -1 | cut -c1-200
