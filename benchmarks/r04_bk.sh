#!/bin/bash
# last confirmation of the committed tree: fail-safe + captured-step tests, smoke, the default bench line
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_failsafe_gpu.py tests/test_decoder_coop_gpu.py -x -q -m gpu > gpurun_out/r04bk_tests.log 2>&1; echo "rc $?"; tail -2 gpurun_out/r04bk_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > gpurun_out/r04bk_bench.log 2>&1; tail -1 gpurun_out/r04bk_bench.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['build_id'], d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline'])"
