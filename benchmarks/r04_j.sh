mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_dp_rccl_gpu.py tests/test_failsafe_gpu.py -x -q > gpurun_out/r04j_dp.log 2>&1; echo "dp rc $?"
grep -n "assert\|Error\|passed\|failed" gpurun_out/r04j_dp.log | head -20
for r in 1 2; do
for tw in 0 1; do
echo -n "REFTR_DDP_FORCE=1 interleave REFTR_DDP_TWIN=$tw  "; REFTR_DDP_TWIN=$tw REFTR_DDP_FORCE=1 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-roofline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), d['config'].get('launch'), d.get('rccl',{}).get('exposed_exchange_ms'))"
done
echo -n "REFTR_DDP_FORCE=1 REFTR_COMM=abi TWIN=1           "; REFTR_COMM=abi REFTR_DDP_FORCE=1 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-roofline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), d['config'].get('launch'))"
echo -n "N=1 schedule                                      "; python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-roofline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), d['config'].get('launch'))"
done
