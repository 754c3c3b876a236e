mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -x -q -m gpu > gpurun_out/r04ae_gpu_tests.log 2>&1; echo "gpu tests rc $?"; tail -4 gpurun_out/r04ae_gpu_tests.log
FLUSH=1 ONLY=conv HINTS=0 timeout 600 python benchmarks/tile_sweep.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04ae_conv_cold.txt; cat gpurun_out/r04ae_conv_cold.txt
for i in 1 2; do python bench.py --no-cpu-baseline --no-kernel-roofline --steps 40 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'img/s', round(d['ms_per_step'],3), 'ms')"; done
