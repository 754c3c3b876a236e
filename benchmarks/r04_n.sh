mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gemm_gpu.py -x -q > gpurun_out/r04n_gemm.log 2>&1; echo "gemm rc $?"; tail -3 gpurun_out/r04n_gemm.log
VAR=REFTR_GEMM_TOUCH VALS="0 1 2" timeout 900 bash benchmarks/ab_env.sh > gpurun_out/r04n_ab_touch.txt 2>&1; cat gpurun_out/r04n_ab_touch.txt
for tv in 0 1; do echo "REFTR_GEMM_TOUCH=$tv cold sweep"; REFTR_GEMM_TOUCH=$tv FLUSH=1 ONLY=lin HINTS=0 timeout 600 python benchmarks/tile_sweep.py 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r04n_touch_cold_sweep.txt 2>&1; cat gpurun_out/r04n_touch_cold_sweep.txt
