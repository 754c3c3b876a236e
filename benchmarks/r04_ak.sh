mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gemm_gpu.py -x -q -k "tiled_im2col" > gpurun_out/r04ak_c3.log 2>&1; echo "c3 rc $?"; tail -15 gpurun_out/r04ak_c3.log
FLUSH=1 ONLY=conv HINTS=0,908,916 timeout 600 python benchmarks/tile_sweep.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04ak_c3_cold.txt; cat gpurun_out/r04ak_c3_cold.txt

