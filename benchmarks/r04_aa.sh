mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gemm_gpu.py -x -q > gpurun_out/r04aa_gemm.log 2>&1; echo "gemm rc $?"; tail -4 gpurun_out/r04aa_gemm.log
FLUSH=1 HINTS=0,31,51,71,72,73,74 timeout 1200 python benchmarks/tile_sweep.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04aa_wide_tiles_cold.txt; cat gpurun_out/r04aa_wide_tiles_cold.txt
