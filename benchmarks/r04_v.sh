mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gemm_gpu.py -x -q > gpurun_out/r04v_gemm.log 2>&1; echo "gemm rc $?"; tail -4 gpurun_out/r04v_gemm.log
FLUSH=1 HINTS=0,31,331,33,333,21,321,51,351 timeout 1200 python benchmarks/tile_sweep.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04v_direct_epilogue_cold.txt; cat gpurun_out/r04v_direct_epilogue_cold.txt
