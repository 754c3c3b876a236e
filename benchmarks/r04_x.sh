mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gemm_gpu.py -x -q -k "split or linear" > gpurun_out/r04x_gemm.log 2>&1; echo "gemm rc $?"; tail -12 gpurun_out/r04x_gemm.log
SH="enc 2048|bert 3072|bert 2304|l4 2048|l3 1024|big"
FLUSH=1 ONLY=lin HINTS=0,33,282,285 KSPLITS=1,2,3,4,6,8 timeout 1200 python benchmarks/tile_sweep.py 2>&1 | grep -E "hints|$SH" > gpurun_out/r04x_ksplit_cold.txt; cat gpurun_out/r04x_ksplit_cold.txt
