mkdir -p gpurun_out
export TMPDIR=/tmp
for v in 0 1; do echo "REFTR_TAPS_INNER=$v"; REFTR_TAPS_INNER=$v FLUSH=1 ONLY=conv HINTS=0,51,21,31,233,252 timeout 600 python benchmarks/tile_sweep.py 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r04aj_taps_inner_cold.txt; cat gpurun_out/r04aj_taps_inner_cold.txt
REFTR_TAPS_INNER=1 timeout 900 python -m pytest tests/test_gemm_gpu.py -x -q -k "conv" > gpurun_out/r04aj_gemm.log 2>&1; echo "gemm rc $?"; tail -3 gpurun_out/r04aj_gemm.log
STEPS=40 timeout 1200 bash benchmarks/ab_multi.sh "REFTR_TAPS_INNER=0" "REFTR_TAPS_INNER=1" > gpurun_out/r04aj_ab.txt 2>&1; cat gpurun_out/r04aj_ab.txt
