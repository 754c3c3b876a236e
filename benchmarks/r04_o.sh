mkdir -p gpurun_out
export TMPDIR=/tmp
# deep-stage small-tile forms on the language-branch / encoder Linears, cold caches
FLUSH=1 ONLY=lin HINTS=0,281,285,286,287,288,33,234,236 timeout 900 python benchmarks/tile_sweep.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04o_deep_stage_cold.txt; cat gpurun_out/r04o_deep_stage_cold.txt
FLUSH=0 ONLY=lin HINTS=0,281,285,286,287,288,33,234,236 timeout 900 python benchmarks/tile_sweep.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04o_deep_stage_warm.txt; cat gpurun_out/r04o_deep_stage_warm.txt
