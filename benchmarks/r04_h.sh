mkdir -p gpurun_out
export TMPDIR=/tmp
python benchmarks/debug_seg_floor.py > gpurun_out/r04h_seg_floor_stages.txt 2>&1
python benchmarks/debug_stem_floor.py >> gpurun_out/r04h_seg_floor_stages.txt 2>&1
timeout 3000 python -m pytest tests -m gpu -q -x > gpurun_out/r04h_gpu_tests.log 2>&1; echo "gpu tests rc $?"
tail -8 gpurun_out/r04h_gpu_tests.log
