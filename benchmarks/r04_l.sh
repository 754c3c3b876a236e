mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_encoder_fused_gpu.py -x -q > gpurun_out/r04l_enc.log 2>&1; echo "enc rc $?"
grep -n "assert\|Error\|passed\|failed" gpurun_out/r04l_enc.log | head -20
VAR=REFTR_ENC_FUSE VALS="0 2" timeout 600 bash benchmarks/ab_env.sh > gpurun_out/r04l_ab_enc2.txt 2>&1; cat gpurun_out/r04l_ab_enc2.txt
