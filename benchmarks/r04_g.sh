mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_parity_fullsize_gpu.py -q -s -k "functional or order_floor" > gpurun_out/r04g_parity.log 2>&1; echo "parity rc $?"
grep -n "^\[cfg\|gates\|mask decisions\|passed\|failed\|^E " gpurun_out/r04g_parity.log | cut -c1-700 | head -40
