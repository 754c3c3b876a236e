cd benchmarks
for x in 0 1; do echo "== REFTR_W2_XCD=$x"; ONLY=conv REFTR_W2_XCD=$x python wgrad_group_bench.py 2>&1 | tail -4; done
cd ..; REFTR_W2_XCD=1 timeout 300 python -m pytest tests/test_gemm_gpu.py -q -m gpu -k "wgrad" 2>&1 | tail -2
MODES="1" bash benchmarks/pmc_w2_fetch.sh 2>&1 | tail -14
