export REFTR_LAB=1   # kernel tuning switches live in the lab library only (benchmarks/README.md): build it with REFTR_LAB=1 first
cd benchmarks
for x in 0 2 0 2; do echo "== REFTR_W2_XCD=$x"; REFTR_W2_XCD=$x python wgrad_group_bench.py 2>&1 | tail -7 | cut -c1-72 | tr '\n' ';'; echo; done
cd ..; REFTR_W2_XCD=2 timeout 300 python -m pytest tests/test_gemm_gpu.py -q -m gpu -k "wgrad" 2>&1 | tail -1
MODES="0 2" bash benchmarks/pmc_w2_fetch.sh 2>&1 | grep "grouped\|==" | head -16
