# BERT-sized (M = 320 rows) Linear weight gradients on the first-generation grouped kernel vs the v2 kernel
cd benchmarks
for sm in 0 512; do echo "== REFTR_W2_MINM=256 SMALLM=$sm"; ONLY=lin REFTR_W2_SMALLM=$sm REFTR_W2_MINM=256 python wgrad_group_bench.py 2>&1 | tail -4; done
