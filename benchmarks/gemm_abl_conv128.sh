export REFTR_LAB=1   # kernel tuning switches live in the lab library only (benchmarks/README.md): build it with REFTR_LAB=1 first
# where a K step of the 128 x 128 tiles goes on the 3 x 3 convolutions (one workgroup per CU at B = 8): the same launches with parts removed
# (REFTR_GEMM_ABL: 1 no operand DMA, 2 no MFMA + no fragment reads, 4 no epilogue; wrong results on purpose).  hints 51 = 2 stages, 252 = 3 stages pipelined
cd benchmarks
for a in 0 1 2 4 3 5 6 0; do echo "== REFTR_GEMM_ABL=$a"; REFTR_GEMM_ABL=$a ONLY=conv HINTS=51,252,21,233 python tile_sweep.py 2>&1 | grep "l3 3x3\|l4 3x3 512 @20\|l2 3x3 128 @80" | cut -c1-64; done
