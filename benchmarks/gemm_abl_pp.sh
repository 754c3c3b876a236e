export REFTR_LAB=1   # lab library (build it with REFTR_LAB=1 first)
# where a K tile of the ping-pong 128 x 128 form (hint 351) goes on the 3 x 3 convolutions: the same launches with parts removed
# (REFTR_GEMM_ABL bits: 1 no operand DMA, 2 no MFMAs, 4 no epilogue, 8 no fragment reads; wrong results on purpose)
cd benchmarks
for a in ${ABLS:-0 1 2 8 10 9 3 11 4 15 0}; do echo "== REFTR_GEMM_ABL=$a"; REFTR_GEMM_ABL=$a ONLY=conv HINTS=${HINTS:-351,331} python tile_sweep.py 2>&1 | grep "l3 3x3\|l4 3x3 512 @20\|l2 3x3 128 @80" | cut -c1-64; done
