export REFTR_LAB=1 TMPDIR=/tmp
cd benchmarks
for a in 0 4 1 2 5 0; do echo "== REFTR_GEMM_ABL=$a (cold)"; REFTR_GEMM_ABL=$a FLUSH=1 ONLY=lin HINTS=0,31,51,33 python tile_sweep.py 2>&1 | grep "hints\|l3 256->1024\|l2 128->512\|l4 512->2048\|enc 256->2048\|l3 1024->256" | cut -c1-70; done
