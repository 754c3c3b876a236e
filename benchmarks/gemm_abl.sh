export REFTR_LAB=1   # kernel tuning switches live in the lab library only (benchmarks/README.md): build it with REFTR_LAB=1 first
# Where the time of the dense 64x64-tile products goes: the same launches with parts of the kernel removed
cd benchmarks
for a in 0 1 2 4 3 5 6; do echo "== REFTR_GEMM_ABL=$a (1 no loads, 2 no MFMA, 4 no epilogue)"; REFTR_GEMM_ABL=$a ONLY=lin HINTS=31 python tile_sweep.py 2>&1 | grep -v "^hints\|amdgpu" | cut -c1-40 | tr '\n' ';'; echo; done
