export REFTR_LAB=1   # kernel tuning switches live in the lab library only (benchmarks/README.md): build it with REFTR_LAB=1 first
cd benchmarks
for a in 0 1 2 4 3 5 6; do echo "== REFTR_GEMM_ABL=$a"; REFTR_GEMM_ABL=$a ONLY=conv HINTS=0 python tile_sweep.py 2>&1 | grep -v "^hints\|amdgpu" | cut -c1-36 | tr '\n' ';'; echo; done
