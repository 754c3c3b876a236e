cd benchmarks
for v in 0 1 0 1; do echo "== REFTR_EPI_PREFETCH=$v"; REFTR_EPI_PREFETCH=$v ONLY=lin HINTS=0 python tile_sweep.py 2>&1 | grep -v "^hints\|amdgpu" | cut -c1-40 | tr '\n' ';'; echo; done
