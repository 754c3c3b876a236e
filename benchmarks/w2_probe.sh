export REFTR_LAB=1   # kernel tuning switches live in the lab library only (benchmarks/README.md): build it with REFTR_LAB=1 first
# ablation / variant probe of the second-generation weight-gradient kernel on the ResNet stage groups
for v in "REFTR_W2_ABL=0" "REFTR_W2_ABL=3" "REFTR_W2_TARGET=512" "REFTR_W2_TARGET=384" "REFTR_W2_TARGET=768" "REFTR_W2_XCD=1" "REFTR_W2_PF=2"; do
  echo "== $v"; env $v ONLY=conv python benchmarks/wgrad_group_bench.py 2>&1 | grep -v amdgpu.ids
done
echo "== all groups, default"; python benchmarks/wgrad_group_bench.py 2>&1 | grep -v amdgpu.ids
echo "== all groups, MINM=100000 (linears on v1)"; REFTR_W2_MINM=100000 python benchmarks/wgrad_group_bench.py 2>&1 | grep -v amdgpu.ids
