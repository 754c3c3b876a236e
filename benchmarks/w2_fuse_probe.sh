export REFTR_LAB=1   # kernel tuning switches live in the lab library only (benchmarks/README.md): build it with REFTR_LAB=1 first
cd benchmarks
for w in "4,2,2,1,3" "4,2.25,2.25,1.5,4"; do echo "== WTS=$w"; REFTR_W2_WTS=$w python wgrad_group_bench.py 2>&1 | tail -7 | cut -c1-70 | tr '\n' ';'; echo; done
cd ..
for r in 1 2; do
echo -n "old policy (no fusion, old weights)  "; REFTR_W2_FUSE3=0 REFTR_W2_FIT=0 REFTR_W2_WTS=4,2,2,1,3 python bench.py --no-cpu-baseline --no-kernel-roofline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],3))"
echo -n "fusion + calibrated weights          "; REFTR_W2_WTS=4,2.25,2.25,1.5,4 python bench.py --no-cpu-baseline --no-kernel-roofline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],3))"
echo -n "fusion + old weights                 "; REFTR_W2_WTS=4,2,2,1,3 python bench.py --no-cpu-baseline --no-kernel-roofline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],3))"
done
