# usage: bash benchmarks/ab_lib.sh <alt .so>   -- interleaved bench.py runs of the tree's library (A) against another build of it (B)
A=reftr_amd/libreftr_hip.so; cp $A /tmp/libA.so; cp $1 /tmp/libB.so
run() { python bench.py --no-cpu-baseline --no-kernel-roofline --steps ${STEPS:-40} 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'img/s', round(d['ms_per_step'],3), 'ms  median', round(d.get('ms_per_step_median',0),3))"; }
for r in 1 2; do echo -n "A (tree)  "; cp /tmp/libA.so $A; run; echo -n "B ($1)  "; cp /tmp/libB.so $A; run; done
cp /tmp/libA.so $A
