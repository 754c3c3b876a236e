P='import sys,json
for l in sys.stdin:
    if l.startswith("{\"metric\""):
        d=json.loads(l); print(round(d["ms_per_step"],3), round(d["value"],1), d["config"]["launch"], d["loss"])
    elif "bench]" in l: print(l.strip()[:300])'
echo "plain:"; timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-roofline 2>&1 | python -c "$P"
for ph in "bert" "main,bert" "main,bert,layer4" "main,bert_hi,bert_mid,bert,layer4"; do
echo "single-rank RCCL, graphs, exchange at: $ph"; REFTR_DDP_PHASES=$ph REFTR_DDP_FORCE=1 timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-roofline 2>&1 | python -c "$P"
done
echo "single-rank RCCL, eager, all boundaries:"; REFTR_DDP_FORCE=1 timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-kernel-roofline --no-graph 2>&1 | python -c "$P"
