# the data-parallel schedules on the one-GPU box (single-rank RCCL): compute cost of each exchange schedule vs the N = 1 step
for r in 1 2; do
echo -n "N=1 schedule                      "; python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-roofline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), d['config'].get('launch'))"
echo -n "REFTR_DDP_FORCE=1 interleave      "; REFTR_DDP_FORCE=1 REFTR_DDP_SCHEDULE=interleave python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-roofline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), d['config'].get('launch'))"
echo -n "REFTR_DDP_FORCE=1 serial          "; REFTR_DDP_FORCE=1 REFTR_DDP_SCHEDULE=serial python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-roofline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), d['config'].get('launch'))"
done
