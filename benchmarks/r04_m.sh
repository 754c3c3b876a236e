export TMPDIR=/tmp
for i in 1 2; do python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['ms_per_step'],3), 'family ms', round(r['kernel_ms_per_step'],3), 'TF', round(r['achieved'],1), 'avg us', round(r['avg_launch_us'],2), 'traffic', r['traffic'], r['traffic_note'][:80])"; done
REFTR_FUSED_NORM=0 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('FUSED_NORM=0', round(d['ms_per_step'],3), 'family ms', round(r['kernel_ms_per_step'],3), 'TF', round(r['achieved'],1), 'avg us', round(r['avg_launch_us'],2))"
REFTR_OPT_EMIT=0 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('OPT_EMIT=0', round(d['ms_per_step'],3), 'family ms', round(r['kernel_ms_per_step'],3), 'TF', round(r['achieved'],1), 'avg us', round(r['avg_launch_us'],2))"
