#!/bin/bash
# fused stem: where the step's time goes with and without it (concurrent timeline per arm) + a longer interleaved A/B
mkdir -p gpurun_out
for v in 0 1; do REFTR_STEM_FUSE=$v python tools/concurrent_timeline.py > gpurun_out/r04am_timeline_fuse$v.txt 2>&1; done
for r in 1 2 3; do for v in 1 0; do echo -n "REFTR_STEM_FUSE=$v  "; REFTR_STEM_FUSE=$v python bench.py --no-cpu-baseline --no-kernel-roofline --steps 60 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'img/s', round(d['ms_per_step'],3), 'ms  median', round(d.get('ms_per_step_median',0),3))"; done; done > gpurun_out/r04am_ab.txt 2>&1
for v in 0 1; do echo "== REFTR_STEM_FUSE=$v"; sed -n 3,20p gpurun_out/r04am_timeline_fuse$v.txt; done; cat gpurun_out/r04am_ab.txt
