#!/bin/bash
mkdir -p gpurun_out
for v in 1 0; do REFTR_LOSS_DIRECT=$v python tools/concurrent_timeline.py > gpurun_out/r04as_timeline_direct$v.txt 2>&1; done
for v in 1 0; do echo "== REFTR_LOSS_DIRECT=$v"; sed -n 3,36p gpurun_out/r04as_timeline_direct$v.txt; done
