#!/bin/bash
# data-parallel control flow on the last build: two gloo ranks on the one GPU (both exchange dtypes) + the single-rank RCCL schedules
mkdir -p gpurun_out
( bash benchmarks/dp_smoke_gloo.sh fp32; bash benchmarks/dp_smoke_gloo.sh bf16 ) > gpurun_out/r04bg_dp_world2_gloo_smoke.txt 2>&1
tail -3 gpurun_out/r04bg_dp_world2_gloo_smoke.txt | cut -c1-300
ls benchmarks | grep -i "ddp\|single_rank" | head
