mkdir -p gpurun_out
export TMPDIR=/tmp
(echo "N = 2 control flow on one GPU over gloo (functional, not a measurement), final build:"; bash benchmarks/dp_smoke_gloo.sh bf16; bash benchmarks/dp_smoke_gloo.sh fp32) > gpurun_out/r04ag_dp_world2_gloo_smoke.txt 2>&1; tail -12 gpurun_out/r04ag_dp_world2_gloo_smoke.txt | cut -c1-400
bash benchmarks/ab_ddp.sh > gpurun_out/r04ag_ddp_single_rank.txt 2>&1; cat gpurun_out/r04ag_ddp_single_rank.txt
