mkdir -p gpurun_out
export TMPDIR=/tmp
VAR=REFTR_GROUP_CONV VALS="1 2" timeout 600 bash benchmarks/ab_env.sh > gpurun_out/r04k_ab_groupconv.txt 2>&1; cat gpurun_out/r04k_ab_groupconv.txt
VAR=REFTR_WG_SIDE VALS="4 5 7" timeout 600 bash benchmarks/ab_env.sh > gpurun_out/r04k_ab_wgside.txt 2>&1; cat gpurun_out/r04k_ab_wgside.txt
