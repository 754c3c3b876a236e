# Regenerates the judged measurement artefacts of the current build on the GPU box:
#   bash benchmarks/round_artifacts.sh r01j
# -> gpurun_out/$TAG_bench.json, $TAG_kernel_stats_graph.md, $TAG_pmc_traffic.json, $TAG_pmc_kernels.txt, $TAG_phases.txt, $TAG_pmc_gemm_stalls.txt
TAG=${1:-rXX}; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; mkdir -p $O
python bench.py > $O/${TAG}_bench.log 2>&1; tail -1 $O/${TAG}_bench.log > $O/${TAG}_bench.json; cut -c1-400 $O/${TAG}_bench.json
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/${TAG}_stats -- python $R/bench.py --no-cpu-baseline > $O/${TAG}_stats.log 2>&1
cd $R
DB=$(find $O/${TAG}_stats -name "*.db" | head -1)
python tools/rocpd_stats.py $DB > $O/${TAG}_kernel_stats_graph.md 2>$O/${TAG}_stats.err; head -12 $O/${TAG}_kernel_stats_graph.md
cd /tmp
REFTR_STREAMS=0 timeout 600 rocprofv3 --kernel-trace -d $O/${TAG}_trace1 -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-kernel-roofline > $O/${TAG}_trace1.log 2>&1
cd $R
DB1=$(find $O/${TAG}_trace1 -name "*.db" | head -1)
python tools/step_phases.py $DB1 > $O/${TAG}_phases.txt 2>&1; tail -16 $O/${TAG}_phases.txt
rm -rf $O/${TAG}_trace1
bash benchmarks/pmc_passes.sh > $O/${TAG}_pmc.log 2>&1
F=$(find $O/pmc_FETCH_SIZE -name "*.db" | head -1); W=$(find $O/pmc_WRITE_SIZE -name "*.db" | head -1)
python tools/pmc_traffic.py $F $W --steps 5 --bench-json $O/${TAG}_bench.json --stats-md $O/${TAG}_kernel_stats_graph.md --json $O/${TAG}_pmc_traffic.json > $O/${TAG}_pmc_kernels.txt 2>&1; head -14 $O/${TAG}_pmc_kernels.txt
rm -rf $O/${TAG}_stats $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
# MFMA utilisation / stall breakdown of the step's representative GEMM shapes (SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES, waits): its own pass
bash benchmarks/pmc_kernels.sh > $O/${TAG}_pmc_gemm_stalls.txt 2>&1; tail -14 $O/${TAG}_pmc_gemm_stalls.txt | cut -c1-150
rm -rf $O/pmc_k
