# Round 4: every judged measurement artefact of the final build in one GPU session (build_id binds the PMC traffic file to these sources).
TAG=${1:-r04b}
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; mkdir -p $O
bash benchmarks/round_artifacts.sh ${TAG} 2>&1 | cut -c1-300
timeout 600 python tools/concurrent_timeline.py --out $O/${TAG}_concurrent_timeline.txt > $O/${TAG}_timeline.log 2>&1; echo "timeline rc $?"
timeout 600 python benchmarks/step_breakdown.py > $O/${TAG}_shape_breakdown.txt 2>&1; echo "breakdown rc $?"
bash benchmarks/milestones.sh ${TAG} > $O/${TAG}_milestones.log 2>&1; echo "milestones rc $?"
( timeout 300 python benchmarks/cfg4_seg.py 2>&1 | grep hipgraph; timeout 300 python benchmarks/cfg5_stress.py 2>&1 | grep hipgraph; timeout 300 python benchmarks/eval_throughput.py 2>&1 | grep hipgraph ) > $O/${TAG}_other_configs.txt; cat $O/${TAG}_other_configs.txt
timeout 600 python benchmarks/epoch_throughput.py 2>&1 | tail -1 > $O/${TAG}_epoch_throughput.json; cat $O/${TAG}_epoch_throughput.json | cut -c1-400
timeout 600 python benchmarks/wgrad_group_bench.py > $O/${TAG}_wgrad_group_bench.txt 2>&1; tail -6 $O/${TAG}_wgrad_group_bench.txt
FLUSH=1 ONLY=lin HINTS=0,31,33,51,21 timeout 600 python benchmarks/tile_sweep.py > $O/${TAG}_tile_sweep_cold.txt 2>&1; head -14 $O/${TAG}_tile_sweep_cold.txt
