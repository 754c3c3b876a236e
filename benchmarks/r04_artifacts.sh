# Round 4: every judged measurement artefact of the final build in one GPU session (build_id binds the PMC traffic file to these sources).
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; mkdir -p $O
bash benchmarks/round_artifacts.sh r04 2>&1 | cut -c1-300
timeout 600 python tools/concurrent_timeline.py --out $O/r04_concurrent_timeline.txt > $O/r04_timeline.log 2>&1; echo "timeline rc $?"
timeout 600 python benchmarks/step_breakdown.py > $O/r04_shape_breakdown.txt 2>&1; echo "breakdown rc $?"
bash benchmarks/milestones.sh r04 > $O/r04_milestones.log 2>&1; echo "milestones rc $?"
( timeout 300 python benchmarks/cfg4_seg.py 2>&1 | grep hipgraph; timeout 300 python benchmarks/cfg5_stress.py 2>&1 | grep hipgraph; timeout 300 python benchmarks/eval_throughput.py 2>&1 | grep hipgraph ) > $O/r04_other_configs.txt; cat $O/r04_other_configs.txt
timeout 600 python benchmarks/epoch_throughput.py 2>&1 | tail -1 > $O/r04_epoch_throughput.json; cat $O/r04_epoch_throughput.json | cut -c1-400
timeout 600 python benchmarks/wgrad_group_bench.py > $O/r04_wgrad_group_bench.txt 2>&1; tail -6 $O/r04_wgrad_group_bench.txt
FLUSH=1 ONLY=lin HINTS=0,31,33,51,21 timeout 600 python benchmarks/tile_sweep.py > $O/r04_tile_sweep_cold.txt 2>&1; head -14 $O/r04_tile_sweep_cold.txt
