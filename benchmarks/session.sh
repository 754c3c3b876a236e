#!/bin/bash
# ONE GPU-box session = a list of named steps (replaces the 63 one-off r04_*.sh scripts of round 4; their outputs live on under profiles/).
#   gpurun -- 'bash benchmarks/session.sh TAG step [step ...]'
# steps:
#   tests            the full GPU suite                      -> gpurun_out/TAG_gpu_tests.log
#   bench            default bench.py line                   -> gpurun_out/TAG_bench.json
#   artifacts        bench + rocprofv3 kernel stats + phases + PMC traffic (benchmarks/round_artifacts.sh)
#   timeline         concurrent two-stream timeline of the replayed step (tools/concurrent_timeline.py)
#   region           kernel sequence of the few-row region + stage stamps of its three launches
#   shapes           every GEMM-family launch of one step against its roofline (benchmarks/step_breakdown.py)
#   wgrad            the step's weight-gradient groups alone (benchmarks/wgrad_group_bench.py)
#   tiles            cold tile sweep of the Linear / conv shapes (benchmarks/tile_sweep.py)
#   configs          configs[3] (RefTRSeg), configs[4] (R101 800x800 16 phrases), inference throughput
#   epoch            engine_vg.train_one_epoch against the resident-batch body
#   ab:VAR=a,b       interleaved A/B of one environment knob (two rounds), e.g. ab:REFTR_QFUSE=0,1
TAG=${1:?tag}; shift
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; mkdir -p $O
for step in "$@"; do
  echo "== $step"
  case $step in
    tests)     python -m pytest tests -m gpu -q > $O/${TAG}_gpu_tests.log 2>&1; tail -25 $O/${TAG}_gpu_tests.log ;;
    bench)     python bench.py > $O/${TAG}_bench.log 2>&1; tail -1 $O/${TAG}_bench.log > $O/${TAG}_bench.json; cut -c1-400 $O/${TAG}_bench.json ;;
    artifacts) bash benchmarks/round_artifacts.sh $TAG 2>&1 | cut -c1-300 ;;
    timeline)  timeout 600 python tools/concurrent_timeline.py --out $O/${TAG}_concurrent_timeline.txt > $O/${TAG}_timeline.log 2>&1; head -34 $O/${TAG}_concurrent_timeline.txt ;;
    region)    bash benchmarks/region_trace.sh $TAG 60; python benchmarks/qregion_trace.py 2>&1 | grep "^rt_" | tee $O/${TAG}_qregion_stages.txt ;;
    shapes)    timeout 600 python benchmarks/step_breakdown.py > $O/${TAG}_shape_breakdown.txt 2>&1; head -30 $O/${TAG}_shape_breakdown.txt ;;
    wgrad)     timeout 600 python benchmarks/wgrad_group_bench.py > $O/${TAG}_wgrad_group_bench.txt 2>&1; tail -8 $O/${TAG}_wgrad_group_bench.txt ;;
    tiles)     REFTR_LAB=1 FLUSH=1 ONLY=lin HINTS=0,31,33,51,21 timeout 600 python benchmarks/tile_sweep.py > $O/${TAG}_tile_sweep_cold.txt 2>&1; head -24 $O/${TAG}_tile_sweep_cold.txt ;;
    configs)   ( timeout 300 python benchmarks/cfg4_seg.py 2>&1 | grep hipgraph; timeout 300 python benchmarks/cfg5_stress.py 2>&1 | grep hipgraph; timeout 300 python benchmarks/eval_throughput.py 2>&1 | grep hipgraph ) > $O/${TAG}_other_configs.txt; cat $O/${TAG}_other_configs.txt ;;
    epoch)     timeout 600 python benchmarks/epoch_throughput.py 2>&1 | tail -1 > $O/${TAG}_epoch_throughput.json; cut -c1-400 $O/${TAG}_epoch_throughput.json ;;
    ab:*)      kv=${step#ab:}; VAR=${kv%%=*} VALS="$(echo ${kv#*=} | tr ',' ' ')" bash benchmarks/ab_env.sh | tee $O/${TAG}_ab_${kv%%=*}.txt ;;
    *)         echo "unknown step $step" ;;
  esac
done
