# usage: bash benchmarks/ab_trace.sh tagA "ENV=.. ENV=.." tagB "ENV=.."   -- one-stream kernel traces of 5 replayed steps per arm:
# phase timeline (tools/step_phases.py) and per-kernel totals (tools/rocpd_stats.py) under gpurun_out/abt_<tag>_{phases,stats}.txt
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; mkdir -p $O
while [ $# -ge 2 ]; do
  TAG=$1; ARM=$2; shift 2
  cd /tmp
  env $ARM REFTR_STREAMS=${STREAMS:-0} timeout 600 rocprofv3 --kernel-trace -d $O/abt_$TAG -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-kernel-roofline > $O/abt_$TAG.log 2>&1
  cd $R
  DB=$(find $O/abt_$TAG -name "*.db" | head -1)
  python tools/step_phases.py $DB > $O/abt_${TAG}_phases.txt 2>&1
  python tools/rocpd_stats.py $DB --top 30 > $O/abt_${TAG}_stats.txt 2>&1
  rm -rf $O/abt_$TAG
  echo "== $TAG ($ARM)"; tail -16 $O/abt_${TAG}_phases.txt
done
