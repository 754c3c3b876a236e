PH=${1:-decoder}; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; cd /tmp
REFTR_STREAMS=0 timeout 600 rocprofv3 --kernel-trace -d $O/dump_tr -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-kernel-roofline > $O/dump_tr.log 2>&1
cd $R; python tools/step_phases.py $(find $O/dump_tr -name "*.db" | head -1) "$PH" > $O/dump_phase.txt 2>&1; rm -rf $O/dump_tr
grep -A200 "^-- " $O/dump_phase.txt | head -${2:-140}
