export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; mkdir -p $O
cd /tmp
REFTR_STREAMS=0 timeout 600 rocprofv3 --kernel-trace -d $O/r04ah_trace1 -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-kernel-roofline > $O/r04ah_trace1.log 2>&1
cd $R
DB1=$(find $O/r04ah_trace1 -name "*.db" | head -1)
python tools/step_phases.py $DB1 head | tail -8
python tools/step_phases.py $DB1 "loss" | grep -A12 "^-- loss"
python tools/step_phases.py $DB1 "gradient norm" | grep -A4 "^-- gradient"
rm -rf $O/r04ah_trace1
