#!/bin/bash
# kernel sequence of the single-stream step (every phase) + per-kernel totals
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; mkdir -p $O
cd /tmp
REFTR_STREAMS=0 timeout 600 rocprofv3 --kernel-trace -d $O/aq_trace1 -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-kernel-roofline > $O/aq_trace1.log 2>&1
cd $R
DB1=$(find $O/aq_trace1 -name "*.db" | head -1)
python tools/step_phases.py $DB1 " " > $O/r04aq_step_sequence.txt 2>&1
python tools/step_phases.py $DB1 hist > $O/r04aq_step_hist.txt 2>&1; python tools/step_phases.py $DB1 head > $O/r04aq_step_head.txt 2>&1
rm -rf $O/aq_trace1
sed -n 16,30p $O/r04aq_step_head.txt | cut -c1-200
