export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; mkdir -p $O
cd /tmp
REFTR_STREAMS=0 timeout 600 rocprofv3 --kernel-trace -d $O/r04ab_trace1 -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-kernel-roofline > $O/r04ab_trace1.log 2>&1
cd $R
DB1=$(find $O/r04ab_trace1 -name "*.db" | head -1)
python tools/step_phases.py $DB1 "ResNet bwd" > $O/r04ab_bwd_dump.txt 2>&1
python tools/step_phases.py $DB1 "AdamW" > $O/r04ab_adamw_dump.txt 2>&1
rm -rf $O/r04ab_trace1
wc -l $O/r04ab_*dump.txt
