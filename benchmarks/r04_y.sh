export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; mkdir -p $O
cd /tmp
REFTR_STREAMS=0 timeout 600 rocprofv3 --kernel-trace -d $O/r04y_trace1 -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-kernel-roofline > $O/r04y_trace1.log 2>&1
cd $R
DB1=$(find $O/r04y_trace1 -name "*.db" | head -1)
for ph in "encoder fwd" "query encoder + decoder fwd" "loss" "head + decoder bwd" "qenc bwd"; do python tools/step_phases.py $DB1 "$ph" 2>&1 | sed -n '/^  *[0-9.]* us/p' > "$O/r04y_phase_$(echo $ph | tr ' +' '__').txt"; done
python tools/step_phases.py $DB1 "encoder fwd" > $O/r04y_enc_fwd_dump.txt 2>&1
python tools/step_phases.py $DB1 "qenc bwd" > $O/r04y_enc_bwd_dump.txt 2>&1
python tools/step_phases.py $DB1 "decoder" > $O/r04y_dec_dump.txt 2>&1
python tools/step_phases.py $DB1 "loss" > $O/r04y_loss_dump.txt 2>&1
rm -rf $O/r04y_trace1
wc -l $O/r04y_*dump.txt
