TAG=${1:-rXX}; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; mkdir -p $O; cd /tmp
timeout 600 rocprofv3 --kernel-trace -d $O/${TAG}_tl -- python $R/bench.py --steps 12 --warmup 6 --no-cpu-baseline --no-kernel-roofline > $O/${TAG}_tl.log 2>&1
cd $R; DB=$(find $O/${TAG}_tl -name "*.db" | head -1)
python tools/step_milestones.py $DB > $O/${TAG}_milestones.txt 2>&1; cat $O/${TAG}_milestones.txt
python tools/step_timeline.py $DB 30 > $O/${TAG}_timeline.txt 2>&1; head -12 $O/${TAG}_timeline.txt
rm -rf $O/${TAG}_tl
