# kernel + copy trace of the epoch loop, analysed by tools/epoch_gaps.py -> gpurun_out/${1:-rXX}_epoch_gaps.txt
TAG=${1:-rXX}; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; mkdir -p $O; cd /tmp
timeout 900 rocprofv3 --kernel-trace --memory-copy-trace -d $O/${TAG}_ep -- python $R/benchmarks/epoch_throughput.py --batches 60 --warm 12 > $O/${TAG}_ep.log 2>&1
cd $R; DB=$(find $O/${TAG}_ep -name "*.db" | head -1)
python tools/epoch_gaps.py $DB > $O/${TAG}_epoch_gaps.txt 2>&1; cat $O/${TAG}_epoch_gaps.txt; tail -1 $O/${TAG}_ep.log | cut -c1-300
rm -rf $O/${TAG}_ep
