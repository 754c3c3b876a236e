# one-stream kernel trace of the replayed step -> phase table + the kernel sequence of the few-row region (query encoder ... qenc backward)
TAG=${1:-rXX}; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; mkdir -p $O; cd /tmp
REFTR_STREAMS=0 timeout 600 rocprofv3 --kernel-trace -d $O/${TAG}_tr -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-kernel-roofline > $O/${TAG}_tr.log 2>&1
cd $R; DB=$(find $O/${TAG}_tr -name "*.db" | head -1)
python tools/step_phases.py $DB "" > $O/${TAG}_sequence.txt 2>&1
head -16 $O/${TAG}_sequence.txt
awk "/^-- query encoder/,/^-- GN/" $O/${TAG}_sequence.txt | grep -v "conv_gemm_dma\|layernorm_bwd_vec_kernel<1>\|attn_bwd_fused" | cut -c1-150 | head -${2:-60}
rm -rf $O/${TAG}_tr
