mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_encoder_fused_gpu.py -x -q > gpurun_out/r04f_enc.log 2>&1; echo "enc rc $?"
grep -n "assert\|Error\|passed\|failed" gpurun_out/r04f_enc.log | head -20
VAR=REFTR_ENC_FUSE VALS="0 1" timeout 600 bash benchmarks/ab_env.sh > gpurun_out/r04f_ab_enc.txt 2>&1; cat gpurun_out/r04f_ab_enc.txt
cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_r04f -o r04f -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-kernel-roofline --steps 10 > $GRAFT_REPO_ROOT/gpurun_out/r04f_rocprof.log 2>&1; cd $GRAFT_REPO_ROOT
DB=$(find /tmp/prof_r04f -name "*.db" | head -1)
python tools/rocpd_stats.py $DB --top 60 > gpurun_out/r04f_kernel_stats.md 2>&1; grep -n "enc_tail" gpurun_out/r04f_kernel_stats.md
