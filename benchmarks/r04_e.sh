mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_r04e -o r04e -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-kernel-roofline --steps 20 > $GRAFT_REPO_ROOT/gpurun_out/r04e_rocprof.log 2>&1; cd $GRAFT_REPO_ROOT
DB=$(find /tmp/prof_r04e -name "*.db" | head -1)
python tools/rocpd_stats.py $DB --top 60 > gpurun_out/r04e_kernel_stats.md 2>&1; grep -n "enc_tail\|attn_fwd_reg_kernel<32\|attn_bwd_fused_kernel<32\|adamw\|sq_" gpurun_out/r04e_kernel_stats.md
python tools/step_phases.py $DB > gpurun_out/r04e_phases.txt 2>&1; cat gpurun_out/r04e_phases.txt
python tools/step_phases.py $DB "encoder" > gpurun_out/r04e_phases_enc.txt 2>&1; grep -n "us " gpurun_out/r04e_phases_enc.txt | head -80
