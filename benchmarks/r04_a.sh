# round 4, first GPU session: fail-safe + matrix-AdamW tests, A/B of the operand-emitting optimizer, the concurrent timeline, the
# per-shape breakdown.  Everything lands under gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_failsafe_gpu.py -x -q > gpurun_out/r04a_failsafe.log 2>&1; echo "failsafe rc $?" 
tail -15 gpurun_out/r04a_failsafe.log
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_decoder_coop_gpu.py -x -q > gpurun_out/r04a_model.log 2>&1; echo "model rc $?"
tail -5 gpurun_out/r04a_model.log
VAR=REFTR_OPT_EMIT VALS="0 1" timeout 600 bash benchmarks/ab_env.sh > gpurun_out/r04a_ab_emit.txt 2>&1; cat gpurun_out/r04a_ab_emit.txt
timeout 600 python tools/concurrent_timeline.py --out gpurun_out/r04a_concurrent_timeline.txt > gpurun_out/r04a_timeline.log 2>&1; echo "timeline rc $?"; tail -70 gpurun_out/r04a_timeline.log
timeout 600 python benchmarks/step_breakdown.py > gpurun_out/r04a_shape_breakdown.txt 2>&1; echo "breakdown rc $?"; head -5 gpurun_out/r04a_shape_breakdown.txt
