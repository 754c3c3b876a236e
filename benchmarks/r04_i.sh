mkdir -p gpurun_out
export TMPDIR=/tmp
VAR=REFTR_OPT_SERIAL VALS="0 1" timeout 600 bash benchmarks/ab_env.sh > gpurun_out/r04i_ab_serial.txt 2>&1; cat gpurun_out/r04i_ab_serial.txt
REFTR_OPT_SERIAL=1 timeout 600 python tools/concurrent_timeline.py --out gpurun_out/r04i_timeline_serial.txt > gpurun_out/r04i_timeline.log 2>&1; head -36 gpurun_out/r04i_timeline_serial.txt
