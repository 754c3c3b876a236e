mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python tools/concurrent_timeline.py --reps 50 --out gpurun_out/r04r_concurrent_timeline.txt > gpurun_out/r04r_tl.log 2>&1; tail -3 gpurun_out/r04r_tl.log; cat gpurun_out/r04r_concurrent_timeline.txt | head -40
timeout 900 python bench.py --steps 50 --warmup 10 > gpurun_out/r04r_bench.json 2> gpurun_out/r04r_bench.err; tail -1 gpurun_out/r04r_bench.json
