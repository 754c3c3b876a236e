REFTR_STREAMS=3 timeout 900 python -m pytest tests/test_model_gpu.py -x -q 2>&1 | tail -2
for s in 1 3 1 3; do
  echo "STREAMS=$s"; REFTR_STREAMS=$s timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-roofline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
done
