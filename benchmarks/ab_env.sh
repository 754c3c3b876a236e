# usage: ab_env.sh "ENV=a" "ENV=b" ...   (alternates, 2 rounds)
for r in 1 2; do for e in "$@"; do
  echo -n "$e: "; env $e timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-roofline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), round(d['value'],1))"
done; done
