for v in "X=1" "REFTR_ATTN_NW=4" "REFTR_ATTN_NW=16" "REFTR_ATTN_REG=0" "REFTR_W2_TARGET=384" "REFTR_W2_TARGET=512" "REFTR_W2_MINM=256" "REFTR_WG_SIDE=7" "REFTR_WG_SIDE=0" "REFTR_GROUP_CONV=2" "REFTR_ZERO_SIDE=1"; do
  echo "== $v: $(env $v python bench.py --no-cpu-baseline --no-kernel-roofline --steps 40 --warmup 8 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.3f ms  %.1f img/s" % (d["ms_per_step_median"], d["value"]))')"
done
