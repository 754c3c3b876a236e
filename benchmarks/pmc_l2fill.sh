# L2 -> compute-unit read traffic of every kernel of the step (TCP_TCC_READ_REQ_sum: read requests the vector L1s send to the L2s;
# 64-B requests, two per 128-B line) next to the kernel's duration: the achieved L2 -> CU fill rate.  Eager launches (--no-graph),
# one stream, 3 steps.  -> gpurun_out/$TAG_l2fill.txt
TAG=${1:-rXX}; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; mkdir -p $O; cd /tmp
REFTR_STREAMS=0 timeout 800 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum -d $O/${TAG}_l2 -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-roofline --no-graph > $O/${TAG}_l2.log 2>&1
tail -2 $O/${TAG}_l2.log | cut -c1-200
cd $R; DB=$(find $O/${TAG}_l2 -name "*.db" | head -1)
python - "$DB" > $O/${TAG}_l2fill.txt <<'PY'
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
agg = {}
q = "select kernel_name, value, start, end from counters_collection" if "start" in cols else None
if q is None:
    print("columns:", cols); sys.exit(0)
for name, val, st, en in db.execute(q):
    a = agg.setdefault(name, [0, 0.0, 0.0]); a[0] += 1; a[1] += float(val); a[2] += (en - st)
print("kernel                                                                                     calls  L2->CU MB/call  us/call  TB/s  (TCP_TCC_READ_REQ_sum x 64 B; profiled clocks)")
for k, (n, v, t) in sorted(agg.items(), key=lambda kv: -kv[1][2])[:40]:
    mb = v * 64 / n / 1e6; us = t / n / 1e3
    print("%-90s %5d  %10.2f  %8.1f  %5.2f" % (re.sub(r"\(anonymous namespace\)::|void ", "", k)[:90], n, mb, us, mb / max(us, 1e-9)))
PY
head -45 $O/${TAG}_l2fill.txt; rm -rf $O/${TAG}_l2
