export TMPDIR=/tmp; R=$PWD; cd /tmp
timeout 600 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum -d $R/gpurun_out/pmc_l2 -- python $R/benchmarks/pmc_kernels.py > $R/gpurun_out/pmc_l2.log 2>&1
cd $R; tail -2 gpurun_out/pmc_l2.log | cut -c1-200
python - <<'PY'
import sqlite3, glob, collections
dbs = glob.glob("gpurun_out/pmc_l2/**/*.db", recursive=True)
db = sqlite3.connect(dbs[0])
rows = db.execute("select kernel_name, grid_size_x, grid_size_y, counter_name, value from counters_collection").fetchall()
agg = collections.OrderedDict()
for k, gx, gy, c, v in rows:
    if "conv_" not in k: continue
    key = (k[22:72], gx // 256, gy)
    a = agg.setdefault(key, collections.defaultdict(lambda: [0, 0.0]))
    a[c][0] += 1; a[c][1] += v
print("%-52s %9s | %10s %10s %8s %12s" % ("kernel", "grid", "L2 req", "L2 miss", "hit %", "EA rd req"))
for (k, gx, gy), a in agg.items():
    g = lambda c: a[c][1] / max(a[c][0], 1)
    print("%-52s %5dx%-3d | %10.0f %10.0f %8.1f %12.0f" % (k, gx, gy, g("TCC_REQ_sum"), g("TCC_MISS_sum"), 100 * g("TCC_HIT_sum") / max(g("TCC_HIT_sum") + g("TCC_MISS_sum"), 1), g("TCC_EA0_RDREQ_sum")))
PY
