# Fabric fetch bytes of the v2 weight-gradient groups (layer3, layer4, layer2 of the bench workload) per placement mode
export TMPDIR=/tmp; R=$PWD; cd /tmp
for x in ${MODES:-0 1}; do
rm -rf $R/gpurun_out/pmc_w2f_$x
REFTR_W2_XCD=$x timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_w2f_$x -- python $R/benchmarks/pmc_w2.py > $R/gpurun_out/pmc_w2f_$x.log 2>&1
done
cd $R
python - <<'PY'
import sqlite3, glob, collections, os
for tag in os.environ.get("MODES", "0 1").split():
    dbs = glob.glob(f"gpurun_out/pmc_w2f_{tag}/**/*.db", recursive=True)
    if not dbs: print("no db", tag); continue
    db = sqlite3.connect(dbs[0])
    rows = db.execute("select kernel_name, grid_size_x, counter_name, value, duration from counters_collection").fetchall()
    agg = collections.OrderedDict()
    for k, gx, c, v, d in rows:
        if "w2_" not in k: continue
        a = agg.setdefault((k[:50], gx), [0, 0.0, 0.0]); a[0] += 1; a[1] += v; a[2] += d
    print("== REFTR_W2_XCD =", tag)
    for (k, gx), (n, v, d) in agg.items():
        print("%-52s blocks %6d  %7.1f us  fetch %8.1f MB" % (k, gx // (512 if 'grouped' in k else 256), d / n / 1e3, 2 * v / n * 1024 / 1e6))
PY
rm -rf gpurun_out/pmc_w2f_*
