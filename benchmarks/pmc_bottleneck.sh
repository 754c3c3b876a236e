export TMPDIR=/tmp; R=$PWD; mkdir -p $R/gpurun_out; cd /tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES -d $R/gpurun_out/pmc_b -- python $R/benchmarks/pmc_bottleneck.py > $R/gpurun_out/pmc_b.log 2>&1
cd $R; tail -2 gpurun_out/pmc_b.log | cut -c1-200
python - <<'PY' > gpurun_out/r04ad_pmc_bottleneck.txt
import sqlite3, glob, collections
db = sqlite3.connect(glob.glob("gpurun_out/pmc_b/**/*.db", recursive=True)[0])
rows = db.execute("select kernel_name, grid_size_x, grid_size_y, counter_name, value, duration from counters_collection").fetchall()
agg = collections.OrderedDict()
for k, gx, gy, c, v, d in rows:
    if "bottleneck" not in k: continue
    key = (k.replace("(anonymous namespace)::", "").replace("void ", "")[:60], gx)
    a = agg.setdefault(key, {"n": collections.Counter(), "v": collections.Counter(), "d": 0.0})
    a["n"][c] += 1; a["v"][c] += v; a["d"] += d / 8.0
print("%-62s %7s %6s | %6s %6s %6s %6s %6s %7s" % ("kernel", "threads", "us", "wait", "iwait", "ilds", "active", "mfma", "ldsconf"))
for (k, gx), a in agg.items():
    g = lambda c: a["v"][c] / max(a["n"][c], 1)
    wc = g("SQ_WAVE_CYCLES") or 1
    print("%-62s %7d %6.1f | %6.2f %6.2f %6.2f %6.2f %6.2f %7.3f" % (k, gx, a["d"] / max(a["n"]["SQ_WAVE_CYCLES"], 1) / 1e3 * 8,
          g("SQ_WAIT_ANY") / wc, g("SQ_WAIT_INST_ANY") / wc, g("SQ_WAIT_INST_LDS") / wc, g("SQ_ACTIVE_INST_ANY") / wc,
          g("SQ_VALU_MFMA_BUSY_CYCLES") / (g("SQ_BUSY_CYCLES") or 1), g("SQ_LDS_BANK_CONFLICT") / wc))
PY
cat gpurun_out/r04ad_pmc_bottleneck.txt; rm -rf gpurun_out/pmc_b
