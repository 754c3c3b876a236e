export REFTR_LAB=1   # kernel tuning switches live in the lab library only (benchmarks/README.md): build it with REFTR_LAB=1 first
export TMPDIR=/tmp; R=$PWD; cd /tmp
for abl in 0 1; do
rm -rf $R/gpurun_out/pmc_w2_$abl
REFTR_W2_ABL=$abl ONLY=none timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES -d $R/gpurun_out/pmc_w2_$abl -- python $R/benchmarks/pmc_w2.py > $R/gpurun_out/pmc_w2_$abl.log 2>&1
done
rm -rf $R/gpurun_out/pmc_w2_l; timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAVE_CYCLES TCC_HIT_sum TCC_MISS_sum -d $R/gpurun_out/pmc_w2_l -- python $R/benchmarks/pmc_w2.py > $R/gpurun_out/pmc_w2_l.log 2>&1
cd $R
python - <<'PY'
import sqlite3, glob, collections
for tag in ("0", "1", "l"):
    dbs = glob.glob(f"gpurun_out/pmc_w2_{tag}/**/*.db", recursive=True)
    if not dbs: print("no db", tag); continue
    db = sqlite3.connect(dbs[0])
    rows = db.execute("select kernel_name, grid_size_x, counter_name, value, duration from counters_collection").fetchall()
    agg = collections.OrderedDict()
    for k, gx, c, v, d in rows:
        if "w2_" not in k: continue
        key = (k[:60], gx)
        a = agg.setdefault(key, {"n": collections.Counter(), "v": collections.Counter(), "d": collections.Counter()})
        a["n"][c] += 1; a["v"][c] += v; a["d"][c] += d
    print("== ablation / set", tag)
    for (k, gx), a in agg.items():
        g = lambda c: a["v"][c] / max(a["n"][c], 1)
        names = sorted(a["n"])
        dur = a["d"][names[0]] / max(a["n"][names[0]], 1) / 1e3
        print("%-62s grid %6d  %7.1f us  " % (k, gx // 512, dur) + "  ".join("%s=%.4g" % (c.replace("SQ_", ""), g(c)) for c in names))
PY
