export TMPDIR=/tmp; R=$PWD; cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  REFTR_STREAMS=0 timeout 800 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/pmc_$c -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-roofline --no-graph > $R/gpurun_out/pmc_$c.log 2>&1
  tail -2 $R/gpurun_out/pmc_$c.log | cut -c1-200
done
cd $R; find gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE -name "*.db" | head; 
python - <<'PY'
import sqlite3, glob
for c in ("FETCH_SIZE","WRITE_SIZE"):
    dbs = glob.glob(f"gpurun_out/pmc_{c}/**/*.db", recursive=True)
    if not dbs: print("no db", c); continue
    db = sqlite3.connect(dbs[0])
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table' or type='view'")]
    print(c, [t for t in tabs if 'pmc' in t.lower() or 'counter' in t.lower()])
PY
