# one-stream kernel trace of the replayed step; dumps the ordered kernel list (name, grid, duration, gap to the previous kernel) between
# marker kernels:  bash benchmarks/dump_steps.sh <out tag> <from-kernel substring> <to-kernel substring>
TAG=${1:-dump}; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; cd /tmp
REFTR_STREAMS=0 timeout 600 rocprofv3 --kernel-trace -d $O/${TAG}_tr -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-kernel-roofline > $O/${TAG}_tr.log 2>&1
cd $R; DB=$(find $O/${TAG}_tr -name "*.db" | head -1)
python - "$DB" "$2" "$3" > $O/${TAG}_kernels.txt <<'PY'
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1]); a_key, b_key = sys.argv[2], sys.argv[3]
rows = db.execute("select s.display_name, d.start, d.end, d.grid_size_x, d.grid_size_y, d.workgroup_size_x from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start").fetchall()
ad = [i for i, r in enumerate(rows) if "sqnorm_kernel" in r[0]]
step = rows[ad[-2] + 1: ad[-1] + 1]
ia = next(i for i, r in enumerate(step) if a_key in r[0])
ib = max(i for i, r in enumerate(step) if b_key in r[0])
tot = gaps = 0
for j in range(ia, ib + 1):
    r = step[j]
    n = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", r[0]); n = re.sub(r"EvPKDF16bS2_8GemmArgs", "", n)[:58]
    d = (r[2] - r[1]) / 1e3; g = (r[1] - step[j - 1][2]) / 1e3 if j > 0 else 0.0
    tot += d; gaps += max(g, 0)
    print("%-58s grid %6d x %-3d %7.1f us  gap %5.1f" % (n, r[3] // max(r[5], 1), r[4], d, g))
print("total %.1f us busy + %.1f us gaps over %d kernels" % (tot, gaps, ib - ia + 1))
PY
rm -rf $O/${TAG}_tr; tail -1 $O/${TAG}_kernels.txt
