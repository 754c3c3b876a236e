# what the data-parallel schedule (single-rank RCCL, REFTR_DDP_FORCE=1) launches per step that the N = 1 step does not: per-kernel-name
# totals of two rocprofv3 kernel traces, per step -> gpurun_out/${1:-rXX}_ddp_kernel_diff.txt
TAG=${1:-rXX}; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; mkdir -p $O; cd /tmp
timeout 600 rocprofv3 --kernel-trace -d $O/${TAG}_n1 -- python $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-kernel-roofline > $O/${TAG}_n1.log 2>&1
REFTR_DDP_FORCE=1 timeout 600 rocprofv3 --kernel-trace -d $O/${TAG}_dp -- python $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-kernel-roofline > $O/${TAG}_dp.log 2>&1
cd $R
python - $O/${TAG}_n1 $O/${TAG}_dp > $O/${TAG}_ddp_kernel_diff.txt <<'PY'
import sys, glob, sqlite3, collections
def load(d):
    rows = []
    for f in glob.glob(d + "/**/*.db", recursive=True):
        db = sqlite3.connect(f)
        r = db.execute("select s.display_name, d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start").fetchall()
        if len(r) > len(rows):
            rows = r
    # the timed region: the last 10 steps = the last 10 step-ending kernels back to the 11th-last
    ends = [i for i, r in enumerate(rows) if "finish_stats_kernel" in r[0] or "finish_step_kernel" in r[0]]
    print(d, len(rows), "dispatches,", len(ends), "step ends", file=sys.stderr)
    a, b = ends[-11] + 1, ends[-1] + 1
    agg = collections.defaultdict(lambda: [0, 0.0])
    for n, s, e in rows[a:b]:
        k = n.replace("(anonymous namespace)::", "").replace("void ", "")[:78]
        agg[k][0] += 1; agg[k][1] += (e - s) / 1e3
    return {k: (c / 10.0, t / 10.0) for k, (c, t) in agg.items()}, (rows[b - 1][2] - rows[a - 1][2]) / 1e4
n1, w1 = load(sys.argv[1]); dp, w2 = load(sys.argv[2])
print("per step (traced, streams serialised by the profiler): N=1 %.0f kernels %.0f us busy, wall %.0f us | data-parallel %.0f kernels %.0f us busy, wall %.0f us" % (
    sum(c for c, _ in n1.values()), sum(t for _, t in n1.values()), w1, sum(c for c, _ in dp.values()), sum(t for _, t in dp.values()), w2))
diff = sorted(((dp.get(k, (0, 0))[1] - n1.get(k, (0, 0))[1], k) for k in set(n1) | set(dp)), reverse=True)
print("%-80s %8s %8s %9s %9s" % ("kernel", "n N=1", "n DP", "us N=1", "us DP"))
for d, k in diff[:28] + diff[-8:]:
    print("%-80s %8.1f %8.1f %9.1f %9.1f" % (k, n1.get(k, (0, 0))[0], dp.get(k, (0, 0))[0], n1.get(k, (0, 0))[1], dp.get(k, (0, 0))[1]))
PY
cat $O/${TAG}_ddp_kernel_diff.txt | cut -c1-130; rm -rf $O/${TAG}_n1 $O/${TAG}_dp
