"""Dump one replayed step's kernel sequence (name, grid, duration) between two marker kernels from a rocprofv3 trace."""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1]); a_key, b_key = sys.argv[2], sys.argv[3]
rows = db.execute("select s.display_name, d.start, d.end, d.grid_size_x, d.grid_size_y, d.workgroup_size_x from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start").fetchall()
ad = [i for i, r in enumerate(rows) if "sqnorm_kernel" in r[0] or "sq_sum_kernel" in r[0]]      # rt_sqnorm / rt_sqnorm_finish (fused norm, round 4)
ad = [i for n, i in enumerate(ad) if n + 1 == len(ad) or any("adamw_" in r[0] for r in rows[i + 1:ad[n + 1]])]   # the step-ending one of a run
step = rows[ad[-2] + 1: ad[-1] + 1]
ia = next(i for i, r in enumerate(step) if a_key in r[0])
ib = next(i for i, r in enumerate(step) if b_key in r[0] and i > ia)
tot = 0
for r in step[ia:ib]:
    n = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", r[0]); n = re.sub(r"EvPKDF16bS2_NS_\d+\w+", "", n)[:60]
    d = (r[2] - r[1]) / 1e3; tot += d
    print("%-60s grid %6d x %-3d %7.1f us" % (n, r[3] // max(r[5], 1), r[4], d))
print("total %.1f us over %d kernels" % (tot, ib - ia))
