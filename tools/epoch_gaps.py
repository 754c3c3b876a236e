"""Where the loop loses time against its body: from a rocprofv3 --kernel-trace --memory-copy-trace of benchmarks/epoch_throughput.py,
per iteration of the replayed loop: the step itself (first kernel of the replay -> its sq_sum kernel), the device-idle gap to the next
replay's first kernel, and what ran in between (kernels and copies).

    python tools/epoch_gaps.py <trace.db>
"""
import sqlite3, statistics, sys

db = sqlite3.connect(sys.argv[1])
k = db.execute("select s.display_name, d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start").fetchall()
try:
    c = db.execute("select name, start, end, size from rocpd_memory_copy order by start").fetchall()
except Exception as e:                                   # schema differs: list the tables
    print("no rocpd_memory_copy:", e, [r[0] for r in db.execute("select name from sqlite_master where type='table'")][:40])
    c = []
ends = [i for i, r in enumerate(k) if "sq_sum_kernel" in r[0]]
# a replayed step starts with the AdamW pass (adamw_mat_kernel) that follows the previous sq_sum by a few launches
steps = []
for a, b in zip(ends[:-1], ends[1:]):
    first = next((i for i in range(a + 1, b) if "adamw_mat_kernel" in k[i][0]), None)
    if first is None:
        continue
    steps.append((k[a][2], k[first][1], k[b][2], a, first, b))
steps = steps[len(steps) // 3:]                           # the steady state (the warm-up epoch and the capture come first)
gap = [s[1] - s[0] for s in steps]
body = [s[2] - s[1] for s in steps]
print("iterations analysed: %d" % len(steps))
print("replayed step, first AdamW kernel -> gradient norm: median %.1f us" % (statistics.median(body) / 1e3))
print("gap, gradient norm -> next step's first kernel:      median %.1f us  (p10 %.1f, p90 %.1f)" % (
    statistics.median(gap) / 1e3, sorted(gap)[len(gap) // 10] / 1e3, sorted(gap)[9 * len(gap) // 10] / 1e3))
mid = steps[len(steps) // 2]
print("\none gap in detail (t relative to the gradient-norm kernel's end, us):")
for n, s, e in k[mid[3] + 1:mid[4] + 1]:
    print("  %8.1f .. %8.1f  kernel  %s" % ((s - mid[0]) / 1e3, (e - mid[0]) / 1e3, n[:90]))
for n, s, e, sz in c:
    if mid[0] - 2e6 <= s <= mid[1] + 7e6:
        print("  %8.1f .. %8.1f  copy    %s  %.2f MB" % ((s - mid[0]) / 1e3, (e - mid[0]) / 1e3, n, (sz or 0) / 1e6))
