"""Concurrency timeline of ONE replayed step (multi-stream hipGraph) from a rocprofv3 kernel trace:
how long 0 / 1 / 2 / 3+ kernels are resident, and which kernels run ALONE (the serial sections that bound the step).

    python tools/step_timeline.py <trace.db> [top]
"""
import collections, sqlite3, sys

db = sqlite3.connect(sys.argv[1])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
rows = db.execute("select s.display_name, d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s "
                  "on d.kernel_id = s.id order by d.start").fetchall()
ad = [i for i, r in enumerate(rows) if "sqnorm_kernel" in r[0] or "sq_sum_kernel" in r[0]]      # rt_sqnorm / rt_sqnorm_finish (fused norm, round 4)
ad = [i for n, i in enumerate(ad) if n + 1 == len(ad) or any("adamw_" in r[0] for r in rows[i + 1:ad[n + 1]])]   # the step-ending one of a run
cands = [(rows[ad[i + 1]][2] - rows[ad[i]][2], ad[i] + 1, ad[i + 1] + 1) for i in range(len(ad) - 1)
         if not any("spin_kernel" in r[0] for r in rows[ad[i] + 1:ad[i + 1] + 1])]
cands.sort()
wall, a, b = cands[len(cands) // 2]                 # the median step
step = rows[a:b]
t0 = rows[a - 1][2]                                 # end of the previous step's last kernel
ev = []
for n, s, e in step:
    ev.append((s, 1, n)); ev.append((e, -1, n))
ev.sort(key=lambda x: (x[0], x[1]))
live = collections.Counter()
hist = collections.Counter(); alone = collections.Counter(); pair = collections.Counter()
cur, last = 0, t0
for tm, d, n in ev:
    dt = tm - last
    if dt > 0:
        hist[min(cur, 3)] += dt
        if cur == 1:
            alone[next(iter(k for k, v in live.items() if v > 0))] += dt
        elif cur == 2:
            pair[" + ".join(sorted(k[:40] for k, v in live.items() if v > 0))] += dt
    live[n] += d; cur += d; last = tm
short = lambda k: k.replace("(anonymous namespace)::", "").replace("void ", "")[:86]
print("median step: %d kernels, wall %.3f ms, kernel-busy sum %.3f ms" % (len(step), wall / 1e6, sum(e - s for _, s, e in step) / 1e6))
for c in range(4):
    print("  %s kernels resident: %7.3f ms  (%4.1f %%)" % (("3+" if c == 3 else str(c)), hist[c] / 1e6, 100.0 * hist[c] / wall))
print("kernels running ALONE (no other kernel resident), by total time:")
for k, v in alone.most_common(top):
    n = sum(1 for r in step if r[0] == k)
    print("  %8.1f us  %4d launches  %s" % (v / 1e3, n, short(k)))
print("pairs:")
for k, v in pair.most_common(8):
    print("  %8.1f us  %s" % (v / 1e3, k))
