"""Phase timeline of ONE replayed step from a rocprofv3 kernel trace (single-stream run): time between marker kernels."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select s.display_name, d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start").fetchall()
# the gradient-norm kernel ends an iteration in both schedules (deferred: [AdamW of i-1 | fwd | bwd | norm]; eager: [... norm | AdamW])
def is_norm(n):      # the launch that ends an iteration: rt_sqnorm (rounds 1-3) or rt_sqnorm_finish's slot sum (round 4: fused norm)
    return "sqnorm_kernel" in n or "sq_sum_kernel" in n
ad = [i for i, r in enumerate(rows) if is_norm(r[0])]
# (round 3: the BERT slice's share of the norm is a second, earlier sqnorm launch -- the iteration ends at the one that is followed by
# the next iteration's AdamW, i.e. the last of each run of sqnorm launches without an adamw_kernel in between)
ad = [i for n, i in enumerate(ad) if n + 1 == len(ad) or any("adamw_" in r[0] for r in rows[i + 1:ad[n + 1]])]
# the shortest step of the run (skips warm-up and bench.py's backlogged roofline pass behind a spin kernel)
cands = [(rows[ad[i + 1]][2] - rows[ad[i] + 1][1], ad[i] + 1, ad[i + 1] + 1) for i in range(len(ad) - 1)
         if not any("spin_kernel" in r[0] for r in rows[ad[i] + 1:ad[i + 1] + 1])]
_, a, b = min(cands)
step = rows[a:b]
t0 = step[0][1]
marks = [("AdamW of the previous step (deferred) + weight prep + pack", "adamw_"), ("pack", "img_pack"), ("stem", "stem_"), ("language branch inline (REFTR_STEM_FIRST: forked behind the stem): AdamW (BERT slice) + BERT fwd", "adamw_"), ("ResNet fwd (+BERT if 1 stream)", ("maxpool", "bottleneck_fwd")), ("input_proj+GN", "gn_stats_kernel"),
         ("encoder fwd", "gn_apply"), ("query encoder + decoder fwd + head", ("qenc_attn_fwd", "qenc_fwd_kernel")), ("loss", ("box_loss", "head_loss_kernel")),
         ("head + decoder bwd", ("box_loss", "decoder_bwd_kernel", "layernorm_bwd")), ("qenc bwd + encoder bwd", ("qenc_attn_bwd", "qenc_bwd_kernel")), ("GN/input_proj bwd", "gn_bwd_stats"),
         ("ResNet bwd (+BERT bwd)", "gn_bwd_apply"), ("gradient norm", "sqnorm")]
idx, pos = [], 0
for label, key in marks:
    rng = range(pos, len(step)) if key != "sqnorm" else range(len(step) - 1, pos - 1, -1)      # the step-ending norm launch
    for i in rng:
        if (is_norm(step[i][0]) if key == "sqnorm" else any(k in step[i][0] for k in ((key,) if isinstance(key, str) else key))):
            idx.append((label, i)); pos = i + 1; break
print("step: %d kernels, %.2f ms (kernel-busy %.2f ms)" % (len(step), (step[-1][2] - t0) / 1e6, sum(r[2] - r[1] for r in step) / 1e6))
for n, (label, i) in enumerate(idx):
    j = idx[n + 1][1] if n + 1 < len(idx) else len(step)
    seg = step[i:j]
    print("  %-40s %4d kernels  %7.3f ms  (busy %7.3f)" % (label, len(seg), (seg[-1][2] - seg[0][1]) / 1e6, sum(r[2] - r[1] for r in seg) / 1e6))
print("  %-40s %4d kernels  %7.3f ms" % ("(before first marker: zero-fill, prep)", idx[0][1], (step[idx[0][1]][1] - t0) / 1e6))
if len(sys.argv) > 2 and sys.argv[2] == "head":      # the kernels in front of the first marker (zero-fill, staging, counters)
    for k in range(idx[0][1]):
        nm, st, en = step[k]
        print("  %9.1f us  %6.1f us  %s" % ((st - t0) / 1e3, (en - st) / 1e3, nm.replace("(anonymous namespace)::", "").replace("void ", "")[:110]))
if len(sys.argv) > 2:               # dump the kernel sequence of the phases whose label contains argv[2]: start offset, duration, gap, name
    for n, (label, i) in enumerate(idx):
        if sys.argv[2] not in label:
            continue
        j = idx[n + 1][1] if n + 1 < len(idx) else len(step)
        print("--", label)
        for k in range(i, j):
            nm, st, en = step[k]
            gap = (st - step[k - 1][2]) / 1e3 if k > 0 else 0.0
            print("  %9.1f us  %6.1f us  gap %5.1f  %s" % ((st - t0) / 1e3, (en - st) / 1e3, gap, nm.replace("(anonymous namespace)::", "").replace("void ", "")[:100]))
if len(sys.argv) > 2 and sys.argv[2] == "hist":     # per-kernel-name totals of the one step
    agg = {}
    for nm, st, en in step:
        k = nm.replace("(anonymous namespace)::", "").replace("void ", "")[:120]
        c, t = agg.get(k, (0, 0.0)); agg[k] = (c + 1, t + (en - st) / 1e3)
    print("-- kernels of the step by total time: launches, total us, name")
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("  %4d  %8.1f us  %s" % (c, t, k))
