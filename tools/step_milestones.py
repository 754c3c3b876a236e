"""Milestones of ONE replayed two-stream step from a rocprofv3 kernel trace: when (ms after the step's first kernel) the phases of
the main chain begin and end, and when the language branch (BERT: the dh = 64 attention kernels) is done in forward and backward.
    python tools/step_milestones.py <trace.db>
"""
import sqlite3, sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select s.display_name, d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s "
                  "on d.kernel_id = s.id order by d.start").fetchall()
ad = [i for i, r in enumerate(rows) if "sqnorm_kernel" in r[0] or "sq_sum_kernel" in r[0]]      # rt_sqnorm / rt_sqnorm_finish (fused norm, round 4)
ad = [i for n, i in enumerate(ad) if n + 1 == len(ad) or any("adamw_" in r[0] for r in rows[i + 1:ad[n + 1]])]   # the step-ending one of a run
cands = [(rows[ad[i + 1]][2] - rows[ad[i]][2], ad[i] + 1, ad[i + 1] + 1) for i in range(len(ad) - 1)
         if not any("spin_kernel" in r[0] for r in rows[ad[i] + 1:ad[i + 1] + 1])]
cands.sort()
wall, a, b = cands[len(cands) // 2]
step = rows[a:b]
t0 = rows[a - 1][2]
ms = lambda t: (t - t0) / 1e6


def first(pat, key=1):
    for r in step:
        if pat in r[0]:
            return ms(r[key])
    return float("nan")


def last(pat, key=2):
    v = float("nan")
    for r in step:
        if pat in r[0]:
            v = ms(r[key])
    return v


def first_of_last(pat):
    """start of the last launch whose name contains `pat`"""
    v = float("nan")
    for r in step:
        if pat in r[0]:
            v = ms(r[1])
    return v


print("median step: %d kernels, wall %.3f ms" % (len(step), wall / 1e6))
for label, v in (
        ("first AdamW launch starts", first("adamw_")), ("last AdamW launch ends", last("adamw_")),
        ("weight prep (last) ends", last("weight_prep")), ("stem conv starts (ResNet forward)", first("stem_conv")),
        ("BERT forward: last dh=64 attention ends", last("attn_fwd_reg_kernel<64")),
        ("input_proj GroupNorm stats (ResNet forward done)", first("gn_stats")),
        ("encoder forward: first dh=32 attention starts", first("attn_fwd_reg_kernel<32")),
        ("encoder forward: last dh=32 attention ends", last("attn_fwd_reg_kernel<32")),
        ("decoder forward: cooperative launch starts", first("decoder_fwd_kernel")), ("decoder forward: cooperative launch ends", last("decoder_fwd_kernel")),
        ("box loss", first("box_loss")),
        ("decoder backward: cooperative launch starts", first("decoder_bwd_kernel")), ("decoder backward: cooperative launch ends", last("decoder_bwd_kernel")),
        ("encoder backward: first attention bwd (dh=32)", first("attn_bwd_fused_kernel<32")), ("encoder backward: last", last("attn_bwd_fused_kernel<32")),
        ("GroupNorm backward (encoder chain done)", first("gn_bwd")),
        ("BERT backward: first dh=64 attention bwd", first("attn_bwd_fused_kernel<64")), ("BERT backward: last", last("attn_bwd_fused_kernel<64")),
        ("last backward-data / forward product (conv_gemm_dma) ends", last("conv_gemm_dma_kernel")),
        ("last weight-gradient group starts", first_of_last("w2_grouped")), ("last weight-gradient group ends", last("w2_grouped")), ("gradient norm ends (step end)", (last("sq_sum_kernel") if last("sq_sum_kernel") == last("sq_sum_kernel") else last("sqnorm_kernel")))):
    print("  %7.3f ms  %s" % (v, label))
