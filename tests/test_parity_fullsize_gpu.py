"""GPU parity at FULL depth (BERT 12 + encoder 6 + decoder 6 layers) against the CPU oracle, element-wise:

  * configs[0] of BASELINE.json -- RefCOCO-shaped, ResNet-50, 320x320, batch 2, L = 40 -- single- and multi-phrase;
  * configs[3] -- the REC+RES multitask model (RefTRSeg) -- at 320x320 / batch 2 against the oracle, and at its full size
    (640x640, batch 8) through size-independent properties (the oracle needs minutes per step there).

Two oracle modes (oracle/reftr_oracle.py): q=False is the reference's fp32 arithmetic, q=True applies the HIP path's bf16
rounding points (GEMM operands, stored backbone activations).  What is gated and why (measured values are printed by every
test; `MEASURED` holds the values of the build these thresholds were set on, the asserts sit at <= 1.5 x them):

  boxes / losses vs q=False   the north star's quantities: 1.2-1.3e-3 rel-L2 on the boxes, <= 2.4e-3 on single loss terms.
  logits vs q=True            5.9-6.9e-3.  This is NOT 1e-3 and cannot be for ANY implementation that stores bf16
                              activations: two valid fp32 summation orders of the same dot product differ in the last bit,
                              which flips the bf16 rounding (2^-9 relative) of a few per cent of the stored activations of
                              every one of the ~70 GEMM layers; the flips accumulate to 8e-3 on c5 and 6e-3 on the logits
                              between the oracle's own torch-CPU order and the MFMA order (the oracle's q=True and q=False
                              outputs differ from each other by the same amount).  The per-kernel tests (test_gemm_gpu.py,
                              test_ops_gpu.py) feed IDENTICAL bf16 inputs to both sides and are tight (fp32 outputs 2e-5).
  gradients                   compared for a LINEAR functional of the logits (same upstream gradient on both sides: no L1-sign
                              / GIoU kink of the criterion can flip), at d_logits, d_hs, d_memory, d_c5 and globally.
"""
import os

import numpy as np
import pytest
import torch

from oracle import reftr_oracle as O
from oracle.shapes import param_shapes
from oracle.synth import make_inputs
from oracle.weights import formula_state, formula_tensor
from test_model_gpu import rel, to_cuda

pytestmark = pytest.mark.gpu

# values measured on the build the thresholds were set on (MI355X, ROCm 7.2; printed again by every run)
MEASURED = {
    "single": dict(boxes=1.29e-3, logits_q=5.9e-3, c5_q=8.7e-3, memory_q=4.3e-3, loss=2.4e-3, total=2.1e-4),
    "multi": dict(boxes=1.22e-3, logits_q=6.9e-3, c5_q=8.7e-3, memory_q=4.3e-3, loss=2.4e-3, total=2.1e-4),
}


def build_full(masks=False):
    from reftr_amd.models import layout as L
    from reftr_amd.models.criterion import CriterionVGMultiPhrase, CriterionVGOnePhraseSeg
    from reftr_amd.models.reftr_transformer import RefTR
    ocfg = O.Cfg(masks=True, aux_loss=False) if masks else O.Cfg()
    cfg = L.ModelConfig(masks=True) if masks else L.ModelConfig()
    P = formula_state(param_shapes(ocfg))
    model = RefTR(cfg, device="cuda", aux_loss=not masks)
    model.load_state_dict(P, strict=True)
    model.eval()
    if masks:
        crit = CriterionVGOnePhraseSeg(O.weight_dict(ocfg), ["masks", "boxes"])
    else:
        crit = CriterionVGMultiPhrase(O.weight_dict(ocfg), ["boxes"])
    return model, crit, P, ocfg


def oracle_run(P, samples, targets, ocfg, q, functional=None):
    """Oracle forward + backward.  `functional`: fixed tensor W -> the scalar is sum(logits * W) instead of the criterion's
    total loss.  Returns (outputs, losses, total, parameter gradients, gradients w.r.t. logits / hs / memory / c5)."""
    names = [k for k in P if O.is_trainable(k) and torch.is_floating_point(P[k])]
    Pq = {k: v.clone() for k, v in P.items()}
    leaves = {k: Pq[k].requires_grad_(True) for k in names}
    o = O.reftr_forward(Pq, samples, ocfg, q=q)
    losses = O.criterion(o, targets)
    tot = O.total_loss(losses, O.weight_dict(ocfg))
    scalar = tot if functional is None else (o["logits"] * functional).sum()
    inter = [o["logits"], o["hs"], o["memory"], o["c5"]]
    allg = torch.autograd.grad(scalar, [leaves[k] for k in names] + inter, allow_unused=True)
    grads = {k: (g if g is not None else torch.zeros_like(P[k])) for k, g in zip(names, allg[:len(names)])}
    return o, losses, tot, grads, allg[len(names):], names


def hip_scalar_backward(model, scalar):
    model.store.flat_g.zero_()
    scalar.backward()
    torch.cuda.synchronize()


@pytest.mark.parametrize("kind", ["single", "multi"])
def test_cfg1_full_depth_forward_and_losses_vs_oracle(hip, kind):
    """configs[0]: 320x320, batch 2, L = 40, 12 + 6 + 6 layers (reftr_transformer.py:159-297, criterion.py:113-202)."""
    n_phrase = 3 if kind == "multi" else 0
    samples, targets = make_inputs("e2e_" + kind, B=2, H=320, W=320, L=40, n_phrase=n_phrase)
    model, crit, P, ocfg = build_full()
    s, tg = to_cuda(samples, targets)
    with torch.no_grad():
        out = model(s)
        ld = crit(out, tg)
        total = sum(ld[k] * crit.weight_dict[k] for k in ld if k in crit.weight_dict)
    sv = model._saved
    got = {}
    with torch.no_grad():
        for q in (False, True):
            o = O.reftr_forward(P, samples, ocfg, q=q)
            losses = O.criterion(o, targets)
            tot = O.total_loss(losses, O.weight_dict(ocfg))
            Bn, C, h, w = o["c5"].shape
            tag = "_q" if q else ""
            got["c5" + tag] = rel(sv["c5"].view(Bn, h, w, C).permute(0, 3, 1, 2), o["c5"])
            got["memory" + tag] = rel(sv["memory"].view(Bn, -1, 256).transpose(0, 1), o["memory"])
            got["logits" + tag] = rel(out["pred_logits"], o["logits"])
            got["boxes" + tag] = rel(out["pred_logits"].sigmoid(), o["logits"].sigmoid())
            got["loss" + tag] = max(abs(float(ld[k]) - float(losses[k])) / max(abs(float(losses[k])), 1e-6) for k in losses)
            got["total" + tag] = abs(float(total) - float(tot)) / float(tot)
            if not q:
                assert np.array_equal(out["phrase_mask"].cpu().numpy(), o["phrase_mask"].numpy())       # exact
    print(f"\n[cfg1 {kind}] " + "  ".join(f"{k}={v:.2e}" for k, v in got.items()))
    m = MEASURED[kind]
    assert got["boxes"] < 1.5 * m["boxes"], got                 # the north star's output quantity: ~1e-3 relative
    assert got["loss"] < 1.5 * m["loss"] and got["total"] < 1e-3, got
    assert got["logits_q"] < 1.5 * m["logits_q"], got           # bf16-flip noise floor (module docstring)
    assert got["c5_q"] < 1.5 * m["c5_q"] and got["memory_q"] < 1.5 * m["memory_q"], got
    # both oracle modes are equally far away: the distance is rounding-flip noise, not a systematic difference
    assert 0.6 < got["logits"] / got["logits_q"] < 1.6, got


def test_cfg2_size_full_depth_forward_vs_oracle_and_its_order_floor(hip):
    """configs[1]'s image size, element-wise: 640 x 640, batch 2, L = 40, 12 + 6 + 6 layers against the
    q=True oracle -- AND against the oracle's own floor: the same q=True forward with every contraction accumulated in fp64
    (`O.accumulate_fp64`: identical operands and rounding points, another summation order).  The HIP path must sit within
    1.5 x that floor on every stage: it has no systematic term beyond what two summation orders of the same bf16-operand
    computation already differ by (profiles/r03_noise_floor_*.json: logits 6-7.5e-3, boxes 1.3-1.6e-3 at 320 x 320)."""
    samples, targets = make_inputs("e2e_single", B=2, H=640, W=640, L=40)
    model, crit, P, ocfg = build_full()
    s, tg = to_cuda(samples, targets)
    with torch.no_grad():
        out = model(s)
        sv = model._saved
        o = O.reftr_forward(P, samples, ocfg, q=True)
        with O.accumulate_fp64():
            o2 = O.reftr_forward(P, samples, ocfg, q=True)
    Bn, C, h, w = o["c5"].shape
    mine = {"c5": sv["c5"].view(Bn, h, w, C).permute(0, 3, 1, 2), "memory": sv["memory"].view(Bn, -1, 256).transpose(0, 1),
            "logits": out["pred_logits"], "boxes": out["pred_logits"].sigmoid()}
    ref = {"c5": o["c5"], "memory": o["memory"], "logits": o["logits"], "boxes": o["logits"].sigmoid()}
    alt = {"c5": o2["c5"], "memory": o2["memory"], "logits": o2["logits"], "boxes": o2["logits"].sigmoid()}
    got = {k: rel(mine[k], ref[k]) for k in ref}
    floor = {k: rel(alt[k], ref[k]) for k in ref}
    print("\n[cfg2 size 640x640 B=2] HIP vs q-oracle " + "  ".join(f"{k}={v:.2e}" for k, v in got.items())
          + " | order floor " + "  ".join(f"{k}={v:.2e}" for k, v in floor.items()))
    assert np.array_equal(out["phrase_mask"].cpu().numpy(), o["phrase_mask"].numpy())
    for k in ref:
        assert got[k] < 1.5 * floor[k], (k, got[k], floor[k])
    assert got["boxes"] < 2.4e-3, got


def test_cfg2_exact_B8_forward_and_losses_vs_oracle(hip, parity_table):
    """configs[1] at its EXACT batch: 640 x 640, B = 8, L = 40, 12 + 6 + 6 layers, one fp32 oracle forward
    (no rounding mirror, ~10 s of CPU): boxes, every loss term, the total, `phrase_mask` (exact).  The gates are the B = 2 test's
    (bf16 operands against an fp32 reference: boxes 1.3e-3 measured -- north_star's 1e-3 on LOGITS is not met, see README)."""
    samples, targets = make_inputs("e2e_single", B=8, H=640, W=640, L=40)
    model, crit, P, ocfg = build_full()
    s, tg = to_cuda(samples, targets)
    with torch.no_grad():
        out = model(s)
        ld = crit(out, tg)
        total = sum(ld[k] * crit.weight_dict[k] for k in ld if k in crit.weight_dict)
        o = O.reftr_forward(P, samples, ocfg, q=False)
        losses = O.criterion(o, targets)
        tot = O.total_loss(losses, O.weight_dict(ocfg))
    assert np.array_equal(out["phrase_mask"].cpu().numpy(), o["phrase_mask"].numpy())          # exact
    got = {"boxes": rel(out["pred_logits"].sigmoid(), o["logits"].sigmoid()), "logits": rel(out["pred_logits"], o["logits"]),
           "loss": max(abs(float(ld[k]) - float(losses[k])) / max(abs(float(losses[k])), 1e-6) for k in losses),
           "total": abs(float(total) - float(tot)) / float(tot)}
    print("\n[cfg2 EXACT 640x640 B=8] HIP vs fp32 oracle " + "  ".join(f"{k}={v:.2e}" for k, v in got.items()))
    gates = {"boxes": 2.4e-3, "logits": 1.1e-2, "loss": 5e-3, "total": 1e-3}
    for k, g in gates.items():
        parity_table("cfg2_exact_B8", k + " vs fp32 oracle", got[k], g)
        assert got[k] < g, (k, got)


def test_cfg5_exact_size_r101_800_elementwise_vs_oracle(hip, parity_table):
    """configs[4] at its image size, element-wise: ResNet-101, 800 x 800, B = 2, L = 40, 16 phrase slots
    (S = 40 + 625 rows), one fp32 oracle forward: boxes, losses, `phrase_mask`."""
    from reftr_amd.models import layout as L
    from reftr_amd.models.criterion import CriterionVGMultiPhrase
    from reftr_amd.models.reftr_transformer import RefTR
    ocfg = O.Cfg(resnet_layers=(3, 4, 23, 3))
    cfg = L.ModelConfig(resnet_layers=(3, 4, 23, 3))
    P = formula_state(param_shapes(ocfg))
    model = RefTR(cfg, device="cuda", aux_loss=True)
    model.load_state_dict(P, strict=True)
    model.eval()
    crit = CriterionVGMultiPhrase(O.weight_dict(ocfg), ["boxes"])
    samples, targets = make_inputs("cfg5_arch", B=2, H=800, W=800, L=40, n_phrase=16, Lp=12)
    s, tg = to_cuda(samples, targets)
    with torch.no_grad():
        out = model(s)
        ld = crit(out, tg)
        o = O.reftr_forward(P, samples, ocfg, q=False)
        losses = O.criterion(o, targets)
    assert np.array_equal(out["phrase_mask"].cpu().numpy(), o["phrase_mask"].numpy())
    got = {"boxes": rel(out["pred_logits"].sigmoid(), o["logits"].sigmoid()), "logits": rel(out["pred_logits"], o["logits"]),
           "loss": max(abs(float(ld[k]) - float(losses[k])) / max(abs(float(losses[k])), 1e-6) for k in losses)}
    print("\n[cfg5 EXACT SIZE R101 800x800 B=2 P=16] HIP vs fp32 oracle " + "  ".join(f"{k}={v:.2e}" for k, v in got.items()))
    gates = {"boxes": 3e-3, "logits": 1.5e-2, "loss": 5e-3}
    for k, g in gates.items():
        parity_table("cfg5_exact_800_R101", k + " vs fp32 oracle", got[k], g)
        assert got[k] < g, (k, got)


def test_cfg5_architecture_r101_long_sentence_16_phrases_vs_oracle_and_its_order_floor(hip):
    """configs[4]'s ARCHITECTURE element-wise (VERDICT r02 'weak' item 2: ResNet-101 and the long-sequence attention had only
    property tests): ResNet-101 (3, 4, 23, 3), L = 90 tokens, 16 phrase slots with ragged validity (Lp = 22), 12 + 6 + 6 layers, at
    384 x 384 (S = 90 + 144 rows; the 800 x 800 run of tests/test_fullsize_gpu.py covers the size) against the q=True oracle and
    against the oracle's own fp32-order / fp64-accumulation floor, plus the losses against the fp32 oracle."""
    from reftr_amd.models import layout as L
    from reftr_amd.models.criterion import CriterionVGMultiPhrase
    from reftr_amd.models.reftr_transformer import RefTR
    ocfg = O.Cfg(resnet_layers=(3, 4, 23, 3))
    cfg = L.ModelConfig(resnet_layers=(3, 4, 23, 3))
    P = formula_state(param_shapes(ocfg))
    model = RefTR(cfg, device="cuda", aux_loss=True)
    model.load_state_dict(P, strict=True)
    model.eval()
    crit = CriterionVGMultiPhrase(O.weight_dict(ocfg), ["boxes"])
    samples, targets = make_inputs("cfg5_arch", B=2, H=384, W=384, L=90, n_phrase=16, Lp=22)
    s, tg = to_cuda(samples, targets)
    with torch.no_grad():
        out = model(s)
        ld = crit(out, tg)
        sv = model._saved
        o = O.reftr_forward(P, samples, ocfg, q=True)
        with O.accumulate_fp64():
            o2 = O.reftr_forward(P, samples, ocfg, q=True)
        of = O.reftr_forward(P, samples, ocfg, q=False)
        losses = O.criterion(of, targets)
    assert np.array_equal(out["phrase_mask"].cpu().numpy(), o["phrase_mask"].numpy())           # 16 slots, ragged: exact
    Bn, C, h, w = o["c5"].shape
    mine = {"c5": sv["c5"].view(Bn, h, w, C).permute(0, 3, 1, 2), "memory": sv["memory"].view(Bn, -1, 256).transpose(0, 1),
            "logits": out["pred_logits"], "boxes": out["pred_logits"].sigmoid()}
    got = {k: rel(mine[k], (o[k] if k != "boxes" else o["logits"].sigmoid())) for k in mine}
    floor = {k: rel((o2[k] if k != "boxes" else o2["logits"].sigmoid()), (o[k] if k != "boxes" else o["logits"].sigmoid())) for k in mine}
    rl = max(abs(float(ld[k]) - float(losses[k])) / max(abs(float(losses[k])), 1e-6) for k in losses)
    print("\n[cfg5 architecture R101 L=90 P=16 @384] HIP vs q-oracle " + "  ".join(f"{k}={v:.2e}" for k, v in got.items())
          + " | order floor " + "  ".join(f"{k}={v:.2e}" for k, v in floor.items()) + f" | worst loss vs fp32 oracle {rl:.2e}")
    for k in mine:
        assert got[k] < 1.5 * floor[k], (k, got[k], floor[k])
    assert got["boxes"] < 3e-3 and rl < 5e-3, (got, rl)


# measured: d_logits exact (the functional's own gradient); d_hs / d_memory / d_c5 and the global parameter gradient sit on the
# ReLU-mask-flip floor: the 0.5 % forward noise flips ~1 % of the ReLU decisions of the 3-layer box head / FFNs / bottlenecks,
# and a flipped unit contributes its whole gradient -> sqrt(1 %) = 10 % in L2 already at d_hs, one ReLU MLP below the logits.
# The sharp statement about the backward pass is the directional-derivative test below.
# Gates of the gradient test = mean + 3 sigma (cosine: mean - 3 sigma) of the ORACLE'S OWN order floor: the same q=True gradient in 17
# other summation orders (16 permuted chunk orders + fp64 accumulation; oracle/noise_floor.py GRAD_FLOOR=samples ->
# profiles/r04_noise_floor_gradients.json).  Round 3 gated at 1.5 x the HIP path's own measured value, i.e. from the implementation
# under test.
GRAD_FLOOR_GATE = None


def _grad_floor_gate(kind):
    import json
    global GRAD_FLOOR_GATE
    if GRAD_FLOOR_GATE is None:
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "grad_floor_gates.json")
        GRAD_FLOOR_GATE = json.load(open(path))
    return GRAD_FLOOR_GATE[kind]


@pytest.mark.parametrize("kind", ["single", "multi"])
def test_cfg1_full_depth_backward_of_a_linear_functional_vs_oracle(hip, kind):
    """Mid-network gradients at full depth against the q=True oracle, with the SAME upstream gradient on both sides
    (scalar = sum(logits * W), W fixed): d_hs (head), d_memory (decoder + query encoder), d_c5 (encoder + input_proj +
    GroupNorm) and every parameter gradient.  What is left is the bf16 noise of the backward GEMM operands and the
    ReLU-mask flips of the forward noise."""
    n_phrase = 3 if kind == "multi" else 0
    samples, targets = make_inputs("e2e_" + kind, B=2, H=320, W=320, L=40, n_phrase=n_phrase)
    model, crit, P, ocfg = build_full()
    model._debug = True
    s, tg = to_cuda(samples, targets)
    out = model(s)
    W = formula_tensor("functional.w", tuple(out["pred_logits"].shape), 1.0, bf16=False)
    if n_phrase:            # padded phrase slots carry no loss in the reference either: keep them out of the functional
        W = W * out["phrase_mask"].view(1, *out["phrase_mask"].shape, 1, 1).cpu().float()
    hip_scalar_backward(model, (out["pred_logits"] * W.cuda()).sum())
    o, _, _, grads, ig, names = oracle_run(P, samples, targets, ocfg, q=True, functional=W)
    d = model._dbg
    Bn, C, h, w = o["c5"].shape
    got = {
        "d_logits": rel(d["dlogits"], ig[0]),
        "d_hs": rel(d["dhs"].view(ig[1].shape), ig[1]),
        "d_memory": rel(d["dmem"].view(Bn, -1, 256).transpose(0, 1), ig[2]),
        "d_c5": rel(d["g_c5"].view(Bn, h, w, C).permute(0, 3, 1, 2), ig[3] * (o["c5"] > 0)),
    }
    a = torch.cat([model.store.G[k].detach().double().cpu().reshape(-1) for k in names])
    b = torch.cat([grads[k].double().reshape(-1) for k in names])
    got["glob"] = float((a - b).norm() / b.norm())
    got["cos"] = float((a * b).sum() / (a.norm() * b.norm()))
    per = sorted(rel(model.store.G[k], grads[k]) for k in names if float(grads[k].norm()) > 1e-6 * float(b.norm()))
    got["median_tensor"] = per[len(per) // 2]
    print(f"\n[cfg1 {kind} functional backward] " + "  ".join(f"{k}={v:.3e}" for k, v in got.items()))
    m = _grad_floor_gate(kind)
    print("    gates (floor mean + 3 sigma): " + "  ".join(f"{k}={v:.3e}" for k, v in m.items()))
    assert got["d_logits"] < 1e-6, got
    assert got["d_hs"] < m["d_hs"], (got, m)
    assert got["d_memory"] < m["d_memory"] and got["d_c5"] < m["d_c5"], (got, m)
    assert got["glob"] < m["params"] and got["cos"] > m["params_cosine"], (got, m)


def test_cfg1_full_depth_backward_is_the_derivative_of_the_forward(hip):
    """Self-consistency at full depth, free of any oracle noise: the eval-mode forward is bit-deterministic, so central
    differences of the scalar s(theta) = sum(logits * W) along a direction v measure the directional derivative of the function
    the kernels compute; it must equal <flat_g, v> from the hand-written backward.  v = the normalised gradient of each
    learning-rate group (main | ResNet | BERT), step sized for a 2 % change of s (far above the bf16 staircase of the forward),
    averaged over independent operand dithers (one draw carries +-5 % of rounding noise, see MEASURED_FD).  Together with the
    forward parity above this bounds the gradients: a wrong-but-consistent layer would show up in the forward, a wrong backward
    formula (a missing term, a factor) shows up here at >= 10 %; the per-kernel tests are the tight ones (2e-5)."""
    from reftr_amd.models import layout as L
    samples, targets = make_inputs("e2e_single", B=2, H=320, W=320, L=40)
    model, crit, P, ocfg = build_full()
    # The formula weights sit exactly ON the bf16 grid: a perturbation below half an ulp would never reach the bf16 GEMM
    # operands (the forward would not move at all for the Linear weights).  Dither every fp32 master uniformly over exactly
    # +-1 ulp of its bf16 value (an interval that holds exactly two rounding boundaries whatever the mantissa -- a dither
    # relative to the VALUE would hold 2 boundaries per 1..2 ulps and inflate the difference quotient by E[2/m] = 1.39), so
    # that operand rounding acts as unbiased stochastic rounding of the perturbation.
    st = model.store
    s, tg = to_cuda(samples, targets)
    base = st.flat_p.clone()
    _, ex = torch.frexp(base)                            # |p| = m * 2^ex, m in [0.5, 1): bf16 ulp = 2^(ex - 8)
    ulp = torch.ldexp(torch.ones_like(base), ex - 8) * (base != 0)
    W = None
    runs = []
    for seed in FD_SEEDS:                                # independent dithers: the rounding noise of one draw is a few per cent
        gen = torch.Generator(device="cuda").manual_seed(seed)
        st.flat_p.copy_(base + (torch.rand(base.shape, generator=gen, device="cuda") * 2.0 - 1.0) * ulp)
        model.mark_dirty()
        out = model(s)
        if W is None:
            W = formula_tensor("functional.w", tuple(out["pred_logits"].shape), 1.0, bf16=False).cuda()
        scale = float((out["pred_logits"].detach() * W).abs().sum())
        hip_scalar_backward(model, (out["pred_logits"] * W).sum())
        g, p0 = st.flat_g.clone(), st.flat_p.clone()
        res = {}
        for name, grp in (("main", L.GROUP_MAIN), ("resnet", L.GROUP_BACKBONE), ("bert", L.GROUP_BERT)):
            a, b = st.group_range[grp]
            gn = float(g[a:b].double().norm())
            v = torch.zeros_like(g); v[a:b] = g[a:b] / gn
            eps = 0.02 * scale / gn
            vals = []
            for sign in (1.0, -1.0):
                st.flat_p.copy_(p0 + sign * eps * v)
                model.mark_dirty()
                with torch.no_grad():
                    vals.append(float((model(s)["pred_logits"].double() * W.double()).sum()))
            fd = (vals[0] - vals[1]) / (2 * eps)
            res[name] = (fd - gn) / gn                   # signed: the rounding noise of a draw has no preferred sign
        runs.append(res)
    st.flat_p.copy_(base); model.mark_dirty()
    mean = {k: sum(r[k] for r in runs) / len(runs) for k in runs[0]}
    print("\n[cfg1 directional derivative] (fd - <g,v>) / <g,v> per dither: " +
          "  ".join(f"{k}: " + ", ".join(f"{r[k]:+.2e}" for r in runs) + f" -> mean {mean[k]:+.2e}" for k in mean))
    for k in mean:
        assert abs(mean[k]) < MEASURED_FD[k] * 1.5, (k, mean[k], [r[k] for r in runs])
        assert all(abs(r[k]) < MEASURED_FD_ONE[k] * 1.5 for r in runs), (k, [r[k] for r in runs])


FD_SEEDS = tuple(range(5, 17))
# Measured (both attention-backward variants, 8-12 dithers, several builds): single draws scatter by +-4..10 % (the stochastic rounding of ~10^8 bf16
# operands under a 2 % step), their mean sits at -3 % (main), -0.4 % (resnet), -0.0..-3 % (bert).  The negative mean is the bf16
# noise of the gradient itself, not of the formula: g = g_true + e with e unbiased and |e| / |g| ~ 0.17 (the rel-L2 the oracle
# comparison above measures) gives <g_true, g> / |g| = |g| / (1 + |e|^2 / |g_true|^2) = |g| (1 - 0.03).  Until round 3 this test used
# ONE dither draw whose main-group value happened to be 3.8e-3; a kernel change that moved the gradient by 1e-3 (fused attention
# backward) moved that single draw to 4e-2, which is how the noise was found.
MEASURED_FD = {"main": 5.3e-2, "resnet": 5.3e-2, "bert": 5.3e-2}          # |mean over FD_SEEDS|: -2.5e-2 .. -3.3e-2 seen, sigma of the mean 1.3e-2
MEASURED_FD_ONE = {"main": 1.2e-1, "resnet": 1.2e-1, "bert": 1.2e-1}      # a single dither: sigma 4.5e-2 (worst of 40 draws 9.6e-2)


# ---------------------------------------------------------------------------------------------- configs[3]: RefTRSeg
def box_masks(targets, sizes):
    """Deterministic bool masks [1, h, w]: the pixels inside the image's first target box."""
    out = []
    for t, (h, w) in zip(targets, sizes):
        cx, cy, bw, bh = [float(v) for v in t["boxes"][0]]
        ys = torch.arange(h)[:, None].float() / h; xs = torch.arange(w)[None, :].float() / w
        m = (xs >= cx - bw / 2) & (xs < cx + bw / 2) & (ys >= cy - bh / 2) & (ys < cy + bh / 2)
        out.append(dict(t, masks=m[None]))
    return out


MEASURED_SEG = dict(boxes=2.45e-3, pred_masks=1.65e-2, mask_att=4.4e-3, loss=1.7e-3)


def test_cfg4_seg_full_depth_vs_oracle(hip):
    """RefTRSeg at full depth, 320x320, batch 2 (reftr_segmentation.py:76-175, 314-337) against the fp32 oracle."""
    samples, targets = make_inputs("seg_full", B=2, H=320, W=320, L=40)
    targets = box_masks(targets, [(320, 320), (240, 213)])          # make_inputs: image 1 is valid on 3/4 x 2/3 of the frame
    model, crit, P, ocfg = build_full(masks=True)
    s, tg = to_cuda(samples, targets)
    with torch.no_grad():
        out = model(s)
        ld = crit(out, tg)
        o = O.reftr_forward(P, samples, ocfg, q=False)
        losses = O.criterion(o, targets)
    got = {
        "boxes": rel(out["pred_boxes"], o["pred_boxes"]),
        "pred_masks": rel(out["pred_masks"], o["pred_masks"]),
        "mask_att": rel(out["mask_att"], o["mask_att"]),
        "loss": max(abs(float(ld[k]) - float(losses[k])) / max(abs(float(losses[k])), 0.1) for k in losses),
    }
    print("\n[cfg4 320x320 full depth] " + "  ".join(f"{k}={v:.2e}" for k, v in got.items())
          + "  losses " + " ".join(f"{k}={float(ld[k]):.5f}/{float(losses[k]):.5f}" for k in losses))
    assert out["pred_masks"].shape == o["pred_masks"].shape == (2, 1, 80, 80)
    for k, v in MEASURED_SEG.items():
        assert got[k] < 1.5 * v, (k, got)


@pytest.mark.parametrize("size,B", [(320, 2), (640, 2)])
def test_cfg4_seg_outputs_vs_q_oracle_and_its_order_floor(hip, size, B):
    """the RES outputs -- `pred_masks` (1.65e-2 against the fp32 oracle) and `mask_att` (4.4e-3) -- had no
    order-floor demonstration and no element-wise check above 320 x 320.  Same protocol as the REC outputs
    (test_cfg2_size_...): RefTRSeg at full depth against the q=True oracle (the HIP path's bf16 rounding points,
    reftr_segmentation.py:76-175) AND against that oracle's own floor -- the same q=True forward with every contraction
    accumulated in fp64 and in two permuted chunk orders (`O.accumulate_fp64`, `O.accumulate_permuted`): identical operands and
    rounding points, other summation orders.  `pred_boxes` and `mask_att` must sit within 1.5 x the LARGEST of the three floor
    samples; `pred_masks` within 1.6 x (measured 1.53 x at both sizes): the mask head's stride-8 / stride-4 FPN inputs are the
    layer2 / layer1 outputs, where IEEE-fp32 summation orders of the oracle differ by only 2.7e-3 / 3.4e-4 while the MFMA path is
    5.4e-3 / 2.1e-3 away -- each convolution alone matches the oracle to 2e-6 ... 1.4e-4 on identical inputs (rare bf16 flips,
    profiles/r04_seg_floor_stages.txt), the matrix pipe's accumulation is simply not one of the IEEE orders the floor samples.
    At 640 x 640 this is the element-wise check of configs[3]'s image size."""
    samples, targets = make_inputs("seg_full", B=B, H=size, W=size, L=40)
    model, crit, P, ocfg = build_full(masks=True)
    s, tg = to_cuda(samples, targets)
    keys = ("pred_boxes", "pred_masks", "mask_att")
    with torch.no_grad():
        out = model(s)
        o = O.reftr_forward(P, samples, ocfg, q=True)
        alts = []
        for ctx in (O.accumulate_fp64(), O.accumulate_permuted(3), O.accumulate_permuted(11)):
            with ctx:
                alts.append(O.reftr_forward(P, samples, ocfg, q=True))
    got = {k: rel(out[k], o[k]) for k in keys}
    floors = [{k: rel(a[k], o[k]) for k in keys} for a in alts]
    floor = {k: max(f[k] for f in floors) for k in keys}
    print(f"\n[cfg4 RES {size}x{size} B={B}] HIP vs q-oracle " + "  ".join(f"{k}={v:.2e}" for k, v in got.items())
          + " | order floor (max of fp64 / 2 permuted) " + "  ".join(f"{k}={v:.2e}" for k, v in floor.items())
          + " | samples " + " ".join("/".join(f"{f[k]:.1e}" for f in floors) for k in keys))
    assert out["pred_masks"].shape == o["pred_masks"].shape == (B, 1, size // 4, size // 4)
    for k in keys:
        assert got[k] < (1.6 if k == "pred_masks" else 1.5) * floor[k], (k, got[k], floor[k])
    assert got["pred_masks"] < 1.6e-2, got
    # the thresholded mask (what PostProcessSegm consumes): decisions that differ from the oracle's sit where its own orders disagree
    dec = float(((out["pred_masks"].cpu() > 0) != (o["pred_masks"] > 0)).float().mean())
    dec_floor = max(float(((a["pred_masks"] > 0) != (o["pred_masks"] > 0)).float().mean()) for a in alts)
    print(f"    mask decisions that differ: HIP {dec:.2e}, between the oracle's own orders {dec_floor:.2e}")
    assert dec <= 2.0 * dec_floor + 1e-4, (dec, dec_floor)


@pytest.fixture(scope="module")
def cfg4():
    """configs[3] at its full size: RefTRSeg, ResNet-50, 640x640, batch 8, L = 40 (no aux loss, reftr_segmentation.py:52)."""
    import bench
    from reftr_amd.models import layout as L
    from reftr_amd.models.criterion import CriterionVGOnePhraseSeg
    from reftr_amd.models.reftr_transformer import RefTR
    from reftr_amd.util.misc import NestedTensor
    cfg = L.ModelConfig(masks=True)
    model = RefTR(cfg, device="cuda", aux_loss=False)
    wd = {"loss_giou": 1.0, "loss_bbox": 1.0, "loss_dice": 1.0, "loss_mask": 1.0}
    crit = CriterionVGOnePhraseSeg(wd, ["masks", "boxes"])
    samples, targets = bench.synth_batch(8, 640, 640, 40, "cuda", 1234)
    g = torch.Generator().manual_seed(99)
    for b, t in enumerate(targets):                      # Bernoulli(0.3) masks [1, H, W] on the valid region
        wv = 480 if b % 2 == 0 else 640
        t["masks"] = torch.rand(1, 640, wv, generator=g) < 0.3
    s = {k: v.cuda() for k, v in samples.items() if k not in ("img", "img_mask")}
    s["img"] = NestedTensor(samples["img"].cuda(), samples["img_mask"].cuda())
    tg = [{k: v.cuda() for k, v in t.items()} for t in targets]
    return model, crit, s, tg, cfg


def _seg_init(model):
    model.reset_parameters(seed=0)
    torch.manual_seed(11)
    model.store.P["bbox_embed.layers.2.weight"].normal_(0, 0.05)
    model.mark_dirty()
    model.eval()


def test_cfg4_fullsize_known_answer_determinism_and_batch_equivariance(cfg4):
    from test_fullsize_gpu import closed_form_losses, take
    model, crit, s, tg, cfg = cfg4
    model.reset_parameters(seed=0)                       # the reference's init: last bbox layer zero (reftr_transformer.py:131-132)
    model.eval()
    with torch.no_grad():
        out = model(s)
        ld = crit(out, tg)
    assert out["pred_boxes"].shape == (8, 1, 1, 4) and "aux_outputs" not in out
    assert out["pred_masks"].shape == (8, 1, 160, 160) and out["mask_att"].shape == (8, 8, 20, 20)
    assert bool((out["pred_boxes"] == 0.5).all())
    lb, lg = closed_form_losses(tg)
    assert abs(float(ld["loss_bbox"]) - lb) < 2e-6 * max(1.0, lb) and abs(float(ld["loss_giou"]) - lg) < 2e-6 * max(1.0, lg)
    # the attention map is a joint softmax over heads x h x w of each query (reftr_segmentation.py:205): sums to 1, zero on padding
    att = out["mask_att"].float()
    assert float((att.flatten(1).sum(1) - 1).abs().max()) < 1e-3
    pad = s["img"].mask[:, ::32, ::32][:, :20, :20]
    assert float((att * pad[:, None].float()).abs().max()) == 0.0
    assert np.isfinite(float(ld["loss_mask"])) and 0 < float(ld["loss_dice"]) < 1
    # determinism + batch equivariance of the eval forward (no atomics; samples independent)
    _seg_init(model)
    with torch.no_grad():
        a = model(s); a = {k: a[k].clone() for k in ("pred_logits", "pred_masks", "mask_att")}
        b = model(s)
        for k in a:
            assert torch.equal(a[k], b[k]), k
        perm = [3, 0, 7, 1, 6, 2, 5, 4]
        sp, _ = take(s, tg, perm)
        c = model(sp)
    assert float(a["pred_masks"].std()) > 1e-3
    assert float((c["pred_masks"] - a["pred_masks"][perm]).abs().max()) <= 1e-4 * float(a["pred_masks"].abs().max())
    assert float((c["pred_logits"] - a["pred_logits"][:, perm]).abs().max()) <= 1e-5 * float(a["pred_logits"].abs().max())


def test_cfg4_fullsize_shard_additivity_and_train_step(cfg4):
    """The data-parallel contract for the multitask loss: all four losses are normalised by the GLOBAL number of boxes
    (reftr_segmentation.py:316-337, criterion.py:176-180), so grad(batch of 8) = (grad(first 4) + grad(last 4)) / 2."""
    from reftr_amd.engine_vg import train_step
    from reftr_amd.models import layout as L
    from reftr_amd.optim import FusedAdamW
    from test_fullsize_gpu import _grads, take
    model, crit, s, tg, cfg = cfg4
    _seg_init(model)
    g_full, l_full = _grads(model, crit, s, tg)
    s0, t0 = take(s, tg, [0, 1, 2, 3]); s1, t1 = take(s, tg, [4, 5, 6, 7])
    g0, l0 = _grads(model, crit, s0, t0)
    g1, l1 = _grads(model, crit, s1, t1)
    assert torch.isfinite(g_full).all() and float(g_full.norm()) > 0
    assert abs(l_full - 0.5 * (l0 + l1)) < 2e-5 * abs(l_full)
    want = 0.5 * (g0 + g1)
    st = model.store
    for grp in (L.GROUP_MAIN, L.GROUP_MASK, L.GROUP_BACKBONE, L.GROUP_BERT):
        b, e = st.group_range[grp]
        if e <= b:
            continue
        x, y = g_full[b:e].double(), want[b:e].double()
        relerr = float((x - y).norm() / y.norm()); cos = float((x @ y) / (x.norm() * y.norm()))
        print(f"[cfg4 shard additivity] group {grp}: rel {relerr:.2e} cos {cos:.7f}")
        assert relerr < 3e-3 and cos > 0.99999, (grp, relerr, cos)
    # one REC+RES training step at full size (dropout on, clip 0.1, AdamW): finite, weights move by at most lr
    opt = FusedAdamW(model, lr=1e-4, lr_backbone=1e-5, weight_decay=1e-4)
    model.train()
    before = model.store.flat_p.clone()
    lv, _, _, gn = train_step(model, crit, s, tg, opt, None, 0.1)
    torch.cuda.synchronize()
    assert np.isfinite(lv) and torch.isfinite(model.store.flat_g).all() and torch.isfinite(model.store.flat_p).all()
    step = (model.store.flat_p - before).abs().max()
    assert float(gn) > 0 and 0 < float(step) <= 1.01e-4 + 1e-8 * float(before.abs().max()) + 1e-7
