"""GPU end-to-end parity of the HIP RefTR path (through the C ABI) against
  (1) the golden vectors minted from the imported reference (tests/golden/e2e_*.npz, steps_single.npz) and
  (2) the CPU oracle on the same seeded inputs and formula weights.

Tolerances (bf16 GEMM operands / bf16 backbone activations vs the reference's fp32): every
assert sits at <= 1.5 x the value measured on MI355X (the `TOL` table below; the measured values are printed by each run).
  gradients: noise-limited by ReLU-mask / L1-sign flips that a 0.5 % forward perturbation triggers, so they are
  checked globally (rel-L2 <= 0.25, cosine >= 0.97) and for the head tensors tightly (<= 2e-2).
Integer / bool outputs (phrase_mask) must be exact.
"""
import os

import numpy as np
import pytest
import torch

from oracle import reftr_oracle as O
from oracle.shapes import param_shapes
from oracle.synth import make_inputs
from oracle.weights import formula_state

pytestmark = pytest.mark.gpu
# <= 1.5 x the values measured on MI355X for the reduced-depth fixture (printed by every run): boxes 1.52e-3 (single) /
# 0.99e-3 (multi), single loss terms <= 1.58e-3, total 9.3e-4, raw logits vs the q=True oracle 6.9e-3 (bf16 rounding-flip
# floor, see tests/test_parity_fullsize_gpu.py)
TOL = {"boxes": 2.3e-3, "loss": 2.4e-3, "total": 1.4e-3, "logits_q": 1.0e-2}
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rel(a, b):
    a = torch.as_tensor(a).detach().float().cpu(); b = torch.as_tensor(b).detach().float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def build(small=True):
    from reftr_amd.models import layout as L
    from reftr_amd.models.criterion import CriterionVGMultiPhrase
    from reftr_amd.models.reftr_transformer import RefTR
    if small:
        ocfg = O.Cfg(enc_layers=2, dec_layers=2, bert=O.BertCfg(layers=2))
        cfg = L.ModelConfig(enc_layers=2, dec_layers=2, bert=L.BertConfig(layers=2))
    else:
        ocfg, cfg = O.Cfg(), L.ModelConfig()
    P = formula_state(param_shapes(ocfg))
    model = RefTR(cfg, device="cuda")
    model.load_state_dict(P, strict=True)
    crit = CriterionVGMultiPhrase(O.weight_dict(ocfg), ["boxes"])
    return model, crit, P, ocfg


def to_cuda(samples, targets):
    from reftr_amd.util.misc import NestedTensor
    s = {k: v.cuda() for k, v in samples.items() if k not in ("img", "img_mask")}
    s["img"] = NestedTensor(samples["img"].cuda(), samples["img_mask"].cuda())
    return s, [{k: v.cuda() for k, v in t.items()} for t in targets]


@pytest.mark.parametrize("tag,n_phrase", [("e2e_single", 0), ("e2e_multi", 3)])
def test_forward_backward_vs_reference_golden(hip, tag, n_phrase):
    g = np.load(os.path.join(GOLD, tag + ".npz"))
    model, crit, P, ocfg = build(small=True)
    model.eval()                     # dropout off (the golden vectors were minted in eval mode)
    samples, targets = make_inputs(tag, B=2, H=96, W=128, L=12, n_phrase=n_phrase)
    s, tg = to_cuda(samples, targets)
    out = model(s)
    boxes = torch.cat([torch.stack([a["pred_boxes"] for a in out["aux_outputs"]]), out["pred_boxes"][None]])
    rb = rel(boxes, g["boxes"])
    assert np.array_equal(out["phrase_mask"].cpu().numpy(), g["phrase_mask"])            # exact
    ld = crit(out, tg)
    rl = max(abs(float(v) - float(g["loss." + k])) / max(1.0, abs(float(g["loss." + k]))) for k, v in ld.items())
    total = sum(ld[k] * crit.weight_dict[k] for k in ld if k in crit.weight_dict)
    rt = abs(float(total) - float(g["total_loss"])) / float(g["total_loss"])
    print(f"\n[{tag} vs reference golden] boxes {rb:.2e}  worst loss {rl:.2e}  total {rt:.2e}")
    assert rb < TOL["boxes"], rb
    assert rl < TOL["loss"] and rt < TOL["total"], (rl, rt)
    model.store.flat_g.zero_()
    total.backward()
    G = model.store.G
    assert rel(G["bbox_embed.layers.2.weight"], g["grad_bbox2_w"]) < 2e-2
    names = [str(n) for n in g["grad_names"]]
    norms = np.array([float(G[k].norm()) for k in names])
    big = g["grad_norms"] > 1e-4                 # key-bias gradients are exactly 0 in exact arithmetic
    ratio = norms[big] / g["grad_norms"][big]
    assert 0.8 < np.median(ratio) < 1.2 and ratio.min() > 0.4 and ratio.max() < 2.5
    for key, gk in (("img_backbone.0.body.layer4.2.conv3.weight", "grad_l4_conv3"),
                    ("img_backbone.0.body.layer2.0.conv1.weight", "grad_l2_conv1"),
                    ("lang_backbone.encoder.layer.0.attention.self.query.weight", "grad_bert_q0"),
                    ("vl_transformer.level_embed", "grad_level_embed")):
        ref = torch.from_numpy(g[gk])
        got = G[key].detach().float().cpu()[: ref.shape[0]] if ref.dim() > 1 and key != "vl_transformer.level_embed" else G[key].detach().float().cpu()
        cos = float((got.reshape(-1) * ref.reshape(-1)).sum() / (got.norm() * ref.norm() + 1e-30))
        assert cos > 0.95, (key, cos)


def test_gradients_vs_oracle_global(hip):
    model, crit, P, ocfg = build(small=True)
    model.eval()
    samples, targets = make_inputs("e2e_single", B=2, H=96, W=128, L=12)
    names = [k for k in P if O.is_trainable(k)]
    Pq = {k: v.clone() for k, v in P.items()}
    leaves = {k: Pq[k].requires_grad_(True) for k in names}
    o = O.reftr_forward(Pq, samples, ocfg, q=True)
    tot = O.total_loss(O.criterion(o, targets), O.weight_dict(ocfg))
    ref = dict(zip(names, torch.autograd.grad(tot, [leaves[k] for k in names])))
    s, tg = to_cuda(samples, targets)
    out = model(s)
    rlq = rel(out["pred_logits"], o["logits"])
    print(f"\n[small fixture] logits vs q=True oracle {rlq:.2e}")
    assert rlq < TOL["logits_q"], rlq
    ld = crit(out, tg)
    total = sum(ld[k] * crit.weight_dict[k] for k in ld if k in crit.weight_dict)
    model.store.flat_g.zero_()
    total.backward()
    a = torch.cat([model.store.G[k].detach().float().cpu().reshape(-1) for k in names])
    b = torch.cat([ref[k].reshape(-1) for k in names])
    assert float((a - b).norm() / b.norm()) < 0.25
    assert float((a * b).sum() / (a.norm() * b.norm())) > 0.97


def test_three_training_steps(hip):
    """engine_vg.py:40-72 loop body (clip 0.1 + AdamW, lr 1e-4 / 1e-5 / 1e-5).

    Step 0 is compared with the reference's own run (tests/golden/steps_single.npz).  From step 1 on the fp32
    reference trajectory is NOT reproducible by any bf16-operand implementation on this fixture: the formula
    weights sit exactly on the bf16 grid, so an Adam update of lr = 1e-5..1e-4 (<< half a bf16 ulp of a 0.05-sized
    weight) vanishes when the operand is re-rounded, although it is kept in the fp32 master weights.  The oracle
    in bf16-point mode (q=True) models exactly that, and is the trajectory the HIP path has to follow."""
    from reftr_amd.engine_vg import train_step
    from reftr_amd.optim import FusedAdamW
    g = np.load(os.path.join(GOLD, "steps_single.npz"))
    model, crit, P, ocfg = build(small=True)
    model.eval()
    opt = FusedAdamW(model, lr=1e-4, lr_backbone=1e-5, weight_decay=1e-4)
    samples, targets = make_inputs("e2e_single", B=2, H=96, W=128, L=12)
    s, tg = to_cuda(samples, targets)
    Pq = {k: v.clone() for k, v in P.items()}
    state = {}
    for it in range(3):
        _, ref_loss, ref_gn, _ = O.train_step(Pq, samples, targets, ocfg, state, it + 1, max_norm=0.1, train=False, q=True)
        loss_value, _, _, gnorm = train_step(model, crit, s, tg, opt, None, max_norm=0.1)
        if it == 0:
            e_l = abs(loss_value - float(g["loss"][0])) / float(g["loss"][0]); e_g = abs(float(gnorm) - float(g["gnorm"][0])) / float(g["gnorm"][0])
            print(f"\n[step 0 vs reference golden] loss rel {e_l:.2e}  grad-norm rel {e_g:.2e}")
            assert e_l < TOL["total"] and e_g < 1e-2, (e_l, e_g)       # measured 9.3e-4 / 6.7e-3
        # the two bf16-point trajectories drift apart (zero-true-gradient biases take sign-of-noise Adam steps)
        assert abs(loss_value - ref_loss) < (2e-2 if it < 2 else 6e-2) * ref_loss, (it, loss_value, ref_loss)
        assert abs(float(gnorm) - ref_gn) < 0.25 * ref_gn, (it, float(gnorm), ref_gn)
    sd = model.state_dict()
    w0 = P["bbox_embed.layers.2.weight"]
    upd = sd["bbox_embed.layers.2.weight"].cpu() - w0
    assert rel(upd, Pq["bbox_embed.layers.2.weight"] - w0) < 0.3
    # fp32 master weights keep sub-ulp updates that the bf16 operands cannot show
    k = "img_backbone.0.body.layer4.2.conv3.weight"
    d = (sd[k].cpu() - P[k]).abs()
    assert 1e-6 < float(d.mean()) < 5e-5


@pytest.mark.parametrize("two_phase", [False, True, "all"])
def test_captured_step_matches_eager(hip, two_phase, monkeypatch):
    """CapturedTrainStep (hipGraph replay; two_phase = the data-parallel schedule with backward split in
    [everything but the ResNet | the ResNet]) must walk the same trajectory as the eager loop body: same kernels, so
    only the summation order of the backward's atomics separates them.

    Step 1 is compared tightly (loss, gradients, updated weights).  Later steps only loosely: a 1-ulp difference in
    the weights flips single bf16 roundings in the next forward, and on this fixture (formula weights that sit on the
    bf16 grid) one such flip moves the loss by 4e-4 at step 2 and up to 1.2e-2 at step 3 -- two EAGER runs differ by that
    much from each other in ~20 % of the runs (benchmarks/debug_graph_vs_eager.py)."""
    from reftr_amd.engine_vg import CapturedTrainStep, train_step
    from reftr_amd.optim import FusedAdamW
    # "same kernels" holds for the launched head: round 5's rt_head_loss (captured direct-loss path only) sums its products in another
    # order than the M <= 16 kernel the eager loop uses on this fixture's 16 rows; it is compared in tests/test_qregion_gpu.py
    monkeypatch.setenv("REFTR_HEAD_FUSE", "0")
    samples, targets = make_inputs("e2e_single", B=2, H=96, W=128, L=12)
    s, tg = to_cuda(samples, targets)
    runs = []
    for mode in ("eager", "graph"):
        model, crit, P, ocfg = build(small=True)
        model.eval()
        opt = FusedAdamW(model, lr=1e-4, lr_backbone=1e-5, weight_decay=1e-4)
        if mode == "graph":
            p0, m0, v0 = model.store.flat_p.clone(), opt.m.clone(), opt.v.clone()
            # "all": one graph per exchange boundary of the data-parallel schedule (main | BERT | layer4 | rest of the ResNet)
            cap = CapturedTrainStep(model, crit, opt, 0.1, s, tg, warmup=1, force_two_phase=two_phase is True,
                                    force_phases=model.BOUNDARIES if two_phase == "all" else None)
            assert (cap.g_bb is not None) == bool(two_phase)
            assert len(cap.g_seg if two_phase else []) == {False: 0, True: 1, "all": 3}[two_phase]
            # capture warm-up steps moved the weights: restore the initial state before comparing trajectories
            cap.reset_pending()
            model.store.flat_p.copy_(p0); opt.m.copy_(m0); opt.v.copy_(v0); opt.step_dev.zero_(); opt.step_count = 0
            model.mark_dirty(full=True)
            # AdamW of step i at the head of step i+1: one graph (single process) or across the segment graphs (data parallel)
            assert cap.deferred == (not two_phase) and cap.deferred_dp == bool(two_phase)
        losses, norms, first = [], [], None
        for it in range(3):
            if mode == "eager":
                lv, _, _, gn = train_step(model, crit, s, tg, opt, None, max_norm=0.1)
            else:
                l, _, gn = cap(s, tg)
                lv = float(l)
            losses.append(lv); norms.append(float(gn))
            if it == 0:
                torch.cuda.synchronize()
                g_now = model.store.flat_g.clone()
                if mode == "graph":
                    cap.flush()                         # the deferred update of this step, applied now instead
                first = (g_now, model.store.flat_p.clone())
        if mode == "graph":
            cap.flush()
        runs.append((losses, norms, model.store.flat_p.clone(), first))
    (l0, n0, p_e, f_e), (l1, n1, p_g, f_g) = runs
    assert abs(l0[0] - l1[0]) < 1e-6 * abs(l0[0]) and abs(n0[0] - n1[0]) < 1e-5 * n0[0], (l0, l1, n0, n1)
    assert rel(f_g[0], f_e[0]) < 1e-6 and rel(f_g[1], f_e[1]) < 1e-7        # gradients / weights after the first step
    for a, b, tol in zip(l0[1:], l1[1:], (1e-3, 2e-2)):
        assert abs(a - b) < tol * abs(a), (l0, l1)
    for a, b in zip(n0, n1):
        assert abs(a - b) < 3e-2 * abs(a), (n0, n1)
    assert rel(p_g, p_e) < 2e-4             # after the trajectories may have separated (see above); step 1 is the tight check


@pytest.mark.parametrize("tag,n_phrase", [("e2e_single", 0), ("e2e_multi", 3)])
def test_default_captured_step_vs_oracle_and_golden(hip, tag, n_phrase, monkeypatch, parity_table):
    """The configuration bench.py TIMES -- CapturedTrainStep with NO environment overrides (deferred AdamW, direct loss path,
    rt_head_loss, rt_qenc_fwd / rt_qenc_bwd) -- directly beside the oracle and the reference's golden vectors (VERDICT r05 item 2;
    until round 6 it was covered transitively: fused launches == launched chains == oracle).  Three replays of the loop body
    (/root/reference/engine_vg.py:40-72, criterion.py:113-202): step-0 loss / gradient norm against tests/golden/steps_single.npz
    (the reference's own run), every step's loss / norm against O.train_step in bf16-point mode at test_three_training_steps'
    gates, and the update of bbox_embed.layers.2.weight after the three steps against the oracle's."""
    from reftr_amd.engine_vg import CapturedTrainStep
    from reftr_amd.optim import FusedAdamW
    for v in ("REFTR_HEAD_FUSE", "REFTR_QFUSE", "REFTR_LOSS_DIRECT", "REFTR_DEFER_OPT", "REFTR_FUSED_TOTAL"):
        monkeypatch.delenv(v, raising=False)
    g = np.load(os.path.join(GOLD, "steps_single.npz"))
    model, crit, P, ocfg = build(small=True)
    model.eval()                                             # the golden steps were minted with dropout off
    opt = FusedAdamW(model, lr=1e-4, lr_backbone=1e-5, weight_decay=1e-4)
    samples, targets = make_inputs(tag, B=2, H=96, W=128, L=12, n_phrase=n_phrase)
    s, tg = to_cuda(samples, targets)
    p0, m0, v0 = model.store.flat_p.clone(), opt.m.clone(), opt.v.clone()
    cap = CapturedTrainStep(model, crit, opt, 0.1, s, tg, warmup=1)
    # this IS the timed configuration: one graph, deferred update, direct loss, fused head
    assert cap.deferred and not cap.two_phase and cap._direct_loss_ok()
    assert model._saved.get("head") is not None              # rt_head_loss ran inside the captured forward
    cap.reset_pending()                                      # capture warm-up moved the weights: back to the formula state
    model.store.flat_p.copy_(p0); opt.m.copy_(m0); opt.v.copy_(v0); opt.step_dev.zero_(); opt.step_count = 0
    model.mark_dirty(full=True)
    Pq = {k: v.clone() for k, v in P.items()}
    state = {}
    for it in range(3):
        _, ref_loss, ref_gn, _ = O.train_step(Pq, samples, targets, ocfg, state, it + 1, max_norm=0.1, train=False, q=True)
        l, ld, gn = cap(s, tg)
        loss_value, gnorm = float(l), float(gn)
        if it == 0 and tag == "e2e_single":
            e_l = abs(loss_value - float(g["loss"][0])) / float(g["loss"][0]); e_g = abs(gnorm - float(g["gnorm"][0])) / float(g["gnorm"][0])
            print(f"\n[default captured step 0 vs reference golden] loss rel {e_l:.2e}  grad-norm rel {e_g:.2e}")
            parity_table("default_captured_step", "step-0 loss vs reference golden", e_l, TOL["total"])
            parity_table("default_captured_step", "step-0 grad norm vs reference golden", e_g, 1e-2)
            assert e_l < TOL["total"] and e_g < 1e-2, (e_l, e_g)
        e_l = abs(loss_value - ref_loss) / ref_loss; e_g = abs(gnorm - ref_gn) / ref_gn
        print(f"[default captured {tag} step {it} vs q-oracle] loss rel {e_l:.2e}  grad-norm rel {e_g:.2e}")
        parity_table("default_captured_step", f"{tag} step-{it} loss vs q-oracle", e_l, 2e-2 if it < 2 else 6e-2)
        assert e_l < (2e-2 if it < 2 else 6e-2), (it, loss_value, ref_loss)
        assert e_g < 0.25, (it, gnorm, ref_gn)
        # the loss dict the loop logs: every term against the oracle's, first step (later ones follow the drifting trajectory)
        if it == 0:
            with torch.no_grad():
                o = O.reftr_forward({k: v.clone() for k, v in P.items()}, samples, ocfg, q=False)
                ref_ld = O.criterion(o, targets)
            worst = max(abs(float(ld[k]) - float(ref_ld[k])) / max(1.0, abs(float(ref_ld[k]))) for k in ref_ld)
            parity_table("default_captured_step", f"{tag} step-0 worst loss term vs fp32 oracle", worst, TOL["loss"])
            assert sorted(ld) == sorted(ref_ld) and worst < TOL["loss"], worst
    cap.flush()                                              # the deferred update of the last step
    sd = model.state_dict()
    w0 = P["bbox_embed.layers.2.weight"]
    upd = sd["bbox_embed.layers.2.weight"].cpu() - w0
    e_u = rel(upd, Pq["bbox_embed.layers.2.weight"] - w0)
    parity_table("default_captured_step", f"{tag} bbox_embed.layers.2.weight update after 3 steps vs q-oracle", e_u, 0.3)
    assert e_u < 0.3, e_u


def test_default_captured_step_at_configs1_exact_size_vs_oracle(hip, monkeypatch, parity_table):
    """The same default captured configuration at configs[1]'s exact size (R50, 640 x 640, B = 8, L = 40, 12 + 6 + 6 layers,
    the shape bench.py times), one replay against one fp32 oracle forward: total and every loss term of the step the graph ran."""
    from reftr_amd.engine_vg import CapturedTrainStep
    from reftr_amd.optim import FusedAdamW
    for v in ("REFTR_HEAD_FUSE", "REFTR_QFUSE", "REFTR_LOSS_DIRECT", "REFTR_DEFER_OPT", "REFTR_FUSED_TOTAL"):
        monkeypatch.delenv(v, raising=False)
    model, crit, P, ocfg = build(small=False)
    model.eval()
    opt = FusedAdamW(model, lr=1e-4, lr_backbone=1e-5, weight_decay=1e-4)
    samples, targets = make_inputs("e2e_single", B=8, H=640, W=640, L=40)
    s, tg = to_cuda(samples, targets)
    p0, m0, v0 = model.store.flat_p.clone(), opt.m.clone(), opt.v.clone()
    cap = CapturedTrainStep(model, crit, opt, 0.1, s, tg, warmup=1)
    assert cap.deferred and cap._direct_loss_ok() and model._saved.get("head") is not None
    cap.reset_pending()
    model.store.flat_p.copy_(p0); opt.m.copy_(m0); opt.v.copy_(v0); opt.step_dev.zero_(); opt.step_count = 0
    model.mark_dirty(full=True)
    l, ld, gn = cap(s, tg)
    torch.cuda.synchronize()
    with torch.no_grad():
        o = O.reftr_forward(P, samples, ocfg, q=False)
        ref_ld = O.criterion(o, targets)
        tot = float(O.total_loss(ref_ld, O.weight_dict(ocfg)))
    e_t = abs(float(l) - tot) / tot
    worst = max(abs(float(ld[k]) - float(ref_ld[k])) / max(abs(float(ref_ld[k])), 1e-6) for k in ref_ld)
    print(f"\n[default captured step, configs[1] exact] total rel {e_t:.2e}  worst loss term {worst:.2e}  grad norm {float(gn):.4f}")
    parity_table("default_captured_step", "configs[1] exact: total vs fp32 oracle", e_t, 1e-3)
    parity_table("default_captured_step", "configs[1] exact: worst loss term vs fp32 oracle", worst, 5e-3)
    assert e_t < 1e-3 and worst < 5e-3, (e_t, worst)
    assert np.isfinite(float(gn)) and float(gn) > 0


@pytest.mark.parametrize("aux", [True, False])
def test_direct_loss_path_equals_the_autograd_path(hip, monkeypatch, aux):
    """CapturedTrainStep's direct loss path (losses and d total / d logits from ONE rt_box_loss launch, targets prepared beside
    the step head, no sigmoid, no autograd node) against the criterion + autograd path it replaces: the loss values are the
    same kernel on the same logits -- bit-identical -- and the gradients differ by the order of the backward's atomics only."""
    from reftr_amd.engine_vg import CapturedTrainStep
    from reftr_amd.optim import FusedAdamW
    samples, targets = make_inputs("e2e_single", B=2, H=96, W=128, L=12)
    s, tg = to_cuda(samples, targets)
    out = {}
    monkeypatch.setenv("REFTR_HEAD_FUSE", "0")          # the launched head: same rt_box_loss launch on both sides (round 5's fused head
    for direct in ("0", "1"):                           # has its own comparison, tests/test_qregion_gpu.py)
        monkeypatch.setenv("REFTR_LOSS_DIRECT", direct)
        model, crit, P, ocfg = build(small=True)
        model.eval()
        if not aux:
            model.aux_loss = False
        opt = FusedAdamW(model, lr=1e-4, lr_backbone=1e-5, weight_decay=1e-4)
        p0, m0, v0 = model.store.flat_p.clone(), opt.m.clone(), opt.v.clone()
        cap = CapturedTrainStep(model, crit, opt, 0.1, s, tg, warmup=1)
        assert cap._direct_loss_ok() == (direct == "1")
        cap.reset_pending()
        model.store.flat_p.copy_(p0); opt.m.copy_(m0); opt.v.copy_(v0); opt.step_dev.zero_(); opt.step_count = 0
        model.mark_dirty(full=True)
        l, ld, gn = cap(s, tg)
        torch.cuda.synchronize()
        out[direct] = (float(l), {k: float(v) for k, v in ld.items()}, float(gn), model.store.flat_g.clone())
    a, b = out["0"], out["1"]
    assert a[0] == b[0] and a[1] == b[1], (a[:2], b[:2])
    assert sorted(a[1]) == (sorted(["loss_bbox", "loss_giou"] + [f"loss_{n}_{i}" for n in ("bbox", "giou") for i in range(ocfg.dec_layers - 1)])
                            if aux else ["loss_bbox", "loss_giou"])
    assert abs(a[2] - b[2]) < 1e-5 * a[2]
    assert rel(b[3], a[3]) < 1e-6


@pytest.mark.parametrize("two_phase", [False, True])
def test_captured_deferred_update_follows_the_lr_schedule(hip, two_phase):
    """Deferred schedule: the update of iteration i is applied at the head of replay i+1 but must use the learning rates
    of iteration i (device words synced after each replay), with no re-capture when the schedule moves."""
    from reftr_amd.engine_vg import CapturedTrainStep, train_step
    from reftr_amd.optim import FusedAdamW
    samples, targets = make_inputs("e2e_single", B=2, H=96, W=128, L=12)
    s, tg = to_cuda(samples, targets)
    sched = [(1e-4, 1e-5), (5e-5, 5e-6), (2e-5, 2e-6), (2e-5, 2e-6)]
    key = "bbox_embed.layers.1.weight"
    res = []
    for mode in ("eager", "graph"):
        model, crit, P, ocfg = build(small=True)
        model.eval()
        opt = FusedAdamW(model, lr=sched[0][0], lr_backbone=sched[0][1], weight_decay=1e-4)
        if mode == "graph":
            p0, m0, v0 = model.store.flat_p.clone(), opt.m.clone(), opt.v.clone()
            cap = CapturedTrainStep(model, crit, opt, 0.1, s, tg, warmup=1, force_two_phase=two_phase)
            assert cap.deferred == (not two_phase) and cap.deferred_dp == bool(two_phase)
            cap.reset_pending()
            model.store.flat_p.copy_(p0); opt.m.copy_(m0); opt.v.copy_(v0); opt.step_dev.zero_(); opt.step_count = 0
            graph_id = id(cap.g_fb)
        for lr, lrb in sched:
            for g in opt.param_groups:
                g["lr"] = lrb if g["group_id"] in (1, 2) else lr          # backbone / BERT groups vs main / mask
            if mode == "eager":
                train_step(model, crit, s, tg, opt, None, max_norm=0.1)
            else:
                cap(s, tg)
        if mode == "graph":
            assert id(cap.g_fb) == graph_id and cap._pending
            sd = model.state_dict()                      # applies the pending (4th) update first
            assert not cap._pending
            w = sd[key].float().cpu()
        else:
            w = model.state_dict()[key].float().cpu()
        res.append(w - P[key])
    d_e, d_g = res
    assert float(d_e.abs().max()) > 1e-5                 # the weight moved: sum of four Adam steps
    # a schedule applied one iteration late (or early) would change the total displacement by tens of per cent
    # (measured 3.5e-2: the run-to-run trajectory noise of the fixture)
    assert rel(d_g, d_e) < 0.1, rel(d_g, d_e)


def test_train_one_epoch_replays_graphs_and_matches_the_eager_loop(hip, monkeypatch):
    """engine_vg.train_one_epoch (the reference's entry point, engine_vg.py:22-75): fixed-shape batches are replayed from
    hipGraphs (one capture per shape, whose warm-up updates are taken back), variable shapes fall back to eager launches;
    meters and weights follow the purely eager loop."""
    from reftr_amd.engine_vg import train_one_epoch
    from reftr_amd.optim import FusedAdamW
    b1 = make_inputs("e2e_single", B=2, H=96, W=128, L=12)
    b2 = make_inputs("steps_single", B=2, H=96, W=128, L=12)
    loader = [b1, b2, b1, b2]
    from reftr_amd.util.misc import NestedTensor

    def cpu_batches():
        out = []
        for samples, targets in loader:
            s = {k: v for k, v in samples.items() if k not in ("img", "img_mask")}
            s["img"] = NestedTensor(samples["img"], samples["img_mask"])
            out.append((s, targets))
        return out

    res = {}
    # the launched head on both sides: rt_head_loss (replayed path only) rounds like the launches but sums in another order than the
    # M <= 16 kernel this fixture's 4 head rows go through, and the fixture amplifies any first-step difference (see
    # test_captured_step_matches_eager); fused vs launched head is compared in tests/test_qregion_gpu.py
    monkeypatch.setenv("REFTR_HEAD_FUSE", "0")
    for graph in ("1", "0"):
        monkeypatch.setenv("REFTR_TRAIN_GRAPH", graph)
        model, crit, P, ocfg = build(small=True)
        opt = FusedAdamW(model, lr=1e-4, lr_backbone=1e-5, weight_decay=1e-4)
        sched = torch.optim.lr_scheduler.StepLR(opt, step_size=2, gamma=0.5)
        model.cfg.dropout = 0.0                                          # a deterministic trajectory to compare
        stats = train_one_epoch(model, crit, cpu_batches(), opt, sched, torch.device("cuda"), 0, max_norm=0.1)
        assert model.training and opt.step_count == 4
        caps = getattr(model, "_captured_steps", {})
        assert len(caps) == (1 if graph == "1" else 0)
        res[graph] = (stats, model.state_dict()["bbox_embed.layers.1.weight"].float().cpu() - P["bbox_embed.layers.1.weight"],
                      opt.param_groups[0]["lr"])
    (sg, dg, lrg), (se, de, lre) = res["1"], res["0"]
    assert lrg == lre == 2.5e-5
    assert set(sg) == set(se) and {"loss", "loss_bbox", "loss_giou_unscaled", "lr", "grad_norm"} <= set(sg)
    assert abs(sg["loss"] - se["loss"]) < 2e-2 * se["loss"] and abs(sg["grad_norm"] - se["grad_norm"]) < 5e-2 * se["grad_norm"]
    assert rel(dg, de) < 0.1                 # four Adam steps under the same schedule (trajectory noise of the fixture: 3.5e-2)


def test_captured_forward_is_bit_identical_to_eager(hip):
    """engine_vg.CapturedForward: the eval forward replayed from a hipGraph returns exactly the eager forward's tensors
    (no atomics in the forward), also after the weights changed and for a second batch of the same shape."""
    from reftr_amd.engine_vg import CapturedForward
    model, crit, P, ocfg = build(small=True)
    model.eval()
    s1, _ = to_cuda(*make_inputs("e2e_single", B=2, H=96, W=128, L=12))
    s2, _ = to_cuda(*make_inputs("steps_single", B=2, H=96, W=128, L=12))
    fwd = CapturedForward(model)
    with torch.no_grad():
        for s in (s1, s2, s1):
            a = {k: v.clone() for k, v in model(s).items() if torch.is_tensor(v)}
            b = fwd(s)
            for k in a:
                assert torch.equal(a[k], b[k]), k
        assert len(fwd.graphs) == 1
        model.store.P["bbox_embed.layers.2.weight"].mul_(1.5); model.mark_dirty()
        b = fwd(s2)["pred_boxes"].clone()            # rebuilds the bf16 operands the graph reads
        model.mark_dirty()
        assert torch.equal(model(s2)["pred_boxes"], b)


def test_folded_self_attention_dropout_is_the_attention_kernels(hip, monkeypatch):
    """Train mode, one query per image: the decoder self-attention's per-head probability dropout folded into the V
    projection's epilogue (and into the backward-data product of out_proj) draws the same masks as the attention kernel it
    replaces -- same seed site, hash index b * H + h -- so loss and gradients agree to bf16 rounding of the value path."""
    res = {}
    for fold in ("0", "1"):
        monkeypatch.setenv("REFTR_FOLD_SA", fold)
        model, crit, P, ocfg = build(small=True)
        assert model.net.fold_sa == (fold == "1")
        model.train()
        s, tg = to_cuda(*make_inputs("e2e_single", B=2, H=96, W=128, L=12))
        out = model(s)
        ld = crit(out, tg)
        total = sum(ld[k] * crit.weight_dict[k] for k in ld if k in crit.weight_dict)
        model.store.flat_g.zero_()
        total.backward()
        torch.cuda.synchronize()
        G = model.store.G
        res[fold] = (float(total.detach()), out["pred_logits"].detach().clone(),
                     {k: G[k].clone() for k in ("vl_transformer.decoder.layers.0.self_attn.out_proj.weight",
                                                "vl_transformer.decoder.layers.1.self_attn.in_proj_weight",
                                                "bbox_embed.layers.2.weight")},
                     [r["fold"] for r in model._saved["dec"]])
    (l0, y0, g0, f0), (l1, y1, g1, f1) = res["0"], res["1"]
    assert f0 == [False, False] and f1 == [True, True]
    # (same masks -- tests/test_ops_gpu.py checks the zero pattern exactly; what is left is one bf16 rounding of the value path)
    assert abs(l0 - l1) < 5e-3 * abs(l0) and rel(y1, y0) < 2e-2
    for k in g0:                                 # gradients: the usual flip-limited agreement
        a, b = g1[k].flatten().double(), g0[k].flatten().double()
        assert rel(g1[k], g0[k]) < 0.2 and float(a @ b / (a.norm() * b.norm())) > 0.98, (k, rel(g1[k], g0[k]))
    E = 256                                      # the q / k rows of in_proj get no gradient at all in either path
    assert float(g1["vl_transformer.decoder.layers.1.self_attn.in_proj_weight"][:2 * E].abs().max()) == 0.0


def test_dropout_train_mode_runs_and_is_reproducible(hip):
    model, crit, P, ocfg = build(small=True)
    model.train()
    samples, targets = make_inputs("e2e_single", B=2, H=96, W=128, L=12)
    s, tg = to_cuda(samples, targets)
    model.seed_dev.fill_(10)
    a = model(s)["pred_logits"].detach().clone()
    model.seed_dev.fill_(10)
    b = model(s)["pred_logits"].detach().clone()
    c = model(s)["pred_logits"].detach().clone()           # next step: the device seed word advanced -> new masks
    assert torch.equal(a, b) and not torch.equal(a, c)
    assert torch.isfinite(c).all()
    ld = crit(model(s), tg)
    sum(ld.values()).backward()
    assert torch.isfinite(model.store.flat_g).all()


def test_state_dict_roundtrip_and_errors(hip):
    model, crit, P, ocfg = build(small=True)
    sd = model.state_dict()
    assert set(sd.keys()) == set(P.keys())
    for k in ("img_backbone.0.body.layer2.0.conv2.weight", "lang_backbone.encoder.layer.1.attention.self.key.weight"):
        assert torch.equal(sd[k].cpu(), P[k]) and sd[k].is_contiguous()
    with pytest.raises(AssertionError):
        crit({"pred_boxes": None}, [])


def test_checkpoint_resume_is_exact(hip, tmp_path):
    """Save after two steps in the reference's checkpoint format (main_vg.py:377-384), resume in a fresh model / optimizer
    / scheduler (main_vg.py:306-337), take one more step: same loss and parameters as the uninterrupted run."""
    from reftr_amd.checkpoint import load_checkpoint, save_checkpoint
    from reftr_amd.engine_vg import train_step
    from reftr_amd.optim import FusedAdamW
    samples, targets = make_inputs("e2e_single", B=2, H=96, W=128, L=12)
    s, tg = to_cuda(samples, targets)

    def fresh():
        model, crit, P, ocfg = build(small=True)
        model.eval()
        opt = FusedAdamW(model, lr=1e-4, lr_backbone=1e-5, weight_decay=1e-4)
        sch = torch.optim.lr_scheduler.StepLR(opt, 100)
        return model, crit, opt, sch
    model, crit, opt, sch = fresh()
    for _ in range(2):
        train_step(model, crit, s, tg, opt, sch, max_norm=0.1)
    path = tmp_path / "checkpoint.pth"
    save_checkpoint(path, model, opt, sch, epoch=3, best_val_acc=0.5)
    ck = torch.load(path, map_location="cpu", weights_only=False)
    assert set(ck) == {"model", "optimizer", "lr_scheduler", "epoch", "args", "best_val_acc"}
    assert set(ck["optimizer"]) == {"state", "param_groups"} and len(ck["optimizer"]["param_groups"]) == 4
    l_ref = train_step(model, crit, s, tg, opt, sch, max_norm=0.1)[0]
    model2, crit2, opt2, sch2 = fresh()
    start, best, missing, unexpected = load_checkpoint(str(path), model2, opt2, sch2)
    assert start == 4 and best == 0.5 and not missing and not unexpected and opt2.step_count == 2
    l_res = train_step(model2, crit2, s, tg, opt2, sch2, max_norm=0.1)[0]
    assert abs(l_res - l_ref) < 1e-5 * abs(l_ref)
    assert rel(model2.store.flat_p, model.store.flat_p) < 1e-6


def test_learned_position_embedding_vs_reference_golden(hip):
    """--position_embedding learned (position_encoding.py:59-84): forward, loss and the gradients of the two embedding tables
    against the fixture minted from the reference built with that flag."""
    from reftr_amd.models import layout as L
    from reftr_amd.models.criterion import CriterionVGMultiPhrase
    from reftr_amd.models.reftr_transformer import RefTR, build_config
    g = np.load(os.path.join(GOLD, "e2e_learned_pos.npz"))
    ocfg = O.Cfg(enc_layers=2, dec_layers=2, bert=O.BertCfg(layers=2), pos_learned=True)
    cfg = L.ModelConfig(enc_layers=2, dec_layers=2, bert=L.BertConfig(layers=2), pos_learned=True)
    P = formula_state(param_shapes(ocfg))
    model = RefTR(cfg, device="cuda")
    model.load_state_dict(P, strict=True)
    model.eval()
    crit = CriterionVGMultiPhrase(O.weight_dict(ocfg), ["boxes"])
    samples, targets = make_inputs("e2e_learned", B=2, H=96, W=128, L=12, n_phrase=3)
    s, tg = to_cuda(samples, targets)
    out = model(s)
    assert rel(out["pred_logits"].sigmoid(), g["boxes"]) < 5e-3
    ld = crit(out, tg)
    total = sum(ld[k] * crit.weight_dict[k] for k in ld if k in crit.weight_dict)
    assert abs(float(total) - float(g["total_loss"])) < 5e-3 * float(g["total_loss"])
    model.store.flat_g.zero_()
    total.backward()
    # position gradients are attention q / k gradients: the noisiest quantity under bf16 operands.  The yardstick is the oracle
    # with the HIP path's rounding points (q=True) against the same fp32 fixture.
    Pq = {k: v.clone() for k, v in P.items()}
    keys = ("img_backbone.1.row_embed.weight", "img_backbone.1.col_embed.weight")
    leaves = [Pq[k].requires_grad_(True) for k in keys]
    oq = O.reftr_forward(Pq, samples, ocfg, q=True)
    gq = torch.autograd.grad(O.total_loss(O.criterion(oq, targets), O.weight_dict(ocfg)), leaves)
    for key, name, q_ in zip(("grad_row", "grad_col"), keys, gq):
        mine, ref = model.store.G[name].float().cpu(), torch.from_numpy(g[key])
        floor = rel(q_, ref)
        assert rel(mine, ref) < max(2.0 * floor, 3e-2), (key, rel(mine, ref), floor)
        assert float((mine * ref).sum() / (mine.norm() * ref.norm())) > 0.995
        assert float(mine[4:].abs().sum()) == 0            # 3 x 4 feature map: the other 46 table rows get no gradient
    got, ref = model.store.G["vl_transformer.level_embed"].float().cpu().reshape(-1), torch.from_numpy(g["grad_level_embed"]).reshape(-1)
    assert float((got * ref).sum() / (got.norm() * ref.norm())) > 0.995
    assert rel(model.store.G["bbox_embed.layers.2.weight"], g["grad_bbox2_w"]) < 3e-2


def test_dilation_vs_reference_golden(hip):
    """--dilation (models/modeling/backbone.py:117-125): layer4 at stride 1 with its 3x3 convolutions dilated by 2 from the second
    block on (rt_conv_gemm / rt_conv_wgrad `dil`), c5 at stride 16 -- forward, loss and the convolution gradients of layer4 against
    the fixture minted from the reference built with that flag, with the q-oracle (the HIP path's bf16 rounding points) as the
    yardstick for the gradients."""
    from reftr_amd.models import layout as L
    from reftr_amd.models.criterion import CriterionVGMultiPhrase
    from reftr_amd.models.reftr_transformer import RefTR
    g = np.load(os.path.join(GOLD, "e2e_dilation.npz"))
    ocfg = O.Cfg(enc_layers=2, dec_layers=2, bert=O.BertCfg(layers=2), dilation=True)
    cfg = L.ModelConfig(enc_layers=2, dec_layers=2, bert=L.BertConfig(layers=2), dilation=True)
    P = formula_state(param_shapes(ocfg))
    model = RefTR(cfg, device="cuda")
    model.load_state_dict(P, strict=True)
    model.eval()
    crit = CriterionVGMultiPhrase(O.weight_dict(ocfg), ["boxes"])
    samples, targets = make_inputs("e2e_dilation", B=2, H=96, W=128, L=12)
    s, tg = to_cuda(samples, targets)
    out = model(s)
    assert rel(out["pred_logits"].sigmoid(), g["boxes"]) < 5e-3
    ld = crit(out, tg)
    total = sum(ld[k] * crit.weight_dict[k] for k in ld if k in crit.weight_dict)
    assert abs(float(total) - float(g["total_loss"])) < 5e-3 * float(g["total_loss"])
    model.store.flat_g.zero_()
    total.backward()
    keys = [k[5:] for k in g.files if k.startswith("grad.")]
    Pq = {k: v.clone() for k, v in P.items()}
    leaves = [Pq[k].requires_grad_(True) for k in keys]
    oq = O.reftr_forward(Pq, samples, ocfg, q=True)
    gq = torch.autograd.grad(O.total_loss(O.criterion(oq, targets), O.weight_dict(ocfg)), leaves)
    for name, q_ in zip(keys, gq):
        mine = model.store.G[name].float().cpu()
        ref = torch.from_numpy(g["grad." + name])
        samp = (lambda t: t.reshape(-1)[::97]) if mine.numel() > 100000 else (lambda t: t)
        floor = rel(samp(q_), ref)
        got = rel(samp(mine).reshape(ref.shape), ref)
        assert got < max(2.0 * floor, 3e-2), (name, got, floor)
        a, b, c = samp(mine).reshape(-1), ref.reshape(-1), samp(q_).reshape(-1)
        cos = lambda u, v: float((u * v).sum() / (u.norm() * v.norm()))
        assert 1 - cos(a, b) < max(4.0 * (1 - cos(c, b)), 1e-3), (name, cos(a, b), cos(c, b))     # (1 - cos ~ rel^2 / 2: twice the floor's rel)
        assert abs(float(mine.norm()) / float(g["gnorm." + name]) - 1) < max(2.0 * abs(float(q_.norm()) / float(g["gnorm." + name]) - 1), 2e-2), name


def test_dilation_beyond_one_lds_pass_vs_oracle(hip):
    """--dilation on an image large enough that a head's K / V no longer fit the CU's LDS at once (480 x 480: c5 30 x 30 at stride 16,
    S = 900 + L = 912 tokens > ~830): the attention launches of the encoder and the decoder take the chunked kernels.  Forward
    and selected gradients against the q-oracle (the HIP path's rounding points; pinned to the reference by
    tests/golden/e2e_dilation.npz at the small size)."""
    from reftr_amd.models import layout as L
    from reftr_amd.models.criterion import CriterionVGMultiPhrase
    from reftr_amd.models.reftr_transformer import RefTR
    ocfg = O.Cfg(enc_layers=2, dec_layers=2, bert=O.BertCfg(layers=2), dilation=True)
    cfg = L.ModelConfig(enc_layers=2, dec_layers=2, bert=L.BertConfig(layers=2), dilation=True)
    P = formula_state(param_shapes(ocfg))
    model = RefTR(cfg, device="cuda")
    model.load_state_dict(P, strict=True)
    model.eval()
    crit = CriterionVGMultiPhrase(O.weight_dict(ocfg), ["boxes"])
    samples, targets = make_inputs("e2e_dilation_long", B=2, H=480, W=480, L=12)          # image 1 carries right / bottom padding
    s, tg = to_cuda(samples, targets)
    out = model(s)
    keys = ["vl_transformer.encoder.layers.0.self_attn.in_proj_weight", "vl_transformer.encoder.layers.1.linear1.weight",
            "vl_transformer.decoder.layers.0.multihead_attn.in_proj_weight", "input_proj.0.0.weight",
            "img_backbone.0.body.layer4.2.conv2.weight", "txt_backbone.encoder.layer.1.output.dense.weight"]
    keys = [k for k in keys if k in P]
    assert len(keys) >= 4, [k for k in P if "in_proj" in k][:4]
    Pq = {k: v.clone() for k, v in P.items()}
    leaves = [Pq[k].requires_grad_(True) for k in keys]
    oq = O.reftr_forward(Pq, samples, ocfg, q=True)
    assert rel(out["pred_logits"].sigmoid().reshape(-1), oq["logits"].sigmoid().reshape(-1)) < 5e-3
    assert rel(out["pred_boxes"], oq["pred_boxes"]) < 5e-3
    lq = O.total_loss(O.criterion(oq, targets), O.weight_dict(ocfg))
    gq = torch.autograd.grad(lq, leaves)
    ld = crit(out, tg)
    total = sum(ld[k] * crit.weight_dict[k] for k in ld if k in crit.weight_dict)
    assert abs(float(total) - float(lq)) < 5e-3 * abs(float(lq))
    model.store.flat_g.zero_()
    total.backward()
    # the yardstick: the same q-oracle with every contraction accumulated in fp64 -- identical operands and rounding points, another
    # summation order (tests/test_parity_fullsize_gpu.py); the HIP gradients must sit within 1.5 x of what that alone moves
    P2 = {k: v.clone() for k, v in P.items()}
    leaves2 = [P2[k].requires_grad_(True) for k in keys]
    with O.accumulate_fp64():
        o2 = O.reftr_forward(P2, samples, ocfg, q=True)
        g2 = torch.autograd.grad(O.total_loss(O.criterion(o2, targets), O.weight_dict(ocfg)), leaves2)
    report = []
    for name, q_, f_ in zip(keys, gq, g2):
        mine = model.store.G[name].float().cpu().reshape(q_.shape)
        cos = lambda u, v: float((u.reshape(-1) * v.reshape(-1)).sum() / (u.norm() * v.norm()))       # noqa: E731
        got, floor = rel(mine, q_), rel(f_.float(), q_)
        report.append((name, got, floor, cos(mine, q_), cos(f_.float(), q_)))
    print("\n[--dilation 480x480, S = 912] gradient rel-L2 vs q-oracle (HIP | order floor), cosine (HIP | floor):")
    for r in report:
        print("   %-62s %.3e | %.3e   %.5f | %.5f" % r)
    for name, got, floor, c_m, c_f in report:
        assert got < max(1.5 * floor, 3e-2), (name, got, floor)
        assert 1 - c_m < max(2.25 * (1 - c_f), 1e-3), (name, c_m, c_f)


def test_clip_norm_with_the_bert_share_taken_on_the_language_stream(hip):
    """Single process: the BERT slice's squared gradient norm is reduced on the language stream right after BERT's backward
    (reftr_transformer._backward_gen), the optimizer reads only the rest of the buffer (optim._sqnorm_all): the total equals
    the norm of the whole buffer, in eager steps and with the split switched off."""
    from reftr_amd.engine_vg import train_step
    from reftr_amd.optim import FusedAdamW
    model, crit, P, ocfg = build(small=True)
    model.train()
    samples, targets = make_inputs("e2e_single", B=2, H=96, W=128, L=12)
    s, tg = to_cuda(samples, targets)
    opt = FusedAdamW(model, lr=0.0, lr_backbone=0.0, weight_decay=0.0)
    norms = {}
    assert model._norm_side is False          # off unless an engine that owns backward + clip as one unit switches it on
    for split in (True, False):
        model._norm_side = split
        model.seed_dev.fill_(3)
        _, _, _, gn = train_step(model, crit, s, tg, opt, None, max_norm=0.1)
        torch.cuda.synchronize()
        full = float(model.store.flat_g.double().norm())
        assert abs(float(gn) - full) < 1e-5 * full, (split, float(gn), full)
        if split:
            from reftr_amd.models import layout as L
            b0, b1 = model.store.group_range[L.GROUP_BERT]
            assert abs(float(model._sq_bert) - float(model.store.flat_g[b0:b1].double().pow(2).sum())) < 1e-4 * float(model._sq_bert)
            assert model._norm_split is None                   # consumed by the optimizer
        norms[split] = float(gn)
    assert abs(norms[True] - norms[False]) < 1e-4 * norms[False]
    # the captured single-process step takes the split and switches it off again behind itself
    from reftr_amd.engine_vg import CapturedTrainStep
    model._norm_side = False
    cap = CapturedTrainStep(model, crit, opt, 0.1, s, tg)
    gn = cap(s, tg)[2]
    torch.cuda.synchronize()
    full = float(model.store.flat_g.double().norm())
    assert abs(float(gn) - full) < 1e-5 * full and float(model._sq_bert) > 0
    assert model._norm_side is False and model._norm_split is None
    cap.flush()


def test_roberta_backbone_vs_reference_golden(hip):
    """RoBERTa language backbone (f4): exact integer position ids + the same encoder kernels, against the golden vectors
    minted from the reference with HF RobertaModel."""
    from oracle.synth import roberta_inputs
    from reftr_amd.models import layout as L
    from reftr_amd.models.criterion import CriterionVGMultiPhrase
    from reftr_amd.models.reftr_transformer import RefTR
    g = np.load(os.path.join(GOLD, "e2e_roberta.npz"))
    ocfg = O.Cfg(enc_layers=2, dec_layers=2, bert=O.roberta_cfg(layers=2))
    cfg = L.ModelConfig(enc_layers=2, dec_layers=2, bert=L.roberta_config(layers=2))
    P = formula_state(param_shapes(ocfg))
    model = RefTR(cfg, device="cuda")
    model.load_state_dict(P, strict=True)
    model.eval()
    crit = CriterionVGMultiPhrase(O.weight_dict(ocfg), ["boxes"])
    samples, targets = roberta_inputs()
    s, tg = to_cuda(samples, targets)
    # exact integer check of the position ids
    tok = samples["sentence"].ne(1).int()
    assert torch.equal(hip.roberta_pos_ids(s["sentence"], 1).cpu().long(), (torch.cumsum(tok, 1) * tok).long() + 1)
    out = model(s)
    assert rel(out["pred_logits"].sigmoid(), g["boxes"]) < 5e-3
    ld = crit(out, tg)
    total = sum(ld[k] * crit.weight_dict[k] for k in ld if k in crit.weight_dict)
    assert abs(float(total) - float(g["total_loss"])) < 5e-3 * float(g["total_loss"])
    model.store.flat_g.zero_()
    total.backward()
    gp = model.store.G["lang_backbone.embeddings.position_embeddings.weight"][:20].float().cpu()
    ref = torch.from_numpy(g["grad_pos_emb"])
    assert float((gp * ref).sum() / (gp.norm() * ref.norm())) > 0.97
    assert rel(model.store.G["bbox_embed.layers.2.weight"], g["grad_bbox2_w"]) < 3e-2


@pytest.mark.parametrize("n_phrase", [0, 3])
def test_two_queries_per_phrase_vs_oracle(hip, n_phrase):
    """num_queries_per_phrase = 2 (reftr_transformer.py:60-66, 235-238, 276-280): query rows are phrase-major
    (phrase, q), both queries of a phrase share its validity, the box loss averages over k = n_q predictions per target
    (criterion.py:131-153).  Single-phrase inputs with n_q > 1 crash in the reference (its [B, 1] query mask does not match
    the n_q decoder keys); here the mask is expanded like in the multi-phrase branch -- compared with the oracle, which
    expands it the same way."""
    from reftr_amd.models import layout as L
    from reftr_amd.models.criterion import CriterionVGMultiPhrase
    from reftr_amd.models.reftr_transformer import RefTR
    ocfg = O.Cfg(enc_layers=2, dec_layers=2, bert=O.BertCfg(layers=2), n_q=2)
    cfg = L.ModelConfig(enc_layers=2, dec_layers=2, bert=L.BertConfig(layers=2), n_q=2)
    P = formula_state(param_shapes(ocfg))
    model = RefTR(cfg, device="cuda")
    model.load_state_dict(P, strict=True)
    model.eval()
    crit = CriterionVGMultiPhrase(O.weight_dict(ocfg), ["boxes"])
    samples, targets = make_inputs("e2e_nq2", B=2, H=96, W=128, L=12, n_phrase=n_phrase)
    s, tg = to_cuda(samples, targets)
    out = model(s)
    Pn = max(n_phrase, 1)
    assert out["pred_boxes"].shape == (2, Pn, 2, 4) and out["phrase_mask"].shape == (2, 2 * Pn)
    names = [k for k in P if O.is_trainable(k)]
    Pq = {k: v.clone() for k, v in P.items()}
    leaves = {k: Pq[k].requires_grad_(True) for k in names}
    o = O.reftr_forward(Pq, samples, ocfg, q=False)
    ol = O.criterion(o, targets)
    tot = O.total_loss(ol, O.weight_dict(ocfg))
    ref = dict(zip(names, torch.autograd.grad(tot, [leaves[k] for k in names])))
    assert np.array_equal(out["phrase_mask"].cpu().numpy(), o["phrase_mask"].numpy())
    valid = o["phrase_mask"].view(2, Pn, 2)
    rb = rel(out["pred_logits"].sigmoid()[:, valid.cuda()], o["logits"].sigmoid()[:, valid])
    ld = crit(out, tg)
    rl = max(abs(float(ld[k]) - float(ol[k])) / max(1.0, abs(float(ol[k]))) for k in ol)
    total = sum(ld[k] * crit.weight_dict[k] for k in ld if k in crit.weight_dict)
    model.store.flat_g.zero_()
    total.backward()
    a = torch.cat([model.store.G[k].detach().double().cpu().reshape(-1) for k in names])
    b = torch.cat([ref[k].double().reshape(-1) for k in names])
    rg, cos = float((a - b).norm() / b.norm()), float((a * b).sum() / (a.norm() * b.norm()))
    gq = rel(model.store.G["query_encoder.query_embed.weight"], ref["query_encoder.query_embed.weight"])
    print(f"\n[n_q = 2, {Pn} phrase(s)] boxes {rb:.2e}  worst loss {rl:.2e}  global grad rel {rg:.2e} cos {cos:.4f}  d query_embed rel {gq:.2e}")
    assert rb < TOL["boxes"] * 1.5 and rl < TOL["loss"] * 1.5
    assert rg < 0.25 and cos > 0.97 and gq < 0.1


def test_captured_step_static_batch_filled_in_place(hip):
    """CapturedTrainStep.batch: a batch written INTO the static input buffers and replayed gives the step of the same batch
    handed over as separate tensors (which are copied in); the copy path itself is skipped by identity, not by value."""
    from reftr_amd.engine_vg import CapturedTrainStep, _copy_batch
    from reftr_amd.optim import FusedAdamW
    sa, ta = to_cuda(*make_inputs("e2e_single", B=2, H=96, W=128, L=12))
    sb, tb = to_cuda(*make_inputs("other_batch", B=2, H=96, W=128, L=12))
    model, crit, P, ocfg = build(small=True)
    model.eval()
    opt = FusedAdamW(model, lr=1e-4, lr_backbone=1e-5, weight_decay=1e-4)
    cap = CapturedTrainStep(model, crit, opt, 0.1, sa, ta, warmup=1)
    snap = (model.store.flat_p.clone(), opt.m.clone(), opt.v.clone())

    def restore():
        cap.reset_pending()
        model.store.flat_p.copy_(snap[0]); opt.m.copy_(snap[1]); opt.v.copy_(snap[2]); opt.step_dev.zero_(); opt.step_count = 0
        model.mark_dirty(full=True)
    restore()
    l_copy = [float(cap(sb, tb)[0]) for _ in range(2)]
    restore()
    s_static, t_static = cap.batch
    assert s_static is cap.s and t_static is cap.t
    _copy_batch(s_static, t_static, sa, ta)                 # stale content first: the replay must read what is written next
    s_static["img"].tensors.copy_(sb["img"].tensors); s_static["img"].mask.copy_(sb["img"].mask)
    for k, v in sb.items():
        if torch.is_tensor(v):
            s_static[k].copy_(v)
    for d, t in zip(t_static, tb):
        for k, v in t.items():
            d[k].copy_(v)
    l_static = [float(cap(s_static, t_static)[0]) for _ in range(2)]
    la = float(cap(sa, ta)[0])
    print("static-batch losses", l_copy, l_static, la)
    assert abs(l_copy[0] - l_static[0]) < 1e-6 * abs(l_copy[0]) and abs(l_copy[1] - l_static[1]) < 2e-3 * abs(l_copy[1])
    assert abs(la - l_copy[0]) > 1e-3 * abs(la)             # the two batches are different problems


def test_captured_graphs_of_different_input_kinds_leave_no_stale_gradients(hip, monkeypatch):
    """Overwrite-mode gradients under graph replay: a single-phrase step (T == 1: the decoder's self-attention q/k
    projections produce NO gradient) replayed right after a multi-phrase step (T > 1: they do) must not see the multi-phrase
    step's q/k gradients -- the clear of 'registered but not written this step' is part of every step's own launches, not of
    host history baked into the capture.  Checked against the same alternation with the full per-step clear
    (REFTR_OVERWRITE=0): gradient norms of every step and the never-written matrix itself."""
    from reftr_amd.engine_vg import captured_train_step
    from reftr_amd.optim import FusedAdamW
    single = to_cuda(*make_inputs("e2e_single", B=2, H=96, W=128, L=12))
    multi = to_cuda(*make_inputs("e2e_multi", B=2, H=96, W=128, L=12, n_phrase=3))
    seq = [multi, single, multi, single, single, multi, single]
    key = "vl_transformer.decoder.layers.0.self_attn.in_proj_weight"
    res = {}
    for ow in ("1", "0"):
        monkeypatch.setenv("REFTR_OVERWRITE", ow)
        model, crit, P, ocfg = build(small=True)
        model.cfg.dropout = 0.0
        opt = FusedAdamW(model, lr=1e-5, lr_backbone=1e-6, weight_decay=1e-4)
        model.train()
        E = model.cfg.hidden
        norms, qk = [], []
        for s, tg in seq:
            _, _, _, gn = captured_train_step(model, crit, s, tg, opt, None, max_norm=0.1)
            norms.append(float(gn))
            qk.append(float(model.store.G[key][: 2 * E].norm()))
        assert len(model._captured_steps) == 2
        res[ow] = (norms, qk)
    (n1, q1), (n0, q0) = res["1"], res["0"]
    print("\n[stale-gradient check] grad norms overwrite", ["%.4f" % v for v in n1], "full clear", ["%.4f" % v for v in n0])
    for i, (s, _) in enumerate(seq):
        if "phrase" not in s:               # single-phrase step: the q/k rows of the in_proj gradient are exactly zero
            assert q1[i] == 0.0 and q0[i] == 0.0, (i, q1[i], q0[i])
        else:
            assert q1[i] > 0.0
        assert abs(n1[i] - n0[i]) < 5e-2 * n0[i], (i, n1[i], n0[i])


def test_lr_backbone_zero_trains_everything_but_the_resnet(hip):
    """--lr_backbone 0 (models/modeling/backbone.py:87-89,150): same forward, the ResNet's backward is not run and its weights do
    not move (no update, no weight decay), every other gradient equals the regular model's (the losses do not depend on who is
    trainable), and the clip norm only counts what is trained."""
    from reftr_amd.engine_vg import train_step
    from reftr_amd.models import layout as L
    from reftr_amd.models.criterion import CriterionVGMultiPhrase
    from reftr_amd.models.reftr_transformer import RefTR
    from reftr_amd.optim import FusedAdamW
    samples, targets = make_inputs("e2e_single", B=2, H=96, W=128, L=12)
    s, tg = to_cuda(samples, targets)
    ocfg = O.Cfg(enc_layers=2, dec_layers=2, bert=O.BertCfg(layers=2))
    P = formula_state(param_shapes(ocfg))
    res = {}
    for train_bb in (True, False):
        cfg = L.ModelConfig(enc_layers=2, dec_layers=2, bert=L.BertConfig(layers=2), train_backbone=train_bb)
        model = RefTR(cfg, device="cuda")
        model.load_state_dict(P, strict=True)
        model.eval()
        crit = CriterionVGMultiPhrase(O.weight_dict(ocfg), ["boxes"])
        opt = FusedAdamW(model, lr=1e-4, lr_backbone=1e-5, weight_decay=1e-4)
        lv, _, _, gn = train_step(model, crit, s, tg, opt, None, max_norm=0.1)
        torch.cuda.synchronize()
        sd = model.state_dict()
        grads = {n: model.store.G[n].detach().float().cpu().clone() for n in model.store.G}
        res[train_bb] = (lv, float(gn), sd, grads)
    (l1, g1, sd1, gr1), (l0, g0, sd0, gr0) = res[True], res[False]
    assert abs(l1 - l0) < 1e-6 * abs(l1)                                        # same forward
    bb = [k for k in P if k.startswith("img_backbone.") and "weight" in k and "layer" in k and "bn" not in k and "downsample.1" not in k]
    assert all(torch.equal(sd0[k].cpu(), P[k]) for k in bb)                   # frozen: bit-identical to the loaded weights
    assert any(not torch.equal(sd1[k].cpu(), P[k]) for k in bb)               # regular model: they moved
    assert not any(n.startswith("img_backbone.") for n in gr0)                # no gradient storage for the ResNet at all
    common = [n for n in gr0 if gr0[n].norm() > 0]
    a = torch.cat([gr0[n].reshape(-1) for n in common]); b = torch.cat([gr1[n].reshape(-1) for n in common])
    assert rel(a, b) < 1e-3                                                   # atomics-order noise only
    assert g0 < g1                                                            # the clip norm lost the ResNet's share
    assert rel(sd0["bbox_embed.layers.1.weight"], sd1["bbox_embed.layers.1.weight"]) < 1e-5
