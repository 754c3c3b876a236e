"""GPU: the few-row region between the encoder's forward and backward as three launches (round 5: rt_qenc_fwd, rt_head_loss,
rt_qenc_bwd; csrc/rt_qregion.hip) against the launched chains they replace (rt_conv_gemm + rt_layernorm_* + rt_rows_add +
rt_qenc_attn_* + rt_box_loss + rt_small_dgrad).  Same operands, same rounding points, another accumulation order inside the
products: every tensor the backward / the weight-gradient launches read must agree to fp32-order noise, bf16 tensors to one
rounding flip in a few thousand elements."""
import numpy as np
import pytest
import torch

from test_model_gpu import build, make_inputs, rel, to_cuda

pytestmark = pytest.mark.gpu


def _flips(a, b):
    """fraction of bf16 elements that differ, and the relative L2 distance of the two tensors (a flipped rounding moves an element
    by 2^-8 ... 2^-7 of its value: a fraction f of flips gives ~ sqrt(f) * 5e-3)"""
    a, b = a.float(), b.float()
    return float(((a - b).abs() > 0).float().mean()), rel(a, b)


def _qenc_inputs(model, B, Lq, HW, Pn, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    E = model.cfg.hidden
    S = Lq + HW
    mem32 = torch.randn(B * S, E, device="cuda", generator=g)
    mem16 = mem32.to(torch.bfloat16)
    ctx = (torch.rand(B, Pn, Lq, device="cuda", generator=g) < 0.3).to(torch.uint8)
    ctx[:, :, 0] = 0                                     # reftr.py:93: the CLS token is never masked
    cat16 = (0.5 * torch.randn(B * Pn, 2 * E, device="cuda", generator=g)).to(torch.bfloat16)
    qmask = torch.zeros(B, Pn, dtype=torch.uint8, device="cuda")
    return mem32, mem16, ctx, cat16, qmask, S


@pytest.mark.parametrize("B,Lq,Pn,train", [(8, 40, 1, True), (2, 12, 1, False), (2, 90, 16, True), (3, 96, 5, True), (1, 7, 3, False)])
def test_qenc_forward_one_launch_equals_the_launched_chain(hip, B, Lq, Pn, train):
    model, crit, P, ocfg = build(small=True)
    model.train(train)
    net = model.net
    model.refresh_now()
    mem32, mem16, ctx, cat16, qmask, S = _qenc_inputs(model, B, Lq, 20, Pn)
    hip.set_seed_dev(model.seed_dev)
    outs = {}
    for mode in ("chain", "fused"):
        net.begin_step(train)
        cat = cat16.clone()
        if mode == "chain":
            r = model._qenc_fwd_chain(mem16, mem32, ctx, cat, cat.view(2 * B * Pn, -1), qmask, B, S, Lq, Pn)
            names = ("cls16", "lang16", "kq", "qs", "vs", "qw", "c16", "co", "cmean", "crstd", "fq_ctx", "tgt32", "tgt16", "qpos", "tgtq16")
            o = dict(zip(names, r[:15]))
        else:
            assert model._qfuse_ok(Lq, Pn)
            o = model._qenc_fwd_fused(mem16, mem32, ctx, cat, B, S, Lq, Pn)
        fq = o.pop("fq_ctx")
        o.update(cat16=cat, t1=fq["t1"], m1=fq["st1"][0], r1=fq["st1"][1], a16=fq["a16"], t2=fq["t2"], m2=fq["st2"][0], r2=fq["st2"][1])
        assert fq["st1"][2] == (0.1 if train else 0.0)
        torch.cuda.synchronize()
        outs[mode] = {k: v.clone() for k, v in o.items()}
    hip.set_seed_dev(None)
    a, f = outs["chain"], outs["fused"]
    worst = {}
    for k in a:
        assert a[k].shape == f[k].shape and a[k].dtype == f[k].dtype, k
        if a[k].dtype == torch.bfloat16:
            frac, mx = _flips(f[k], a[k])
            worst[k] = frac
            assert frac < 2e-2 and mx < 1e-3, (k, frac, mx)          # few rounding flips
        else:
            worst[k] = rel(f[k], a[k])
            assert worst[k] < 6e-4, (k, worst[k])            # fp32-order noise (1e-7) or a bf16 flip of an input propagated (1e-5 ... 1e-4)
    assert torch.equal(a["cls16"], f["cls16"]) and torch.equal(a["lang16"], f["lang16"])         # copies
    print(f"\n[qenc fwd B={B} L={Lq} P={Pn} train={train}] " + "  ".join(f"{k}={v:.1e}" for k, v in worst.items()))


@pytest.mark.parametrize("B,Lq,Pn,train,with_gb", [(8, 40, 1, True, False), (2, 12, 1, False, False), (2, 90, 16, True, True), (3, 96, 5, True, True)])
def test_qenc_backward_one_launch_equals_the_launched_chain(hip, B, Lq, Pn, train, with_gb):
    """Same saved forward, same upstream gradients: d memory, d cat (map_phrase's share), every weight / bias / LayerNorm / query_embed
    gradient of the QueryEncoder."""
    model, crit, P, ocfg = build(small=True)
    model.train(train)
    net, st = model.net, model.store
    model.refresh_now()
    E = model.cfg.hidden
    mem32, mem16, ctx, cat16, qmask, S = _qenc_inputs(model, B, Lq, 20, Pn, seed=1)
    hip.set_seed_dev(model.seed_dev)
    net.begin_step(train)
    o = model._qenc_fwd_fused(mem16, mem32, ctx, cat16, B, S, Lq, Pn)
    sv = dict(Nf=B * Pn, fq_ctx=o["fq_ctx"], co=o["co"], cst=(o["cmean"], o["crstd"]), c16=o["c16"], cls16=o["cls16"], lang16=o["lang16"],
              kq=o["kq"], qs=o["qs"], vs=o["vs"], qw=o["qw"])
    g = torch.Generator(device="cuda").manual_seed(5)
    N = B * Pn
    ga = torch.randn(N, E, device="cuda", generator=g) * 1e-2
    gb = torch.randn(N, E, device="cuda", generator=g) * 1e-2 if with_gb else None
    dqpos = torch.randn(N, E, device="cuda", generator=g) * 1e-2
    dmem0 = torch.randn(B * S, E, device="cuda", generator=g) * 1e-2
    res = {}
    for mode in ("chain", "fused"):
        st.flat_g.zero_()
        dmem = dmem0.clone()
        if mode == "chain":
            dcat, _ = model._qenc_bwd_chain(sv, ga, gb, dqpos, dmem, B, S, Lq, Pn, N)
        else:
            dcat = model._qenc_bwd_fused(sv, ga, gb, dqpos, dmem, B, S, Lq, Pn)
        net.flush_wgrads()
        torch.cuda.synchronize()
        grads = {n: st.G[n].clone() for n in st.G if n.startswith("query_encoder.")}
        res[mode] = (dcat.clone(), dmem.clone(), grads)
    hip.set_seed_dev(None)
    (dc_a, dm_a, g_a), (dc_f, dm_f, g_f) = res["chain"], res["fused"]
    # the bf16 dy operands inside differ by rounding flips (the products' accumulation order), so fp32 results agree to ~1e-3 of a
    # bf16 step relative to their own size
    r_dcat, r_dmem = rel(dc_f, dc_a), rel(dm_f - dmem0, dm_a - dmem0)
    # linear2.bias is exempt from the relative comparison: softmax is shift-invariant, so its gradient k * sum_l ds[l] is exactly zero
    # in exact arithmetic and pure rounding noise on both sides (checked to be noise-sized below)
    zero_in_theory = "query_encoder.linear2.bias"
    worst = max((rel(g_f[n], g_a[n]), n) for n in g_a if float(g_a[n].abs().sum()) > 0 and n != zero_in_theory)
    for tag, gr in (("chain", g_a), ("fused", g_f)):
        bad = [n for n in gr if not bool(torch.isfinite(gr[n]).all())]
        assert not bad, (tag, bad)
        assert float(gr[zero_in_theory].norm()) <= 5e-2 * float(gr["query_encoder.linear3.bias"].norm()) + 1e-12, tag
    print(f"\n[qenc bwd B={B} L={Lq} P={Pn} train={train} gb={with_gb}] dcat {r_dcat:.1e}  dmem {r_dmem:.1e}  worst grad {worst[0]:.1e} ({worst[1]})")
    assert r_dcat < 2e-3 and r_dmem < 2e-3 and worst[0] < 3e-3, (r_dcat, r_dmem, worst)
    assert set(n for n in g_a if float(g_a[n].abs().sum()) > 0) == set(n for n in g_f if float(g_f[n].abs().sum()) > 0)


@pytest.mark.parametrize("kind,n_phrase", [("single", 0), ("multi", 3)])
def test_eager_step_with_the_fused_query_encoder_matches_the_chain(hip, monkeypatch, kind, n_phrase):
    """Whole model, eager loop body, dropout on: REFTR_QFUSE=1 vs 0 -- the same dropout sites are drawn (net._drop), losses to
    fp32-order noise, gradients to the noise of rounding flips."""
    from reftr_amd.engine_vg import train_step
    from reftr_amd.optim import FusedAdamW
    samples, targets = make_inputs("e2e_" + kind, B=2, H=96, W=128, L=12, n_phrase=n_phrase)
    s, tg = to_cuda(samples, targets)
    out = {}
    for q in ("0", "1"):
        monkeypatch.setenv("REFTR_QFUSE", q)
        model, crit, P, ocfg = build(small=True)
        model.train()
        opt = FusedAdamW(model, lr=1e-4, lr_backbone=1e-5, weight_decay=1e-4)
        loss, ld, _, gn = train_step(model, crit, s, tg, opt, None, max_norm=0.1)
        torch.cuda.synchronize()
        out[q] = (loss, float(gn), model.store.flat_g.clone())
    (l0, n0, g0), (l1, n1, g1) = out["0"], out["1"]
    print(f"\n[eager step {kind}] loss {l0:.6f} / {l1:.6f}  grad norm {n0:.5f} / {n1:.5f}  grad rel {rel(g1, g0):.1e}")
    assert abs(l1 - l0) < 2e-5 * abs(l0) and abs(n1 - n0) < 5e-3 * n0 and rel(g1, g0) < 2e-2


def test_captured_step_with_the_fused_head_matches_the_launched_head(hip, monkeypatch):
    """CapturedTrainStep's direct loss path with rt_head_loss (decoder.norm + box head + losses + their backward-data in one launch)
    against the same path with the launched head: every loss term, the gradient norm, the whole gradient buffer, the weights after
    the update."""
    from reftr_amd.engine_vg import CapturedTrainStep
    from reftr_amd.optim import FusedAdamW
    samples, targets = make_inputs("e2e_single", B=2, H=96, W=128, L=12)
    s, tg = to_cuda(samples, targets)
    out = {}
    for hf in ("0", "1"):
        monkeypatch.setenv("REFTR_HEAD_FUSE", hf)
        model, crit, P, ocfg = build(small=True)
        model.eval()
        opt = FusedAdamW(model, lr=1e-4, lr_backbone=1e-5, weight_decay=1e-4)
        p0, m0, v0 = model.store.flat_p.clone(), opt.m.clone(), opt.v.clone()
        cap = CapturedTrainStep(model, crit, opt, 0.1, s, tg, warmup=1)
        assert cap._direct_loss_ok()
        cap.reset_pending()
        model.store.flat_p.copy_(p0); opt.m.copy_(m0); opt.v.copy_(v0); opt.step_dev.zero_(); opt.step_count = 0
        model.mark_dirty(full=True)
        l, ld, gn = cap(s, tg)
        cap.flush()
        torch.cuda.synchronize()
        assert (model._saved.get("head") is not None) == (hf == "1")
        out[hf] = (float(l), {k: float(v) for k, v in ld.items()}, float(gn), model.store.flat_g.clone(), model.store.flat_p.clone())
    a, b = out["0"], out["1"]
    assert sorted(a[1]) == sorted(b[1])
    worst = max(abs(a[1][k] - b[1][k]) / max(abs(a[1][k]), 1e-9) for k in a[1])
    print(f"\n[captured step, fused head] total {a[0]:.6f} / {b[0]:.6f}  worst loss term {worst:.1e}  grad norm {a[2]:.5f} / {b[2]:.5f}  "
          f"grad rel {rel(b[3], a[3]):.1e}  weights rel {rel(b[4], a[4]):.1e}")
    assert abs(a[0] - b[0]) < 2e-6 * abs(a[0]) and worst < 5e-6
    assert abs(a[2] - b[2]) < 5e-4 * a[2] and rel(b[3], a[3]) < 5e-3 and rel(b[4], a[4]) < 1e-5     # rounding / ReLU-gate flips of y1, y2 (see the head test)


def test_head_loss_rows_follow_rt_box_loss(hip):
    """rt_head_loss's loss rows against rt_box_loss on the SAME logits (it writes them): losses to the order of its in-workgroup
    sum, d logits bit-identical (both evaluate rt_box_loss_row), invalid phrases exactly zero; multi-phrase with ragged validity."""
    from reftr_amd.models.criterion import _box_weights
    model, crit, P, ocfg = build(small=True)
    model.eval()
    model.refresh_now()
    B, Pn, nq, NL, E = 3, 4, 1, model.cfg.dec_layers, model.cfg.hidden
    N = B * Pn * nq
    g = torch.Generator(device="cuda").manual_seed(3)
    t3 = torch.randn(NL * N, E, device="cuda", generator=g)
    hs16 = torch.empty(NL * N, E, dtype=torch.bfloat16, device="cuda")
    valid = torch.tensor([[1, 1, 0, 0], [1, 0, 1, 1], [0, 0, 0, 0]], dtype=torch.bool, device="cuda")
    nt = [2, 3, 0]
    tg = [{"boxes": torch.rand(n, 4, device="cuda", generator=g) * 0.4 + 0.3, "labels": torch.zeros(n, dtype=torch.long, device="cuda")} for n in nt]
    prepared = crit.prepare(tg, torch.device("cuda"))
    model.store.flat_g.zero_()
    h = model._head_loss_fused((crit, prepared), t3, hs16, valid, NL, B, Pn, nq, N)
    logits = h["logits"].view(NL, B, Pn, nq, 4)
    losses, _, dl = hip.box_loss(logits.contiguous(), valid.to(torch.uint8).contiguous(), *prepared, want_grad=True,
                                 weights=_box_weights(crit, NL, logits.device))
    torch.cuda.synchronize()
    assert rel(h["dlogits"], dl.view(-1, 4)) < 1e-6          # the same row function, inlined into two kernels (contraction may differ)
    assert rel(h["losses"], losses) < 1e-6
    inval = (~valid).view(1, B, Pn, 1, 1).expand(NL, B, Pn, nq, 4)
    assert float(h["dlogits"].view(NL, B, Pn, nq, 4)[inval].abs().max()) == 0.0
    assert rel(h["db2_part"][:, 0].sum(0), dl.view(-1, 4).sum(0)) < 1e-5 and float(h["db2_part"][:, 1].abs().max()) == 0.0


@pytest.mark.parametrize("B,Pn", [(8, 1), (2, 16), (5, 3)])
def test_head_loss_one_launch_equals_the_launched_head(hip, B, Pn):
    """decoder.norm -> bbox MLP -> box loss -> the MLP's and the norm's backward-data: every intermediate of rt_head_loss against the
    launches RefTR._forward_impl / _backward_phases issue for the same rows."""
    from reftr_amd.models.criterion import _box_weights
    from reftr_amd.models.net import RELU
    model, crit, P, ocfg = build(small=True)
    model.eval()
    model.refresh_now()
    net, st = model.net, model.store
    nq, NL, E = 1, model.cfg.dec_layers, model.cfg.hidden
    N = B * Pn * nq
    g = torch.Generator(device="cuda").manual_seed(11)
    t3 = torch.randn(NL * N, E, device="cuda", generator=g)
    valid = torch.rand(B, Pn, device="cuda", generator=g) < 0.7
    valid[:, 0] = True
    nt = [int(v) for v in valid.sum(1).tolist()]
    tg = [{"boxes": torch.rand(n, 4, device="cuda", generator=g) * 0.4 + 0.3, "labels": torch.zeros(n, dtype=torch.long, device="cuda")} for n in nt]
    prepared = crit.prepare(tg, torch.device("cuda"))
    vt = "vl_transformer."
    # ---- launched
    st.flat_g.zero_()
    hs16 = torch.empty(NL * N, E, dtype=torch.bfloat16, device="cuda")
    _, _, _, hm, hr = net.ln_fwd(t3, vt + "decoder.norm.", y_bf16=hs16, want_f32=False)
    y1, _ = net.lin_fwd("bbox_embed.layers.0.", hs16, act=RELU)
    y2, _ = net.lin_fwd("bbox_embed.layers.1.", y1, act=RELU)
    _, logits = net.lin_fwd("bbox_embed.layers.2.", y2, out_bf16=False, out_f32=True)
    w = _box_weights(crit, NL, logits.device)
    losses, _, dl = hip.box_loss(logits.view(NL, B, Pn, nq, 4), valid.to(torch.uint8).contiguous(), *prepared, want_grad=True, weights=w)
    dl = dl.reshape(NL * N, 4)
    l2 = net.lins["bbox_embed.layers.2."]
    hip.colsum(dl, l2.gb)
    dy2 = hip.small_dgrad(dl, l2.w32, gate=y2)
    dy1, _ = net.lin_bwd("bbox_embed.layers.1.", dy2, y1, gate=y1)
    _, dhs = net.lin_bwd("bbox_embed.layers.0.", dy1, hs16, out_bf16=False, out_f32=True)
    dnorm, _ = net.ln_bwd(dhs, t3, vt + "decoder.norm.", hm, hr, want_bf16=False)
    net.flush_wgrads()
    torch.cuda.synchronize()
    ref = dict(hs16=hs16, y1=y1, y2=y2, hmean=hm, hrstd=hr, logits=logits, losses=losses, dlogits=dl, dy2=dy2, dy1=dy1, dhs=dhs, dnorm=dnorm)
    gref = {n: st.G[n].clone() for n in st.G if n.startswith("bbox_embed.") or "decoder.norm" in n}
    # ---- one launch
    st.flat_g.zero_()
    hs16f = torch.empty_like(hs16)
    h = model._head_loss_fused((crit, prepared), t3, hs16f, valid, NL, B, Pn, nq, N)
    net._wgrad_only("bbox_embed.layers.1.", h["dy2"], h["y1"])
    net._wgrad_only("bbox_embed.layers.0.", h["dy1"], hs16f)
    net.ln_batch.jobs.append(hip.LnPgJob(hip._p(h["part_n"]), hip._p(st.G[vt + "decoder.norm.weight"]), hip._p(st.G[vt + "decoder.norm.bias"]), NL, E))
    net.ln_batch.jobs.append(hip.LnPgJob(hip._p(h["db2_part"]), hip._p(net.lins["bbox_embed.layers.2."].gb), None, NL, 4))
    net.flush_wgrads()
    torch.cuda.synchronize()
    gf = {n: st.G[n].clone() for n in gref}
    got, mxs = {}, {}
    for k, r in ref.items():
        f = h[k]
        if r.dtype == torch.bfloat16:
            got[k], mxs[k] = _flips(f, r)
        else:
            got[k] = rel(f.reshape(r.shape), r)
    print(f"\n[head B={B} P={Pn}] " + "  ".join(f"{k}={v:.1e}" for k, v in got.items()))
    for k, r in ref.items():
        if r.dtype == torch.bfloat16:
            # forward tensors: rounding flips only (one bf16 step), few of them; the gradients behind a ReLU gate also carry the
            # gate decisions of flipped activations (a whole element appears / disappears)
            assert got[k] < 2e-2 and mxs[k] < (1e-2 if k in ("dy2", "dy1") else 1e-3), (k, got[k], mxs[k])
        else:
            tol = {"hmean": 3e-6, "hrstd": 3e-6, "losses": 2e-5, "dlogits": 3e-5, "logits": 3e-4, "dhs": 5e-3, "dnorm": 5e-3}[k]
            assert got[k] < tol, (k, got[k])
    for n in gref:
        if n == "bbox_embed.layers.2.weight":
            continue                      # its launch (rt_conv_wgrad on dl16) is issued by _backward_phases on both paths
        assert rel(gf[n], gref[n]) < 5e-3, (n, rel(gf[n], gref[n]))


def test_cfg2_step_with_the_three_launches_matches_the_launched_region(hip, monkeypatch):
    """configs[1] itself (640 x 640, B = 8, L = 40, 12 + 6 + 6 layers, dropout on, captured step): the region as three launches against
    the region as ~75 launches -- every loss term, the gradient norm, the whole gradient buffer.  At this size the launched head's
    products go through the MFMA kernel (48 rows), whose K order the fused kernels share, so the two agree far below the noise of
    the backward's atomics."""
    from oracle.synth import make_inputs as mk
    from reftr_amd.engine_vg import CapturedTrainStep
    from reftr_amd.optim import FusedAdamW
    from test_parity_fullsize_gpu import build_full
    samples, targets = mk("e2e_single", B=8, H=640, W=640, L=40)
    s, tg = to_cuda(samples, targets)
    out = {}
    for fuse in ("0", "1"):
        monkeypatch.setenv("REFTR_HEAD_FUSE", fuse)
        monkeypatch.setenv("REFTR_QFUSE", fuse)
        model, crit, P, ocfg = build_full()
        model.train()
        opt = FusedAdamW(model, lr=1e-4, lr_backbone=1e-5, weight_decay=1e-4)
        p0, m0, v0 = model.store.flat_p.clone(), opt.m.clone(), opt.v.clone()
        cap = CapturedTrainStep(model, crit, opt, 0.1, s, tg, warmup=1)
        cap.reset_pending()
        model.store.flat_p.copy_(p0); opt.m.copy_(m0); opt.v.copy_(v0); opt.step_dev.zero_(); opt.step_count = 0
        model.mark_dirty(full=True)
        seed0 = (model.seed_dev.clone(), model._step)
        l, ld, gn = cap(s, tg)
        torch.cuda.synchronize()
        assert (model._saved.get("head") is not None) == (fuse == "1")
        out[fuse] = (float(l), {k: float(v) for k, v in ld.items()}, float(gn), model.store.flat_g.clone(), seed0)
        del cap, model, opt
        torch.cuda.empty_cache()
    a, b = out["0"], out["1"]
    assert int(a[4][0]) == int(b[4][0]) and a[4][1] == b[4][1]           # the same dropout seeds on both sides
    worst = max(abs(a[1][k] - b[1][k]) / max(abs(a[1][k]), 1e-9) for k in a[1])
    print(f"\n[cfg2 step, region fused / launched] total {b[0]:.6f} / {a[0]:.6f}  worst loss term {worst:.1e}  grad norm {b[2]:.5f} / {a[2]:.5f}  "
          f"grad rel {rel(b[3], a[3]):.1e}")
    assert worst < 1e-5 and abs(a[2] - b[2]) < 2e-3 * a[2] and rel(b[3], a[3]) < 5e-3
