"""GPU: (1) the cooperative decoder launches fail SAFE -- an iteration whose stage hand-off timed out never updates a parameter
(decided on the device, inside the replayed graph) and the loop goes on, on the launched chain, with the trajectory a chain-only run
has (the reference stops before the update when an iteration is bad, engine_vg.py:55-58); (2) the matrix-aware AdamW pass
(rt_adamw_mat + rt_adamw_chunks), which also writes the bf16 GEMM operands, against the flat pass + rt_weight_prep_batched it
replaces (main_vg.py:234-268's AdamW, engine_vg.py:62-66's clip)."""
import os

import pytest
import torch

from oracle import reftr_oracle as O
from oracle.shapes import param_shapes
from oracle.synth import make_inputs
from oracle.weights import formula_state

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import __graft_entry__ as g
    g.build()
    from reftr_amd import hip as H
    return H


def to_cuda(samples, targets):
    from reftr_amd.util.misc import NestedTensor
    s = {k: v.cuda() for k, v in samples.items() if k not in ("img", "img_mask")}
    s["img"] = NestedTensor(samples["img"].cuda(), samples["img_mask"].cuda())
    return s, [{k: v.cuda() for k, v in t.items()} for t in targets]


def build(dec_layers=3, B=4, H=96, W=128, lr=1e-4):
    from reftr_amd.models import layout as L
    from reftr_amd.models.criterion import CriterionVGMultiPhrase
    from reftr_amd.models.reftr_transformer import RefTR
    from reftr_amd.optim import FusedAdamW
    ocfg = O.Cfg(enc_layers=2, dec_layers=dec_layers, bert=O.BertCfg(layers=2))
    cfg = L.ModelConfig(enc_layers=2, dec_layers=dec_layers, bert=L.BertConfig(layers=2))
    model = RefTR(cfg, device="cuda")
    model.load_state_dict(formula_state(param_shapes(ocfg)), strict=True)
    torch.manual_seed(3)
    model.store.P["bbox_embed.layers.2.weight"].normal_(0, 0.02)       # a zero head hides the decoder from the loss
    model.mark_dirty()
    model.train()
    crit = CriterionVGMultiPhrase(O.weight_dict(ocfg), ["boxes"])
    samples, targets = make_inputs("coop", B=B, H=H, W=W, L=12)
    s, tg = to_cuda(samples, targets)
    opt = FusedAdamW(model, lr=lr, lr_backbone=lr / 10)
    return model, crit, s, tg, opt


@pytest.fixture
def coop_env(hip):
    """Restores what a forced failure changes for the whole process: the poll budget and the REFTR_DEC_COOP* switches."""
    import reftr_amd.engine_vg as E
    saved = {k: os.environ.get(k) for k in ("REFTR_DEC_COOP", "REFTR_DEC_COOP_BWD")}
    logged = E._COOP_LOGGED
    try:
        yield hip
    finally:
        hip.decoder_set_spin(0)
        E._COOP_LOGGED = logged
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_residency_query_admits_the_launches_on_this_device(hip):
    assert hip.decoder_supported(2048)
    assert not hip.decoder_supported(1024)           # a width the kernels are not built for


def test_a_timed_out_iteration_never_updates_a_parameter(coop_env):
    """Device-side veto: a graph captured with a poll budget of 1 fails in every replay; neither the weights, nor the moments, nor
    the optimizer's step counter move, however often it is replayed (nobody on the host looks at the failure word here)."""
    from reftr_amd.engine_vg import CapturedTrainStep
    hip = coop_env
    model, crit, s, tg, opt = build()
    assert model.net.dec_stack_coop_ok(4, 1, 12 + 12, 3, True)
    hip.decoder_set_spin(1)
    cap = CapturedTrainStep(model, crit, opt, 0.1, s, tg)
    assert cap.deferred and cap.fail_word is not None
    torch.cuda.synchronize()
    assert int(cap.fail_word) != 0, "the poll budget of 1 did not produce a timeout"
    p0, m0, v0, sd0 = model.store.flat_p.clone(), opt.m.clone(), opt.v.clone(), int(opt.step_dev)
    for _ in range(4):
        cap(*cap.batch)
    torch.cuda.synchronize()
    assert torch.equal(model.store.flat_p, p0) and torch.equal(opt.m, m0) and torch.equal(opt.v, v0)
    assert int(opt.step_dev) == sd0 and int(opt.active) == 0
    assert float(cap.stats[len(cap.stat_names)]) != 0          # the word travels in the stats vector, behind the losses


def test_forced_timeout_falls_back_to_the_chain_and_matches_the_chain_only_trajectory(coop_env):
    """The loop's path: the first replayed iteration times out -> `finish()` sees the word before the next replay would apply
    anything, drops the captures, switches the launches off for the process and runs the SAME batch again on the chain.  The
    losses and the parameters after three iterations are those of a run that never used the cooperative launches."""
    from reftr_amd.engine_vg import captured_train_step
    hip = coop_env

    def trajectory(force_fail):
        model, crit, s, tg, opt = build()
        if force_fail:
            assert model.net.dec_coop and model.net.dec_stack_coop_ok(4, 1, 24, 3, True)
            hip.decoder_set_spin(1)
        else:
            model.net.dec_coop = model.net.dec_coop_bwd = False
        p0 = model.store.flat_p.clone()
        out = []
        for it in range(3):
            loss, _, _, gn = captured_train_step(model, crit, s, tg, opt, None, 0.1)
            out.append((loss, float(gn)))
            if it == 0 and force_fail:
                assert not model.net.dec_coop and os.environ.get("REFTR_DEC_COOP") == "0"     # fell back, said so once
                assert int(model.net._dec_handoff[1]) == 0                                      # word cleared for what follows
        for c in model.__dict__["_captured_steps"].values():
            c.flush()
        torch.cuda.synchronize()
        assert opt.step_count == 3 and int(opt.step_dev) == 3
        assert not torch.equal(model.store.flat_p, p0)
        return out, model.store.flat_p.clone()

    ref, p_ref = trajectory(False)
    hip.decoder_set_spin(0)
    os.environ.pop("REFTR_DEC_COOP", None); os.environ.pop("REFTR_DEC_COOP_BWD", None)
    got, p_got = trajectory(True)
    assert abs(got[0][0] - ref[0][0]) <= 1e-6 * abs(ref[0][0]), (got, ref)        # iteration 1: same weights, same seeds, same kernels
    for (lg, gg), (lr_, gr) in zip(got, ref):
        assert abs(lg - lr_) <= 2e-3 * abs(lr_) and abs(gg - gr) <= 2e-2 * gr, (got, ref)   # later: atomics' summation order (ulp) flips bf16 roundings
    d = float((p_got - p_ref).norm() / p_ref.norm())
    assert d < 5e-5, d            # measured 1.3e-5: two runs of the SAME chain differ like this (atomics' order -> flipped bf16 roundings)


def test_eager_loop_body_reruns_a_timed_out_iteration_on_the_chain(coop_env):
    from reftr_amd.engine_vg import train_step
    hip = coop_env
    model, crit, s, tg, opt = build()
    ref_model, _, _, _, ref_opt = build()
    ref_model.net.dec_coop = ref_model.net.dec_coop_bwd = False
    l_ref = train_step(ref_model, crit, s, tg, ref_opt, None, 0.1)[0]
    hip.decoder_set_spin(1)
    l_got = train_step(model, crit, s, tg, opt, None, 0.1)[0]
    torch.cuda.synchronize()
    assert not model.net.dec_coop
    assert abs(l_got - l_ref) <= 1e-6 * abs(l_ref)
    d = float((model.store.flat_p - ref_model.store.flat_p).norm() / ref_model.store.flat_p.norm())
    assert d < 1e-6, d


# ----------------------------------------------------------------------------------------------------------------------------------
def test_matrix_adamw_equals_flat_adamw_plus_weight_prep(hip):
    """rt_adamw_mat + rt_adamw_chunks over a synthetic flat buffer (a vectorisable Linear, a 3x3 convolution with a FrozenBN scale,
    a ragged matrix that takes the element-wise path, loose vectors in between) == rt_adamw_flat followed by rt_weight_prep_batched:
    masters and moments to 1 ulp (the two kernels are compiled separately), operands bit-equal given equal masters."""
    from reftr_amd.optim import cover_span
    H = hip
    dev = "cuda"
    g = torch.Generator(device="cpu").manual_seed(5)
    mats = [(256, 192, 1, 128), (256 + 192 * 128 + 40, 72, 9, 64), (256 + 192 * 128 + 40 + 72 * 9 * 64 + 8, 10, 1, 12)]
    n = mats[-1][0] + 10 * 12 + 4096 + 36
    n = (n + 3) // 4 * 4
    p = torch.randn(n, generator=g).to(dev); gr = (torch.randn(n, generator=g) * 0.1).to(dev)
    m = (torch.randn(n, generator=g) * 0.01).to(dev); v = (torch.rand(n, generator=g) * 1e-3).to(dev)
    sq = (gr.double() ** 2).sum().float().reshape(1).to(dev)
    ranges = [(0, 24840, 1e-3, 1e-4), (24840, n, 1e-4, 0.0)]          # the boundary lies between the first two matrices
    kw = dict(step=3, ranges=ranges, gnorm_sq=sq, grad_scale=0.5, max_norm=0.1)
    # reference: flat pass + prep
    p1, m1, v1 = p.clone(), m.clone(), v.clone()
    gn1 = torch.zeros(1, device=dev)
    H.adamw_flat(p1, gr, m1, v1, gnorm_out=gn1, **kw)
    scale = (torch.rand(72, generator=g) + 0.5).to(dev)
    outs1 = []
    prep = H.WeightPrepBatch(torch.device(dev))
    for i, (off, N, T, C) in enumerate(mats):
        W = torch.zeros(N, T, C, dtype=torch.bfloat16, device=dev); WT = torch.zeros(C, T, N, dtype=torch.bfloat16, device=dev)
        prep.add(p1[off:off + N * T * C], N, T, C, scale=scale if i == 1 else None, dst=W, dst_t=WT)
        outs1.append((W, WT))
    prep.run()
    # matrix-aware pass
    p2, m2, v2 = p.clone(), m.clone(), v.clone()
    gn2 = torch.zeros(1, device=dev)
    jobs, tiles, chunks = cover_span(mats, 0, n)
    outs2, rows = [], []
    for i, (off, N, T, C, first) in enumerate(jobs):
        W = torch.zeros(N, T, C, dtype=torch.bfloat16, device=dev); WT = torch.zeros(C, T, N, dtype=torch.bfloat16, device=dev)
        rows.append([off, scale.data_ptr() if i == 1 else 0, W.data_ptr(), WT.data_ptr(), N, T, C, first])
        outs2.append((W, WT))
    tab = torch.tensor(rows, dtype=torch.int64).to(dev)
    ctab = torch.tensor(chunks, dtype=torch.int64).to(dev)
    H.adamw_flat(p2, gr, m2, v2, gnorm_out=gn2, mat=(tab, len(rows), tiles), chunks=(ctab, len(chunks) // 2), **kw)
    torch.cuda.synchronize()
    assert torch.equal(gn1, gn2)
    for a, b in ((p1, p2), (m1, m2), (v1, v2)):
        assert float((a - b).abs().max()) <= 2e-7 * float(a.abs().max())
    assert not torch.equal(p2, p)
    for i, ((W1, WT1), (W2, WT2)) in enumerate(zip(outs1, outs2)):
        off, N, T, C = mats[i]
        ref = p2[off:off + N * T * C].view(N, T, C)
        if i == 1:
            ref = ref * scale.view(N, 1, 1)
        assert torch.equal(W2, ref.to(torch.bfloat16)) and torch.equal(WT2, ref.to(torch.bfloat16).permute(2, 1, 0).contiguous())
        assert float((W1.float() - W2.float()).abs().max()) <= 2 ** -7 * float(W1.float().abs().max())     # at most a flipped rounding
    # bf16 gradients (the data-parallel exchange buffer) and the `active` word
    g16 = gr.to(torch.bfloat16)
    p3, m3, v3 = p.clone(), m.clone(), v.clone(); p4, m4, v4 = p.clone(), m.clone(), v.clone()
    H.adamw_flat(p3, gr, m3, v3, g16=g16, **kw)
    H.adamw_flat(p4, gr, m4, v4, g16=g16, mat=(tab, len(rows), tiles), chunks=(ctab, len(chunks) // 2), **kw)
    assert float((p3 - p4).abs().max()) <= 2e-7 * float(p3.abs().max())
    off_word = torch.zeros(1, dtype=torch.int32, device=dev)
    p5 = p.clone()
    H.adamw_flat(p5, gr, m.clone(), v.clone(), active=off_word, mat=(tab, len(rows), tiles), chunks=(ctab, len(chunks) // 2), **kw)
    assert torch.equal(p5, p)


def test_training_with_emitted_operands_matches_the_refresh_path(hip, monkeypatch):
    """Three replayed iterations with the optimizer writing the bf16 operands itself against three with REFTR_OPT_EMIT=0 (flat pass +
    operand refresh): same losses, same parameters (to the atomics' ulp), and the operands in use ARE the bf16 image of the masters."""
    from reftr_amd.engine_vg import captured_train_step

    def run(emit):
        monkeypatch.setenv("REFTR_OPT_EMIT", "1" if emit else "0")
        model, crit, s, tg, opt = build()
        assert opt._emit == emit
        losses = [captured_train_step(model, crit, s, tg, opt, None, 0.1)[0] for _ in range(3)]
        cap = next(iter(model.__dict__["_captured_steps"].values()))
        if emit:
            assert opt._last_emitted
        cap.flush()
        torch.cuda.synchronize()
        return losses, model, opt

    l0, m0, _ = run(False)
    l1, m1, _ = run(True)
    assert abs(l1[0] - l0[0]) <= 1e-6 * abs(l0[0])
    for a, b in zip(l1, l0):
        assert abs(a - b) <= 2e-3 * abs(b), (l1, l0)
    d = float((m1.store.flat_p - m0.store.flat_p).norm() / m0.store.flat_p.norm())
    assert d < 1e-5, d
    for key, l in m1.net.lins.items():
        assert torch.equal(l.W, l.w32.to(torch.bfloat16)), key
        assert torch.equal(l.WT, l.w32.to(torch.bfloat16).t().contiguous()), key
    for c in m1.body.all_convs:
        if c.trainable:
            w = (m1.store.phys(c.name) * m1.body.bn[c.bn][0].view(-1, 1, 1)).to(torch.bfloat16)
            assert torch.equal(m1.body.W[c.name], w), c.name
            assert torch.equal(m1.body.W[c.name + ".t"], w.permute(2, 1, 0).contiguous()), c.name


# ----------------------------------------------------------------------------------------------------------------------------------
def _norm_case(kind):
    from reftr_amd.models import layout as L
    from reftr_amd.models.criterion import CriterionVGMultiPhrase, CriterionVGOnePhraseSeg
    from reftr_amd.models.reftr_transformer import RefTR
    from reftr_amd.optim import FusedAdamW
    masks = kind == "seg"
    ocfg = O.Cfg(enc_layers=2, dec_layers=2, bert=O.BertCfg(layers=2), masks=masks, aux_loss=not masks)
    cfg = L.ModelConfig(enc_layers=2, dec_layers=2, bert=L.BertConfig(layers=2), masks=masks)
    model = RefTR(cfg, device="cuda", aux_loss=not masks)
    model.load_state_dict(formula_state(param_shapes(ocfg)), strict=True)
    torch.manual_seed(3)
    model.store.P["bbox_embed.layers.2.weight"].normal_(0, 0.02)
    model.mark_dirty()
    model.train()
    if masks:
        crit = CriterionVGOnePhraseSeg(O.weight_dict(ocfg), ["masks", "boxes"])
        samples, targets = make_inputs("seg_single", B=2, H=96, W=128, L=12)
        g = torch.Generator().manual_seed(1)
        targets = [dict(t, masks=(torch.rand(1, 96, 128, generator=g) > 0.5)) for t in targets]
    else:
        crit = CriterionVGMultiPhrase(O.weight_dict(ocfg), ["boxes"])
        samples, targets = make_inputs("norm_" + kind, B=3, H=96, W=128, L=12, n_phrase=3 if kind == "multi" else 0)
    s, tg = to_cuda(samples, targets)
    return model, crit, s, tg, FusedAdamW(model)


@pytest.mark.parametrize("kind", ["single", "multi", "seg"])
@pytest.mark.parametrize("fast", [True, False])
def test_clip_norm_collected_in_the_weight_gradient_epilogues_equals_the_norm_of_the_buffer(hip, kind, fast):
    """engine_vg.py:62-63's total norm without a pass over the 607 MB buffer: the weight-gradient launches add |dw after|^2 -
    |dw before|^2 of what they write (in-kernel for the second-generation kernels and their reduction, a pass over the freshly
    written matrix for the small first-generation / M <= 16 ones), rt_sqnorm_finish adds the atomically accumulated tensors.
    Cases: one contribution per matrix (single phrase), two BERT passes + real decoder self-attention (multi-phrase: accumulating
    second writes), the RES head (unregistered convolution gradients: counted with the complement); overwrite mode and the full
    clear; a second backward onto the same gradients (accumulation) keeps the accumulator exact."""
    from reftr_amd.engine_vg import _total, _zero_grad
    model, crit, s, tg, opt = _norm_case(kind)
    st = model.store
    assert st.fused_norm and st.sq_slots is not None
    st.flat_g.normal_()                                   # stale garbage the step must not count
    for rep in range(2):
        out = model(s)
        total = _total(crit, crit(out, tg))
        if rep == 0:
            _zero_grad(opt) if fast else opt.zero_grad()
            assert st.norm_valid
        total.backward()                                  # rep 1: accumulates onto rep 0's gradients, no zero_grad in between
    gn = opt.clip_grad_norm_(0.1)
    assert not st.norm_valid                              # consumed: a second clip without a new zero_grad reads the buffer again
    fused = float(opt.sq)
    ref = float(st.flat_g.double().pow(2).sum())
    assert ref > 0 and abs(fused - ref) <= 2e-5 * ref, (kind, fast, fused, ref)
    opt._sqnorm_all()
    assert abs(float(opt.sq) - ref) <= 2e-5 * ref          # the old path, for reference


def test_replayed_step_reports_the_same_gradient_norm_with_and_without_the_fused_norm(hip, monkeypatch):
    from reftr_amd.engine_vg import captured_train_step
    res = []
    for on in ("1", "0"):
        monkeypatch.setenv("REFTR_FUSED_NORM", on)
        model, crit, s, tg, opt = build()
        assert model.store.fused_norm == (on == "1")
        res.append([float(captured_train_step(model, crit, s, tg, opt, None, 0.1)[3]) for _ in range(3)])
    assert abs(res[0][0] - res[1][0]) <= 1e-5 * res[1][0], res
    for a, b in zip(res[0], res[1]):
        assert abs(a - b) <= 2e-2 * b, res


def test_sparse_state_adamw_is_bit_exact_and_skips_untouched_pieces(hip):
    """rt_adamw_mat with state bytes (matrices without bf16 operands = the embedding tables): a KB row piece whose m / v are zero and
    whose gradient is zero is left alone without reading p / m / v.  The mat job WITH the bytes must equal the same job WITHOUT them
    bit for bit -- over several steps, with rows entering the touched set late, with weight decay that rounds to a no-op (the skip
    is taken) and with one that does not (the skip is never taken)."""
    H = hip
    dev = "cuda"
    N, K = 100, 768
    n = N * K
    g0 = torch.Generator(device="cpu").manual_seed(3)
    kt = (K + 255) // 256
    tiles = ((N + 31) // 32) * kt
    for lr, wd, expect_skip in ((1e-5, 1e-4, True), (1e-2, 0.1, False)):
        p = torch.randn(n, generator=g0).to(dev)
        pa, ma, va = p.clone(), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
        pb, mb, vb = p.clone(), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
        flags = torch.ones(N * kt, dtype=torch.uint8, device=dev)
        tab_a = torch.tensor([[0, 0, 0, 0, N, 1, K, 0]], dtype=torch.int64).to(dev)
        tab_b = torch.tensor([[0, flags.data_ptr(), 0, 0, N, 1, K, 0]], dtype=torch.int64).to(dev)
        touched = set()
        for step, rows in enumerate(([3, 50], [3], [], [77, 50]), start=1):
            gr = torch.zeros(N, K)
            for r in rows:
                gr[r] = torch.randn(K, generator=g0) * 0.1
            if step == 4:
                gr[12, 300:310] = 0.5                     # a single piece of a row
            gr = gr.reshape(-1).to(dev)
            touched |= set(rows)
            sq = (gr.double() ** 2).sum().float().reshape(1)
            kw = dict(step=step, ranges=[(0, n, lr, wd)], gnorm_sq=sq, max_norm=0.1)
            H.adamw_flat(pa, gr, ma, va, mat=(tab_a, 1, tiles), **kw)
            H.adamw_flat(pb, gr, mb, vb, mat=(tab_b, 1, tiles), **kw)
            torch.cuda.synchronize()
            assert torch.equal(pa, pb) and torch.equal(ma, mb) and torch.equal(va, vb), (lr, step)
        f = flags.view(N, kt).cpu()
        live = (mb.view(N, K) != 0) | (vb.view(N, K) != 0)
        want = torch.stack([live[:, c * 256:(c + 1) * 256].any(dim=1) for c in range(kt)], dim=1)
        assert torch.equal(f.bool(), want.cpu())          # the bytes say exactly which pieces hold state
        assert int(f.sum()) == 3 * len(touched) + 1       # rows 3, 50, 77 whole + one piece of row 12
        if expect_skip:
            assert torch.equal(pb.view(N, K)[0], p.view(N, K)[0])        # an untouched row never moved (decay rounds to a no-op)
        else:
            assert not torch.equal(pb.view(N, K)[0], p.view(N, K)[0])    # real weight decay: every row moves, nothing was skipped


def test_sparse_state_bytes_follow_moments_written_from_outside(hip):
    """FusedAdamW notices m / v written from outside (a restored snapshot, load_state_dict) through the tensors' version counters
    and resets the state bytes to 'unknown' before the next launch: a trajectory restored from a snapshot -- with the bytes
    deliberately falsified to 'no state anywhere' -- equals the original one bit for bit."""
    from reftr_amd.engine_vg import train_step
    model, crit, s, tg, opt = build(dec_layers=2, B=2)
    model.eval()                                           # no dropout: the two passes see the same masks (none)
    train_step(model, crit, s, tg, opt, None, max_norm=0.1)
    assert opt._sparse_flags, "the embedding tables run as sparse-state jobs"
    f = max(opt._sparse_flags.values(), key=lambda t: t.numel())          # the word embeddings
    torch.cuda.synchronize()
    assert 0 < int(f.sum()) < f.numel() // 8                              # only the batch's tokens hold state
    st = model.store
    snap = (st.flat_p.clone(), opt.m.clone(), opt.v.clone(), opt.step_count)
    for _ in range(2):
        train_step(model, crit, s, tg, opt, None, max_norm=0.1)
    torch.cuda.synchronize()
    end = (st.flat_p.clone(), opt.m.clone(), opt.v.clone())
    st.flat_p.copy_(snap[0]); opt.m.copy_(snap[1]); opt.v.copy_(snap[2]); opt.step_count = snap[3]; opt.step_dev.fill_(snap[3])
    model.mark_dirty(full=True)
    for t in opt._sparse_flags.values():
        t.zero_()                                          # a lie the version check must override
    for _ in range(2):
        train_step(model, crit, s, tg, opt, None, max_norm=0.1)
    torch.cuda.synchronize()
    # the backward has fp32 atomics (bias / LayerNorm / embedding gradients): run to run 1 ulp, not bit-equal -- the moments of the
    # embedding rows that hold state must be there again, which a skipped piece would have left at the snapshot's values
    emb = "lang_backbone.embeddings.word_embeddings.weight"
    m_end, m_now, m_snap = st.view_of(end[1], emb), st.view_of(opt.m, emb), st.view_of(snap[1], emb)
    rows = (m_end != 0).any(dim=1)
    assert int(rows.sum()) > 0
    assert float((m_now[rows] - m_end[rows]).abs().max()) <= 1e-3 * float(m_end[rows].abs().max())
    assert float((m_now[rows] - m_snap[rows]).abs().max()) > 10 * float((m_now[rows] - m_end[rows]).abs().max())
    assert float((st.flat_p - end[0]).abs().max()) <= 1e-5


@pytest.mark.parametrize("head_fuse", ["1", "0"])
def test_a_non_finite_loss_never_arms_its_update(coop_env, monkeypatch, head_fuse):
    """Round 5: the reference stops BEFORE the update when the loss is not finite (engine_vg.py:53-58).  That decision is also taken on
    the device, so that it holds for a caller that launches iteration i + 1 before it has read iteration i:
    rt_finish_step neither advances the step counter nor arms the deferred AdamW pass of an iteration whose weighted total is not
    finite -- however often the graph is replayed, weights and moments stay bit-identical -- and a finite iteration behind it updates
    as usual.  Both head paths (rt_head_loss's own total, the launched head's weighted_total)."""
    from reftr_amd.engine_vg import CapturedTrainStep
    monkeypatch.setenv("REFTR_HEAD_FUSE", head_fuse)
    model, crit, s, tg, opt = build()
    cap = CapturedTrainStep(model, crit, opt, 0.1, s, tg)
    assert cap.deferred
    cap.flush()
    torch.cuda.synchronize()
    sb, tb = cap.batch
    good = tb[0]["boxes"].clone()
    tb[0]["boxes"].fill_(float("nan"))                        # a poisoned target box: every loss term of the batch sum is NaN
    p0, m0, v0, sd0 = model.store.flat_p.clone(), opt.m.clone(), opt.v.clone(), int(opt.step_dev)
    for _ in range(3):
        l, _, _ = cap(sb, tb)
    torch.cuda.synchronize()
    assert not torch.isfinite(l).all()
    assert torch.equal(model.store.flat_p, p0) and torch.equal(opt.m, m0) and torch.equal(opt.v, v0)
    assert int(opt.step_dev) == sd0 and int(opt.active) == 0
    tb[0]["boxes"].copy_(good)
    l, _, _ = cap(sb, tb)                                     # a finite iteration: armed ...
    torch.cuda.synchronize()
    assert torch.isfinite(l).all() and int(opt.active) == 1 and int(opt.step_dev) == sd0 + 1
    assert torch.equal(model.store.flat_p, p0)                # ... and applied by the next replay (or flush)
    cap.flush()
    torch.cuda.synchronize()
    assert not torch.equal(model.store.flat_p, p0)


def test_finish_stats_is_finish_step_plus_the_loops_numbers(hip):
    """rt_finish_stats against the four launches it replaces (rt_finish_step, sqrt, dtype conversion, concatenation): counters and
    veto behaviour identical, grad_norm = sqrt(sq) * scale, stats = [losses | failure word | norm] bit for bit."""
    H = hip
    dev = "cuda"
    losses = torch.tensor([0.25, 1.5, 3.0, 0.125, 7.0], dtype=torch.float32, device=dev)
    srcs = [losses[i:i + 1] for i in (3, 0, 4)]                           # scalars wherever they live, in any order
    for veto_word, loss_val, expect_armed in ((0, 2.0, True), (3, 2.0, False), (0, float("nan"), False), (0, float("inf"), False)):
        step = torch.tensor([7], dtype=torch.int32, device=dev); active = torch.tensor([1], dtype=torch.int32, device=dev)
        veto = torch.tensor([veto_word], dtype=torch.int32, device=dev)
        loss = torch.tensor([loss_val], dtype=torch.float32, device=dev)
        sq = torch.tensor([6.25], dtype=torch.float32, device=dev); gn = torch.zeros(1, dtype=torch.float32, device=dev)
        stats = torch.full((5,), -1.0, dtype=torch.float32, device=dev)
        H.finish_stats(step, active, veto, loss, sq, 0.5, gn, srcs=srcs, cond_in_stats=True, stats=stats)
        torch.cuda.synchronize()
        assert int(step) == (8 if expect_armed else 7) and int(active) == (2 if expect_armed else 0)
        assert float(gn) == 1.25
        assert stats.tolist() == [0.125, 0.25, 7.0, float(veto_word), 1.25]
        # the reference behaviour of the single launches
        step2 = torch.tensor([7], dtype=torch.int32, device=dev); active2 = torch.tensor([1], dtype=torch.int32, device=dev)
        H.finish_step(step2, active2, veto, loss)
        assert int(step2) == int(step) and int(active2) == int(active)
    # no veto word, no stats: counters and norm only
    step = torch.tensor([0], dtype=torch.int32, device=dev); active = torch.tensor([0], dtype=torch.int32, device=dev)
    sq = torch.tensor([4.0], dtype=torch.float32, device=dev); gn = torch.zeros(1, dtype=torch.float32, device=dev)
    H.finish_stats(step, active, None, None, sq, 1.0, gn)
    assert int(step) == 1 and int(active) == 1 and float(gn) == 2.0
