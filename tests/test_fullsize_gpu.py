"""GPU, at BASELINE.json's FULL sizes (configs[1] = RefCOCO R50 640x640 B=8 L=40 6+6+12 layers; configs[4] = R101 800x800 L=90,
16 phrases x 22 tokens), where the CPU oracle would need minutes per step: size-independent properties of the training
step instead of element-wise comparison.

  * known answer at the reference's initialisation (reftr_transformer.py:131-132 zero-inits the last bbox layer): every
    pred_box is exactly (0.5, 0.5, 0.5, 0.5) and every loss is a closed form of the targets (0.92 per layer
    for the box (0.4, 0.5, 0.3, 0.4)); integer / bool outputs exact;
  * eval-mode forward is bit-deterministic and commutes with a permutation of the batch (samples are independent);
  * shard additivity -- the data-parallel contract (main_vg.py:290-296, criterion.py:176-180): with the global num_boxes
    normaliser the gradient of a batch is the sum of the gradients of its shards, so grad(B=8) = (grad(first 4) + grad(last 4))/2
    where each half is normalised by its own 4 boxes;
  * gradient linearity in the loss weights.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def closed_form_losses(targets):
    """loss_bbox / loss_giou of the constant prediction (0.5, 0.5, 0.5, 0.5) (criterion.py:131-163, box_ops.py:17-69), float64."""
    t = np.concatenate([x["boxes"].double().cpu().numpy() for x in targets], 0)
    n = max(len(t), 1)
    l1 = np.abs(t - 0.5).sum()
    px0 = py0 = 0.25; px1 = py1 = 0.75
    tx0, ty0, tx1, ty1 = t[:, 0] - t[:, 2] / 2, t[:, 1] - t[:, 3] / 2, t[:, 0] + t[:, 2] / 2, t[:, 1] + t[:, 3] / 2
    iw = np.clip(np.minimum(px1, tx1) - np.maximum(px0, tx0), 0, None)
    ih = np.clip(np.minimum(py1, ty1) - np.maximum(py0, ty0), 0, None)
    inter = iw * ih
    union = 0.25 + (tx1 - tx0) * (ty1 - ty0) - inter
    cw = np.maximum(px1, tx1) - np.minimum(px0, tx0); ch = np.maximum(py1, ty1) - np.minimum(py0, ty0)
    giou = inter / union - (cw * ch - union) / (cw * ch)
    return l1 / n, (1 - giou).sum() / n


def weight_dict(nl):
    wd = {"loss_giou": 1.0, "loss_bbox": 1.0}
    wd.update({f"{k}_{i}": v for i in range(nl - 1) for k, v in list(wd.items())})
    return wd


def cfg2_batch(B=8, S=640, L=40, seed=1234):
    import bench
    from reftr_amd.util.misc import NestedTensor
    samples, targets = bench.synth_batch(B, S, S, L, "cuda", seed)
    s = {k: v.cuda() for k, v in samples.items() if k not in ("img", "img_mask")}
    s["img"] = NestedTensor(samples["img"].cuda(), samples["img_mask"].cuda())
    return s, [{k: v.cuda() for k, v in t.items()} for t in targets]


def take(s, tg, idx):
    from reftr_amd.util.misc import NestedTensor
    i = torch.as_tensor(idx, device="cuda")
    out = {k: v[i].contiguous() for k, v in s.items() if k != "img"}
    out["img"] = NestedTensor(s["img"].tensors[i].contiguous(), s["img"].mask[i].contiguous())
    return out, [tg[j] for j in idx]


@pytest.fixture(scope="module")
def cfg2():
    from reftr_amd.models import layout as L
    from reftr_amd.models.criterion import CriterionVGMultiPhrase
    from reftr_amd.models.reftr_transformer import RefTR
    cfg = L.ModelConfig()
    model = RefTR(cfg, device="cuda", aux_loss=True)
    crit = CriterionVGMultiPhrase(weight_dict(cfg.dec_layers), ["boxes"])
    s, tg = cfg2_batch()
    return model, crit, s, tg, cfg


def test_cfg2_known_answer_at_reference_init(cfg2):
    model, crit, s, tg, cfg = cfg2
    model.reset_parameters(seed=0)
    model.eval()
    with torch.no_grad():
        out = model(s)
    assert out["pred_boxes"].shape == (8, 1, 1, 4) and len(out["aux_outputs"]) == cfg.dec_layers - 1
    for o in [out] + out["aux_outputs"]:
        assert bool((o["pred_boxes"] == 0.5).all())                  # sigmoid(0), exact
    assert out["phrase_mask"].dtype == torch.bool and bool(out["phrase_mask"].all())
    ld = crit(out, tg)
    lb, lg = closed_form_losses(tg)
    for k, v in ld.items():
        want = lb if "bbox" in k else lg
        assert abs(float(v) - want) < 2e-6 * max(1.0, want), (k, float(v), want)
    total = sum(float(ld[k]) * w for k, w in crit.weight_dict.items())
    assert abs(total - cfg.dec_layers * (lb + lg)) < 1e-4
    # the SURVEY §8c single-box instance of the same closed form
    one = [{"boxes": torch.tensor([[0.4, 0.5, 0.3, 0.4]], dtype=torch.float64)}]
    a, b = closed_form_losses(one)
    assert abs(a - 0.4) < 1e-12 and abs(b - 0.52) < 1e-12        # GIoU = 0.48


def test_cfg2_eval_forward_deterministic_and_batch_equivariant(cfg2):
    model, crit, s, tg, cfg = cfg2
    model.reset_parameters(seed=0)
    torch.manual_seed(7)
    model.store.P["bbox_embed.layers.2.weight"].normal_(0, 0.05); model.mark_dirty()
    model.eval()
    with torch.no_grad():
        a = model(s)["pred_logits"].clone()
        b = model(s)["pred_logits"].clone()
        perm = [3, 0, 7, 1, 6, 2, 5, 4]
        sp, _ = take(s, tg, perm)
        c = model(sp)["pred_logits"].clone()
    assert torch.equal(a, b)                                          # no atomics / no run-to-run freedom in the forward
    assert float(a.std()) > 1e-3                                       # a non-degenerate head: the comparison below means something
    # every sample is computed from its own rows only: the same numbers wherever it sits in the batch
    assert float((c - a[:, perm]).abs().max()) <= 1e-5 * float(a.abs().max())


def _grads(model, crit, s, tg):
    model.store.flat_g.zero_()
    out = model(s)
    ld = crit(out, tg)
    loss = sum(ld[k] * w for k, w in crit.weight_dict.items())
    loss.backward()
    torch.cuda.synchronize()
    return model.store.flat_g.clone(), float(loss.detach())


def test_cfg2_shard_additivity_and_linearity(cfg2):
    from reftr_amd.models import layout as L
    model, crit, s, tg, cfg = cfg2
    model.reset_parameters(seed=0)
    torch.manual_seed(7)
    model.store.P["bbox_embed.layers.2.weight"].normal_(0, 0.05); model.mark_dirty()
    model.eval()                                                       # dropout off: the property is about the arithmetic
    g_full, l_full = _grads(model, crit, s, tg)
    s0, t0 = take(s, tg, [0, 1, 2, 3]); s1, t1 = take(s, tg, [4, 5, 6, 7])
    g0, l0 = _grads(model, crit, s0, t0)
    g1, l1 = _grads(model, crit, s1, t1)
    assert torch.isfinite(g_full).all() and float(g_full.norm()) > 0
    assert abs(l_full - 0.5 * (l0 + l1)) < 1e-5 * abs(l_full)
    want = 0.5 * (g0 + g1)
    st = model.store
    for grp in (L.GROUP_MAIN, L.GROUP_BACKBONE, L.GROUP_BERT):
        b, e = st.group_range[grp]
        x, y = g_full[b:e].double(), want[b:e].double()
        relerr = float((x - y).norm() / y.norm())
        cos = float((x @ y) / (x.norm() * y.norm()))
        # per-sample activation gradients differ only by the power-of-two loss scale; weight gradients then differ by fp32
        # summation order (split-M partials, atomics) -- far below the bf16 noise floor of the e2e parity tests
        assert relerr < 2e-3 and cos > 0.99999, (grp, relerr, cos)
    # linearity in the loss weights: 2 x weights -> 2 x gradients (power of two: commutes with every rounding)
    saved = dict(crit.weight_dict)
    try:
        for k in crit.weight_dict:
            crit.weight_dict[k] = 2.0 * saved[k]
        g2, l2 = _grads(model, crit, s, tg)
    finally:
        crit.weight_dict.update(saved)
    assert abs(l2 - 2 * l_full) < 1e-5 * abs(l2)
    assert float((g2 - 2 * g_full).norm() / g2.norm()) < 2e-3


def cfg5_batch(B=8, S=800, L=90, P=16, Lp=22):
    from reftr_amd.util.misc import NestedTensor
    g = torch.Generator().manual_seed(5)
    img = torch.randn(B, 3, S, S, generator=g); mask = torch.zeros(B, S, S, dtype=torch.bool)
    for b in range(1, B, 2):
        mask[b, :, (S * 3) // 4:] = True; img[b, :, :, (S * 3) // 4:] = 0
    ids = torch.zeros(B, L, dtype=torch.long); sm = torch.zeros(B, L, dtype=torch.long)
    ph = torch.zeros(B, P, Lp, dtype=torch.long); pm = torch.zeros(B, P, Lp, dtype=torch.long)
    pl = torch.zeros(B, P, dtype=torch.long); pr = torch.ones(B, P, dtype=torch.long)
    targets, valid = [], torch.zeros(B, P, dtype=torch.bool)
    for b in range(B):
        n = int(torch.randint(40, L + 1, (1,), generator=g))
        ids[b, :n] = torch.randint(1000, 30000, (n,), generator=g); ids[b, 0] = 101; ids[b, n - 1] = 102; sm[b, :n] = 1
        nv = max(1, P - 2 * b)                                        # ragged: 16, 14, ... valid phrases (1-8 .. 16 in the reference data)
        for j in range(P):
            if j < nv:
                k = 3 + (j % 5)
                ph[b, j, :k] = torch.randint(1000, 30000, (k,), generator=g); ph[b, j, 0] = 101; ph[b, j, k - 1] = 102; pm[b, j, :k] = 1
                pl[b, j] = 1 + 2 * j; pr[b, j] = 1 + 2 * j + (k - 2); valid[b, j] = True
            else:                                                      # padding phrase "[CLS] [SEP]": token 2 is padding -> invalid
                ph[b, j, 0] = 101; ph[b, j, 1] = 102; pm[b, j, :2] = 1
        u = torch.rand(nv, 4, generator=g)
        targets.append({"boxes": torch.stack([0.3 + 0.4 * u[:, 0], 0.3 + 0.4 * u[:, 1], 0.1 + 0.4 * u[:, 2], 0.1 + 0.4 * u[:, 3]], -1).cuda(),
                        "labels": torch.zeros(nv, dtype=torch.long, device="cuda")})
    s = {"img": NestedTensor(img.cuda(), mask.cuda()), "sentence": ids.cuda(), "sentence_mask": sm.cuda(), "phrase": ph.cuda(),
         "phrase_mask": pm.cuda(), "phrase_pos_l": pl.cuda(), "phrase_pos_r": pr.cuda()}
    return s, targets, valid


def test_cfg5_flickr_r101_800_known_answer_and_one_step():
    """configs[4]: ResNet-101, 800x800, S = 90 + 625, 16 queries per image with ragged validity."""
    from reftr_amd.engine_vg import train_step
    from reftr_amd.models import layout as L
    from reftr_amd.models.criterion import CriterionVGMultiPhrase
    from reftr_amd.models.reftr_transformer import RefTR
    from reftr_amd.optim import FusedAdamW
    cfg = L.ModelConfig(resnet_layers=(3, 4, 23, 3))
    model = RefTR(cfg, device="cuda", aux_loss=True)
    n_params = sum(p.numel() for p in model.parameters())
    assert 170e6 < n_params < 172e6, n_params                          # RefTR-R101: 151.76 M + 18.99 M (layer3 blocks 7..23)
    crit = CriterionVGMultiPhrase(weight_dict(cfg.dec_layers), ["boxes"])
    s, tg, valid = cfg5_batch()
    model.eval()
    with torch.no_grad():
        out = model(s)
    assert out["pred_boxes"].shape == (8, 16, 1, 4)
    assert torch.equal(out["phrase_mask"].view(8, 16).cpu(), valid)    # reftr_transformer.py:235-238, exact
    assert bool((out["pred_boxes"] == 0.5).all())
    ld = crit(out, tg)
    lb, lg = closed_form_losses(tg)                                    # normaliser: 72 valid phrases of 128 slots
    assert sum(len(t["labels"]) for t in tg) == int(valid.sum()) == 72
    for k, v in ld.items():
        want = lb if "bbox" in k else lg
        assert abs(float(v) - want) < 5e-6 * max(1.0, want), (k, float(v), want)
    # one full training step (dropout on, clip 0.1, AdamW): finite everywhere, norm clipped, invalid slots get no gradient
    torch.manual_seed(3)
    model.store.P["bbox_embed.layers.2.weight"].normal_(0, 0.02); model.mark_dirty()
    opt = FusedAdamW(model, lr=1e-4, lr_backbone=1e-5, weight_decay=1e-4)
    model.train()
    before = model.store.flat_p.clone()
    lv, _, _, gn = train_step(model, crit, s, tg, opt, None, 0.1)
    torch.cuda.synchronize()
    assert np.isfinite(lv) and torch.isfinite(model.store.flat_g).all() and torch.isfinite(model.store.flat_p).all()
    assert float(gn) > 0.1                                             # the raw norm (clip_grad_norm_ returns the pre-clip norm)
    step = (model.store.flat_p - before).abs().max()
    assert 0 < float(step) <= 1.01e-4 * 1.0 + 1e-4 * 1e-4 * float(before.abs().max()) + 1e-7   # |Adam step| <= lr (+ decay)
