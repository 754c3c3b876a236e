"""CPU: host-side logic of the product package — parameter layout / flat storage / state_dict contract,
factory and optimizer bookkeeping, post-processing (exact vs golden), and the guarantee that the product path
has NO CPU fallback and never touches oracle/."""
import argparse
import os
import re

import numpy as np
import pytest
import torch

from oracle import reftr_oracle as O
from oracle.shapes import param_shapes
from oracle.weights import formula_state

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def small():
    from reftr_amd.models import layout as L
    return L.ModelConfig(enc_layers=2, dec_layers=2, bert=L.BertConfig(layers=2)), \
        O.Cfg(enc_layers=2, dec_layers=2, bert=O.BertCfg(layers=2))


def ref_args(**kw):
    a = argparse.Namespace(hidden_dim=256, nheads=8, enc_layers=2, dec_layers=2, dim_feedforward=2048, dropout=0.1,
                           num_feature_levels=1, max_lang_seq=128, position_embedding="sine", lr_backbone=1e-5, masks=False,
                           backbone="resnet50", dilation=False, num_queries_per_phrase=1, aux_loss=True, ablation="none",
                           freeze_bert=False, giou_loss_coef=1.0, bbox_loss_coef=1.0, device="cpu", no_decoder=False,
                           bert_layers=2, lr=1e-4, weight_decay=1e-4)
    for k, v in kw.items():
        setattr(a, k, v)
    return a


def test_state_dict_contract_and_flat_views():
    from reftr_amd.models.reftr_transformer import RefTR
    cfg, ocfg = small()
    m = RefTR(cfg, device="cpu")
    shapes = param_shapes(ocfg)                       # independent restatement of the reference key table
    sd = m.state_dict()
    assert {k: tuple(v.shape) for k, v in sd.items()} == shapes
    trainable = sorted(n for n, p in m.named_parameters() if p.requires_grad)
    assert trainable == sorted(k for k in shapes if O.is_trainable(k))       # backbone.py:87-89
    P = formula_state(shapes)
    m.load_state_dict(P, strict=True)
    st = m.store
    # conv weights are stored channels-last inside the flat buffer
    k = "img_backbone.0.body.layer3.1.conv2.weight"
    assert torch.equal(st.phys(k), P[k].permute(0, 2, 3, 1).reshape(256, 9, 256))
    assert torch.equal(m.state_dict()[k], P[k]) and m.state_dict()[k].is_contiguous()
    # BERT q/k/v are adjacent: the packed [3H, H] view is free
    q = "lang_backbone.encoder.layer.1.attention.self.query.weight"
    packed = st.packed(q, 3)
    assert torch.equal(packed, torch.cat([P[q], P[q.replace("query", "key")], P[q.replace("query", "value")]]))
    # gradients are views of ONE buffer; parameters keep their .grad across zero_grad
    named = dict(m.named_parameters())
    assert named[k].grad.data_ptr() == st.G[k].data_ptr()
    assert st.flat_g.numel() == st.flat_p.numel() and st.flat_p.numel() % 4 == 0
    # lr groups (main_vg.py:29-33) are three contiguous ranges
    from reftr_amd.models import layout as L
    r = st.group_range
    assert r[L.GROUP_MAIN][0] == 0 and r[L.GROUP_MAIN][1] == r[L.GROUP_BACKBONE][0] and r[L.GROUP_BACKBONE][1] == r[L.GROUP_BERT][0]
    for n, (b, off) in st.offset.items():
        if b == "p":
            lo, hi = r[L.lr_group(n)]
            assert lo <= off < hi, n


def test_build_reftr_factory_and_weight_dict():
    from reftr_amd import build_reftr
    model, criterion, post = build_reftr(ref_args())
    assert set(post) == {"bbox"}
    wd = criterion.weight_dict                       # models/reftr_transformer.py:320-329
    assert set(wd) == {"loss_giou", "loss_bbox", "loss_giou_0", "loss_bbox_0", "loss_giou_enc", "loss_bbox_enc"}
    # --masks -> RefTRSeg (models/__init__.py:5-7, reftr_segmentation.py:343-391)
    seg_args = ref_args(masks=True, aux_loss=False, reftr_type="transformer_single_phrase", dice_loss_coef=1.0, mask_loss_coef=1.0)
    model, criterion, post = build_reftr(seg_args)
    assert set(post) == {"bbox", "segm"} and model.cfg.masks and not model.aux_loss
    assert set(criterion.weight_dict) == {"loss_giou", "loss_bbox", "loss_dice", "loss_mask", "loss_cem"}
    sd = model.state_dict()
    from oracle.shapes import param_shapes
    ocfg = O.Cfg(enc_layers=2, dec_layers=2, bert=O.BertCfg(layers=2), masks=True, aux_loss=False)
    assert {k: tuple(v.shape) for k, v in sd.items()} == {k: tuple(v) for k, v in param_shapes(ocfg).items()}
    w = sd["mask_head.lay1.weight"]
    assert w.shape == (520, 520, 3, 3) and w.is_contiguous()        # padded storage is invisible outside
    with pytest.raises(NotImplementedError):
        build_reftr(ref_args(masks=True, aux_loss=False, reftr_type="transformer_multi_phrase", dice_loss_coef=1.0, mask_loss_coef=1.0))


def test_unsupported_options_raise():
    from reftr_amd import build_reftr
    with pytest.raises(NotImplementedError):
        build_reftr(ref_args(num_feature_levels=4))
    assert build_reftr(ref_args(bert_model="roberta-base"))[0].cfg.bert.pad_idx == 1
    # a --backbone / --bert_model this build has no geometry for must not silently build ResNet-50 / the base config
    for kw in (dict(backbone="resnet152"), dict(backbone="resnet34"), dict(bert_model="bert-large-uncased"), dict(bert_model="roberta-large")):
        with pytest.raises(NotImplementedError):
            build_reftr(ref_args(**kw))
    assert build_reftr(ref_args(backbone="resnet101"))[0].cfg.resnet_layers == (3, 4, 23, 3)
    # --dilation is built (backbone.py:117-125): layer4 at stride 1, its later 3x3 convolutions dilated by 2 with padding 2
    m = build_reftr(ref_args(dilation=True))[0]
    l4 = m.body.blocks[3]
    assert m.cfg.dilation and [(b.conv2.stride, b.conv2.dil, b.conv2.pad) for b in l4] == [(1, 1, 1), (1, 2, 2), (1, 2, 2)]
    assert l4[0].down.stride == 1 and all(b.conv2.dil == 1 for st in m.body.blocks[:3] for b in st)
    # --dilation with --masks builds since round 6 (reftr_segmentation.py:343-384): RefTRSeg on the dilated backbone
    ms = build_reftr(ref_args(dilation=True, masks=True, aux_loss=False, reftr_type="transformer_single_phrase", dice_loss_coef=1.0, mask_loss_coef=1.0))[0]
    assert ms.cfg.dilation and ms.cfg.masks and ms.seg is not None


def test_oracle_post_process_segm_matches_reference_golden_exactly():
    """The oracle's restatement of PostProcessSegm is pinned to the reference's outputs (tests/golden/seg_single.npz); the
    product's HIP post-processor is checked against the same vectors in tests/test_post_gpu.py."""
    from oracle import reftr_oracle as O
    g = np.load(os.path.join(ROOT, "tests", "golden", "seg_single.npz"))
    res = O.postprocess_segm(torch.from_numpy(g["pred_masks"])[:, :, None], torch.from_numpy(g["post_orig"]), torch.from_numpy(g["post_sizes"]))
    for i in range(2):
        assert torch.equal(res[i][0], torch.from_numpy(g[f"post_masks{i}"]))
        assert torch.equal(res[i][1], torch.from_numpy(g[f"post_masks_origin{i}"]))


def test_post_processors_reject_host_tensors():
    from reftr_amd.models.post_process import PostProcessSegm, PostProcessVGMultiPhrase
    g = np.load(os.path.join(GOLD, "postprocess.npz"))
    with pytest.raises(RuntimeError):
        PostProcessVGMultiPhrase()({"pred_boxes": torch.from_numpy(g["pred"]), "phrase_mask": torch.from_numpy(g["mask"])},
                                   torch.from_numpy(g["sizes"]))
    with pytest.raises(RuntimeError):
        PostProcessSegm()([{}], {"pred_masks": torch.zeros(1, 1, 4, 4)}, torch.tensor([[8, 8]]), torch.tensor([[8, 8]]))


def test_no_cpu_fallback_exists():
    from reftr_amd import build_reftr
    from reftr_amd.util.misc import NestedTensor
    model, criterion, _ = build_reftr(ref_args())
    samples = {"img": NestedTensor(torch.zeros(1, 3, 64, 64), torch.zeros(1, 64, 64, dtype=torch.bool)),
               "sentence": torch.tensor([[101, 2000, 102, 0]]), "sentence_mask": torch.tensor([[1, 1, 1, 0]])}
    with pytest.raises(RuntimeError):                # device tensors required / library or GPU missing: loud, not silent
        model(samples)


def test_product_never_imports_oracle():
    pat = re.compile(r"^\s*(from|import)\s+oracle\b", re.M)
    for d, _, files in os.walk(os.path.join(ROOT, "reftr_amd")):
        for f in files:
            if f.endswith(".py"):
                assert not pat.search(open(os.path.join(d, f)).read()), f
    assert not pat.search(open(os.path.join(ROOT, "bench.py")).read().split("def cpu_baseline")[0])


def test_oracle_postprocess_matches_reference_golden_exactly():
    from oracle import reftr_oracle as O
    g = np.load(os.path.join(GOLD, "postprocess.npz"))
    res = O.postprocess_boxes(torch.from_numpy(g["pred"]), torch.from_numpy(g["mask"]), torch.from_numpy(g["sizes"]), True)
    for i, r in enumerate(res):
        assert torch.equal(r, torch.from_numpy(g[f"boxes{i}"]))


def test_nested_tensor_padding():
    from reftr_amd.util.misc import nested_tensor_from_tensor_list
    nt = nested_tensor_from_tensor_list([torch.ones(3, 4, 6), torch.ones(3, 5, 3)])
    t, m = nt.decompose()
    assert t.shape == (2, 3, 5, 6) and m.dtype == torch.bool
    assert not m[0, :4, :6].any() and m[0, 4:].all() and m[1, :, 3:].all() and not m[1, :5, :3].any()
    assert float(t[1, :, :, 3:].abs().sum()) == 0


def test_optimizer_groups_follow_reference_param_groups():
    from reftr_amd.models.reftr_transformer import RefTR
    from reftr_amd.optim import FusedAdamW
    cfg, _ = small()
    m = RefTR(cfg, device="cpu")
    opt = FusedAdamW(m, lr=1e-4, lr_backbone=1e-5, weight_decay=1e-4)
    lrs = [g["lr"] for g in opt.param_groups]
    assert lrs == [1e-4, 1e-5, 1e-5, 1e-4]           # main_vg.py:234-262 (BERT group also uses lr_backbone; 4th = mask
    assert opt.param_groups[3]["params"] == []      # branch, lr * lr_mask_branch_proj, empty for REC models)
    n = sum(p.numel() for g in opt.param_groups for p in g["params"])
    assert n == sum(p.numel() for p in m.parameters() if p.requires_grad)
    sched = torch.optim.lr_scheduler.StepLR(opt, 2)  # engine_vg.py:67: stepped per iteration
    opt.zero_grad()
    assert float(m.store.flat_g.abs().sum()) == 0 and dict(m.named_parameters())["bbox_embed.layers.0.weight"].grad is not None
    for _ in range(2):
        sched.step()
    assert abs(opt.param_groups[0]["lr"] - 1e-5) < 1e-12


def test_sgd_flag_builds_the_momentum_optimizer_with_torch_sgd_state_format():
    """--sgd (main_vg.py:263-265): build_optimizer returns the fused SGD(momentum 0.9) on the reference's four param groups; its
    state_dict has torch.optim.SGD's layout (momentum_buffer per parameter index) and round-trips."""
    from reftr_amd.models.reftr_transformer import RefTR
    from reftr_amd.optim import FusedAdamW, FusedSGD, build_optimizer
    cfg, _ = small()
    m = RefTR(cfg, device="cpu")
    opt = build_optimizer(m, ref_args(sgd=True))
    assert isinstance(opt, FusedSGD) and [g["lr"] for g in opt.param_groups] == [1e-4, 1e-5, 1e-5, 1e-4]
    assert all(g["momentum"] == 0.9 for g in opt.param_groups) and opt.v.numel() == 4
    assert type(build_optimizer(m, ref_args())) is FusedAdamW
    opt.m.uniform_(-1, 1); opt.step_count = 1
    sd = opt.state_dict()
    n_params = sum(len(g["params"]) for g in sd["param_groups"])
    assert sorted(sd["state"]) == list(range(n_params)) and set(sd["state"][0]) == {"momentum_buffer"}
    named = dict(m.named_parameters())
    order = [n for ns in opt._names for n in ns]
    assert sd["state"][3]["momentum_buffer"].shape == named[order[3]].shape
    opt2 = build_optimizer(RefTR(cfg, device="cpu"), ref_args(sgd=True))
    opt2.load_state_dict(sd)
    assert opt2.step_count == 1            # (the flat buffers' alignment padding between groups is not part of any parameter)
    assert all(torch.equal(opt2.model.store.view_of(opt2.m, n), m.store.view_of(opt.m, n)) for n in order)
    ref = torch.optim.SGD([{"params": g["params"], "lr": g["lr"]} for g in opt.param_groups if g["params"]], lr=1e-4, momentum=0.9, weight_decay=1e-4)
    assert set(ref.state_dict()["param_groups"][0]) - {"params"} <= set(sd["param_groups"][0]) | {"initial_lr"}


def test_lr_backbone_zero_freezes_the_resnet_like_the_reference():
    """--lr_backbone 0 (models/modeling/backbone.py:87-89,150: train_backbone = False): every ResNet parameter has requires_grad
    False, leaves the optimizer's param groups (the backbone group is empty, like the reference's filtered list) and the flat
    trainable buffer; the state_dict contract is unchanged."""
    from reftr_amd import build_reftr
    from reftr_amd.models import layout as L
    from reftr_amd.optim import build_optimizer
    model, criterion, _ = build_reftr(ref_args(lr_backbone=0.0))
    ref_model, _, _ = build_reftr(ref_args())
    assert not model.cfg.train_backbone and ref_model.cfg.train_backbone
    assert set(model.state_dict()) == set(ref_model.state_dict())
    named = dict(model.named_parameters())
    assert all(not p.requires_grad for n, p in named.items() if n.startswith("img_backbone."))
    assert all(p.requires_grad for n, p in named.items() if not n.startswith("img_backbone."))
    st = model.store
    b, e = st.group_range[L.GROUP_BACKBONE]
    assert b == e
    opt = build_optimizer(model, ref_args(lr_backbone=0.0))
    assert opt.param_groups[1]["params"] == [] and len(opt.param_groups[0]["params"]) > 0
    n_ref = sum(p.numel() for p in ref_model.parameters() if p.requires_grad)
    n = sum(p.numel() for p in model.parameters() if p.requires_grad)
    assert n_ref - n == sum(p.numel() for nme, p in ref_model.named_parameters() if nme.startswith("img_backbone.") and p.requires_grad)


def test_reference_param_order_matches_fixture():
    """layout.reference_param_order == the imported reference's named_parameters() order (fixture minted in the build
    container from RefTR / RefTRSeg themselves, oracle/gen_golden*.py recipe)."""
    from reftr_amd.models import layout as L
    g = np.load(os.path.join(ROOT, "tests", "golden", "param_order.npz"))
    for tag, masks in (("rec", False), ("seg", True)):
        cfg = L.ModelConfig(enc_layers=2, dec_layers=2, bert=L.BertConfig(layers=2), masks=masks)
        assert L.reference_param_order(cfg) == [str(x) for x in g[tag]]
    g2 = np.load(os.path.join(ROOT, "tests", "golden", "e2e_learned_pos.npz"))     # --position_embedding learned: Joiner[1] holds parameters
    cfg = L.ModelConfig(enc_layers=2, dec_layers=2, bert=L.BertConfig(layers=2), pos_learned=True)
    assert L.reference_param_order(cfg) == [str(x) for x in g2["param_order"]]
    assert L.lr_group("img_backbone.1.row_embed.weight") == L.GROUP_MAIN          # lr_backbone_names = ['img_backbone.0'], main_vg.py:29


@pytest.mark.parametrize("masks", [False, True])
def test_optimizer_state_roundtrip_with_torch_adamw(masks):
    """checkpoint['optimizer'] compatibility (main_vg.py:320-324,377-384): FusedAdamW.state_dict() loads into a real
    torch.optim.AdamW built with the reference's param groups, and that optimizer's state_dict loads back."""
    from reftr_amd.models import layout as L
    from reftr_amd.models.reftr_transformer import RefTR
    from reftr_amd.optim import FusedAdamW
    cfg = L.ModelConfig(enc_layers=1, dec_layers=1, bert=L.BertConfig(layers=1), masks=masks)
    model = RefTR(cfg, device="cpu")
    opt = FusedAdamW(model, lr=1e-4, lr_backbone=1e-5, weight_decay=1e-4)
    g = torch.Generator().manual_seed(5)
    opt.m.copy_(torch.randn(opt.m.shape, generator=g)); opt.v.copy_(torch.rand(opt.v.shape, generator=g))
    opt.step_count = 7
    sd = opt.state_dict()
    named = dict(model.named_parameters())
    order = L.reference_param_order(cfg)

    def match(n, keys):
        return any(k in n for k in keys)
    bk, bert, mb = ["img_backbone.0"], ["lang_backbone"], ["bbox_attention", "mask_head"]
    groups = [
        {"params": [named[n] for n in order if not match(n, bk) and not match(n, bert) and not match(n, mb)], "lr": 1e-4},
        {"params": [named[n] for n in order if match(n, bk)], "lr": 1e-5},
        {"params": [named[n] for n in order if match(n, bert)], "lr": 1e-5},
        {"params": [named[n] for n in order if match(n, mb)], "lr": 1e-4}]
    ref = torch.optim.AdamW(groups, lr=1e-4, weight_decay=1e-4)       # main_vg.py:234-268
    ref.load_state_dict(sd)
    k = "vl_transformer.encoder.layers.0.linear1.weight"
    assert torch.equal(ref.state[named[k]]["exp_avg"], model.store.view_of(opt.m, k))
    if masks:
        k = "mask_head.lay1.weight"       # padded storage <-> logical [520, 520, 3, 3] tensor
        assert ref.state[named[k]]["exp_avg_sq"].shape == (520, 520, 3, 3)
        assert torch.equal(ref.state[named[k]]["exp_avg_sq"], model.store.view_of(opt.v, k))
    assert float(ref.state[named[k]]["step"]) == 7.0
    back = FusedAdamW(model, lr=3e-4, lr_backbone=3e-5, weight_decay=0.0)
    back.load_state_dict(ref.state_dict())
    pad = torch.ones_like(opt.m, dtype=torch.bool)      # compare everything except padding (never part of a state_dict)
    for n in order:
        model.store.view_of(pad, n).fill_(False)
    assert torch.equal(back.m[~pad], opt.m[~pad]) and torch.equal(back.v[~pad], opt.v[~pad]) and back.step_count == 7
    assert [g_["lr"] for g_ in back.param_groups] == [1e-4, 1e-5, 1e-5, 1e-4] and back.param_groups[0]["weight_decay"] == 1e-4


import pytest as _pytest


@_pytest.mark.parametrize("cuts", ["1", "2", "3"])
def test_dp_interleaved_schedule_pairs_bert_parts_with_resnet_stages(monkeypatch, cuts):
    """REFTR_DDP_SCHEDULE=interleave (default): BERT's backward keeps its own stream; the slices that are final together -- a part
    of BERT and the ResNet stage that ran beside it -- are exchanged at one boundary; disjoint and covering.  BERT is walked in
    two legs (default "1": cut at two thirds, or "2": halves -- nothing of BERT is left for the exposed end) or thirds ("3")."""
    from reftr_amd.models import layout as L
    from reftr_amd.models.reftr_transformer import RefTR
    from reftr_amd.parallel import DistributedDataParallel
    monkeypatch.delenv("REFTR_DDP_SCHEDULE", raising=False)
    monkeypatch.setenv("REFTR_DDP_BERT_CUTS", cuts)
    cfg = L.ModelConfig(enc_layers=1, dec_layers=1, bert=L.BertConfig(layers=6))
    m = RefTR(cfg, device="cpu")
    ddp = DistributedDataParallel(m, n_chunks=7)
    assert m.dp_schedule == "interleave" and m.active_boundaries() == ("main", "pair4", "pair3")
    sl = ddp.slice_bounds()
    st = m.store
    spans = sorted(r for v in sl.values() for r in (v if isinstance(v, list) else [v]))
    assert spans[0][0] == 0 and spans[-1][1] == st.flat_g.numel() and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    off = lambda n: st.offset[n][1]                                       # noqa: E731
    q = "lang_backbone.encoder.layer.%d.attention.self.query.weight"
    l3, l4 = off("img_backbone.0.body.layer3.0.conv1.weight"), off("img_backbone.0.body.layer4.0.conv1.weight")
    (ra, rb), (ba, bb) = st.group_range[L.GROUP_BACKBONE], st.group_range[L.GROUP_BERT]
    if cuts == "3":
        assert m.bert_cuts() == {4: "pair4", 2: "pair3"}
        assert sl["pair4"] == [(l4, rb), (off(q % 4), bb)]
        assert sl["pair3"] == [(l3, l4), (off(q % 2), off(q % 4))]
        assert sl["end"] == [(ra, l3), (ba, off(q % 2))]
    else:
        c = 3 if cuts == "2" else 4
        assert m.bert_cuts() == {c: "pair4"}
        assert sl["pair4"] == [(l4, rb), (off(q % c), bb)]
        assert sl["pair3"] == [(l3, l4), (ba, off(q % c))]                # embeddings + the lower layers: BERT is complete here
        assert sl["end"] == [(ra, l3)]                                   # only ResNet layer2 is exchanged exposed
        exposed = sum(b - a for a, b in sl["end"])
        assert exposed < 0.05 * st.flat_g.numel()
    pb = ddp.phase_bounds()
    assert list(pb) == ["main", "pair4", "pair3", "end"]
    chunks = sorted(c for v in pb.values() for c in v)
    assert chunks[0][0] == 0 and chunks[-1][1] == st.flat_g.numel() and all(a[1] == b[0] for a, b in zip(chunks, chunks[1:]))


def test_dp_exchange_slices_follow_the_backward_order(monkeypatch):
    """reftr_amd.parallel: the slice that is final at each backward boundary (main | BERT thirds, walked 11..0 | layer4 |
    rest of the ResNet) -- contiguous, disjoint, covering the flat gradient buffer, cut at parameter boundaries."""
    from reftr_amd.models import layout as L
    from reftr_amd.models.reftr_transformer import RefTR
    from reftr_amd.parallel import DistributedDataParallel
    monkeypatch.setenv("REFTR_DDP_SCHEDULE", "serial")
    cfg = L.ModelConfig(enc_layers=1, dec_layers=1, bert=L.BertConfig(layers=6))
    m = RefTR(cfg, device="cpu")
    ddp = DistributedDataParallel(m, n_chunks=7)
    assert m.active_boundaries() == ("main", "bert_hi", "bert_mid", "bert", "layer4") and not m.dp_mode
    sl = ddp.slice_bounds()
    st = m.store
    spans = sorted(sl.values())
    assert spans[0][0] == 0 and spans[-1][1] == st.flat_g.numel() and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    off = lambda n: st.offset[n][1]                                       # noqa: E731
    q = "lang_backbone.encoder.layer.%d.attention.self.query.weight"
    assert sl["bert_hi"] == (off(q % 4), st.group_range[L.GROUP_BERT][1])                 # layers 4, 5 + pooler
    assert sl["bert_mid"] == (off(q % 2), off(q % 4)) and sl["bert"] == (st.group_range[L.GROUP_BERT][0], off(q % 2))
    assert sl["bert"][0] <= off("lang_backbone.embeddings.word_embeddings.weight") < sl["bert"][1]
    assert sl["bert_hi"][0] <= off("lang_backbone.pooler.dense.weight") < sl["bert_hi"][1]
    assert sl["layer4"][0] == off("img_backbone.0.body.layer4.0.conv1.weight") and sl["end"][0] == st.group_range[L.GROUP_BACKBONE][0]
    for k in ("decoder.layers.0.linear1.weight", "encoder.layers.0.linear2.bias"):
        assert sl["main"][0] <= off("vl_transformer." + k) < sl["main"][1]
    pb = ddp.phase_bounds()
    chunks = sorted(c for v in pb.values() for c in v)
    assert chunks[0][0] == 0 and chunks[-1][1] == st.flat_g.numel() and all(a[1] == b[0] for a, b in zip(chunks, chunks[1:]))
    ddp.phases = ["main", "bert"]                                         # REFTR_DDP_PHASES: skipped boundaries merge forward
    pb = ddp.phase_bounds()
    assert set(pb) == {"main", "bert", "end"}
    assert sorted(pb["bert"])[0][0] == sl["bert"][0] and sorted(pb["bert"])[-1][1] == sl["bert_hi"][1]
    assert sorted(pb["end"])[0][0] == sl["end"][0] and sorted(pb["end"])[-1][1] == sl["layer4"][1]


def test_bench_gpus_flag_decides_the_launch():
    """`bench.py --gpus N` never times a different number of GPUs than asked for -- without a launcher it
    re-executes itself as N ranks (torch.distributed.run on 127.0.0.1), under a launcher the world size must equal the flag."""
    import bench
    assert bench.launch_decision(1, {}) == ("run", 1)
    assert bench.launch_decision(8, {}) == ("spawn", 8)
    assert bench.launch_decision(2, {"WORLD_SIZE": "2", "RANK": "1"}) == ("run", 2)
    assert bench.launch_decision(1, {"WORLD_SIZE": "1"}) == ("run", 1)
    for gpus, env in ((8, {"WORLD_SIZE": "2"}), (1, {"WORLD_SIZE": "4"}), (0, {})):
        what, msg = bench.launch_decision(gpus, env)
        assert what == "refuse" and isinstance(msg, str)


def test_bench_spawn_builds_the_drivers_command(monkeypatch):
    """The command `bench.py --gpus 2` re-executes is the driver's own launch line (one rank per GPU, rendezvous on 127.0.0.1);
    checked up to the launch itself (no GPU here)."""
    import subprocess
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0
    monkeypatch.setattr(subprocess, "call", fake_call)
    assert bench.spawn_ranks(2, ["--gpus", "2", "--steps", "3", "--warmup", "1"]) == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=2" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-6:] == ["--gpus", "2", "--steps", "3", "--warmup", "1"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_stat_board_feeds_and_world2_reduction_shapes():
    """reftr_amd.util.misc.StatBoard: host rows and device vectors give the same global averages the reference's meters would
    (mean over iterations), tensors are read with one stacked copy, windows hold the last values."""
    import torch
    from reftr_amd.util.misc import StatBoard
    b = StatBoard(window=3)
    for i in range(5):
        b.add(loss=float(i), lr=0.1, grad_norm=torch.tensor([2.0 * i]))
        b.add_device(("a", "b_unscaled"), torch.tensor([1.0 * i, 10.0 * i]))
    avg = b.global_avg()
    assert avg["loss"] == 2.0 and abs(avg["lr"] - 0.1) < 1e-12 and avg["grad_norm"] == 4.0
    assert avg["a"] == 2.0 and avg["b_unscaled"] == 20.0
    assert list(b.recent["loss"]) == [2.0, 3.0, 4.0] and b.last["a"] == 4.0
    assert "loss: 4" in str(b)
    b.synchronize_between_processes()          # no process group: a no-op
    assert b.global_avg()["loss"] == 2.0


def test_cooperative_decoder_gating_and_failure_flag_on_the_host():
    """Off the GPU the cooperative decoder launches are never chosen (no compute units to keep 80 workgroups resident), and the
    epoch loops turn the launches' failure word into an error instead of returning numbers."""
    import types
    from reftr_amd.engine_vg import _check_cooperative
    from reftr_amd.models import layout as L
    from reftr_amd.models.reftr_transformer import RefTR
    model = RefTR(L.ModelConfig(enc_layers=1, dec_layers=1, bert=L.BertConfig(layers=1)), device="cpu")
    assert not model.net.dec_stack_coop_ok(8, 1, 440, 6, True)          # shape would qualify; the device does not
    assert model.net._dec_handoff is None
    _check_cooperative(model)                                           # no launch has happened: nothing to report
    fake = types.SimpleNamespace(net=types.SimpleNamespace(dec_counters=torch.tensor([7, 0])))
    _check_cooperative(fake)
    fake.net.dec_counters = torch.tensor([7, 1])
    with pytest.raises(RuntimeError, match="hand-off timed out"):
        _check_cooperative(fake)
    wrapped = types.SimpleNamespace(module=fake)                        # a DistributedDataParallel-style wrapper
    with pytest.raises(RuntimeError):
        _check_cooperative(wrapped)


def test_cover_span_tiles_a_span_exactly_once():
    """The optimizer's matrix jobs + complement chunks (reftr_amd.optim.cover_span -> rt_adamw_mat / rt_adamw_chunks) cover every
    element of a span once: no parameter is skipped or updated twice."""
    import random
    from reftr_amd.optim import cover_span
    rnd = random.Random(0)
    for _ in range(50):
        b = rnd.randrange(0, 8) * 4096
        pos, mats = b, []
        for _ in range(rnd.randrange(0, 7)):
            pos += rnd.randrange(0, 6000) * 4
            N, T, C = rnd.choice([1, 4, 64, 70, 256]), rnd.choice([1, 9]), rnd.choice([4, 12, 64, 768])
            mats.append((pos, N, T, C)); pos += N * T * C
        e = pos + rnd.randrange(0, 9000) * 4
        jobs, tiles, chunks = cover_span(mats, b, e)
        seen = torch.zeros(e - b, dtype=torch.int32)
        for off, N, T, C, first in jobs:
            seen[off - b: off - b + N * T * C] += 1
        for off, cnt in zip(chunks[0::2], chunks[1::2]):
            assert 0 < cnt <= 16384 and off % 4 == 0 and cnt % 4 == 0
            seen[off - b: off - b + cnt] += 1
        assert bool((seen == 1).all())
        assert tiles == sum(((N + 31) // 32) * ((T * C + 255) // 256) for _, N, T, C in mats)
        firsts = [j[4] for j in jobs]
        assert firsts == sorted(firsts) and (not firsts or firsts[0] == 0)
    with pytest.raises(AssertionError):
        cover_span([(0, 64, 1, 64), (1000, 64, 1, 64)], 0, 1 << 20)        # overlapping matrices


def test_replayed_iteration_with_a_raised_failure_word_is_run_again_not_returned():
    """_ReplayInFlight.finish(): the cooperative launches' failure word sits behind the losses in the stats vector; when it is
    raised the handle hands back the result of the retry closure (the iteration on the launched chain) instead of the numbers of the
    void iteration, and a handle without one raises."""
    import types
    from reftr_amd import engine_vg as E

    class Ev:
        def synchronize(self):
            pass
    crit = types.SimpleNamespace(weight_dict={"loss_bbox": 5.0, "loss_giou": 2.0})
    # (round 5: every handle reads ITS OWN (pinned buffer, event) pair -- two iterations may be in flight)
    cap = types.SimpleNamespace(stat_names=("loss_bbox", "loss_giou"), fail_word=object(), model=None)
    slot = (torch.tensor([0.5, 0.25, 0.0, 3.0]), Ev())
    loss, scaled, unscaled, gn = E._ReplayInFlight(cap, crit, retry=lambda: "again", slot=slot).finish()
    assert abs(loss - 3.0) < 1e-6 and gn == 3.0 and scaled["loss_bbox"] == 2.5 and unscaled["loss_giou_unscaled"] == 0.25
    slot2 = (torch.tensor([0.5, 0.25, 2.0, 3.0]), Ev())                        # two ranks reported a timeout
    first = E._ReplayInFlight(cap, crit, retry=lambda: "again", slot=slot)
    assert E._ReplayInFlight(cap, crit, retry=lambda: "again", slot=slot2).finish() == "again"
    assert first.finish()[3] == 3.0                                            # the older handle still reads its own buffer
    cap_nofail = types.SimpleNamespace(stat_names=("loss_bbox", "loss_giou"), fail_word=None, model=None)
    assert E._ReplayInFlight(cap_nofail, crit, slot=(torch.tensor([0.5, 0.25, 3.0]), Ev())).finish()[3] == 3.0   # [losses | norm]


def test_data_parallel_boundary_clears_never_written_matrices_before_the_exchange(monkeypatch):
    """ParamStore.finish_overwrite_range: a registered matrix that this step never writes must be cleared
    BEFORE its slice is exchanged at a data-parallel boundary, and the end-of-backward clear must leave it (and the in-flight
    exchange buffer) alone afterwards.  rt_zero_chunks is replaced by a torch stand-in (host logic under test, no GPU here)."""
    from reftr_amd import hip as H
    from reftr_amd.models import layout as L
    from reftr_amd.models.store import ParamStore
    log = []

    def zero_chunks(base, table, n):
        t = table.view(-1, 2).tolist()
        log.append([tuple(r) for r in t[:n]])
        for off, cnt in t[:n]:
            base[off:off + cnt] = 0
    monkeypatch.setattr(H, "zero_chunks", zero_chunks)
    st = ParamStore(L.ModelConfig(enc_layers=1, dec_layers=1, bert=L.BertConfig(layers=1)), "cpu")
    a = st.G["vl_transformer.encoder.layers.0.linear1.weight"]
    b = st.G["vl_transformer.encoder.layers.0.linear2.weight"]
    c = st.G["lang_backbone.encoder.layer.0.output.dense.weight"]
    for t in (a, b, c):
        st.register_overwritable(t)
    st.flat_g.fill_(3.0)                                   # "the previous step's gradients"
    st.arm_overwrite()
    assert st.claim(a) and not st.claim(a)                 # a is produced (overwritten) in this step; b and c are not
    a.fill_(1.0)
    ma, mb = st.group_range[L.GROUP_MAIN]
    log.clear()
    st.finish_overwrite_range([(ma, mb // 2), (mb // 2, mb)])        # the main slice is final: split in two pieces like the exchange
    assert float(b.abs().sum()) == 0 and float(a.min()) == 1.0 and float(c.min()) == 3.0     # b cleared now, c is not in this slice
    assert len(log) == 1
    exchanged = st.flat_g[ma:mb].clone()                   # what the all-reduce would read
    b.fill_(9.0)                                            # stands for the in-place all-reduce result landing in the buffer
    st.finish_overwrite()                                   # end of backward: only c is left to clear
    assert float(b.min()) == 9.0 and float(c.abs().sum()) == 0
    assert float(exchanged.view(-1)[(st.offset["vl_transformer.encoder.layers.0.linear2.weight"][1] - ma)]) == 0
    offs = {o for o, _ in st.last_stale}
    assert offs == {st.offset["vl_transformer.encoder.layers.0.linear2.weight"][1],
                    st.offset["lang_backbone.encoder.layer.0.output.dense.weight"][1]}
    # not armed (plain zero_grad): nothing to do
    log.clear(); st.finish_overwrite_range([(ma, mb)]); assert not log


def test_evaluate_visualize_dumps(tmp_path):
    """evaluate(visualize=True)'s image dumps (reference engine_vg.py:86-96,157-192): directory layout, file names, the two-colour
    mask images, the boxes drawn on the image and the four attention maps at half the image size."""
    import numpy as np
    from PIL import Image
    from reftr_amd.engine_vg import _vis_dirs, _dump_visuals

    class DS:
        split = "testA"
        def pull_item(self, idx):
            img = np.full((40, 60, 3), 200, dtype=np.uint8)
            mask = np.zeros((40, 60), dtype=np.uint8); mask[10:30, 20:50] = 1
            return img, mask, "the left one", np.array([20., 10., 50., 30.]), "images/coco/COCO_train2014_000000000009.jpg"

    root = _vis_dirs(tmp_path, DS.split)
    assert root == tmp_path / "vis" / "testA" and all((root / d).is_dir() for d in ("mask", "bbox", "att", "gt"))
    pred = torch.zeros(40, 60, dtype=torch.uint8); pred[12:28, 22:48] = 1
    att = torch.rand(8, 10, 10)
    _dump_visuals(root, DS(), 7, pred, torch.tensor([28., 16., 44., 26.]), att)
    tag = "COCO_train2014_000000000009_00007"
    m = np.asarray(Image.open(root / "mask" / f"{tag}.jpg")).astype(int)
    g = np.asarray(Image.open(root / "gt" / f"{tag}.jpg")).astype(int)
    assert m.shape == (40, 60, 3) and g.shape == (40, 60, 3)
    # JPEG is lossy: interior pixels are yellow-ish (255, 255, 0), far background purple-ish (128, 0, 128)
    assert np.abs(m[20, 35] - np.array([255, 255, 0])).max() < 40 and np.abs(m[2, 2] - np.array([128, 0, 128])).max() < 40
    assert np.abs(g[20, 35] - np.array([255, 255, 0])).max() < 40 and np.abs(g[35, 5] - np.array([128, 0, 128])).max() < 40
    b = np.asarray(Image.open(root / "bbox" / f"{tag}.jpg")).astype(int)
    assert b.shape == (40, 60, 3)
    assert b[20, 21, 0] > 150 and b[20, 21, 2] < 120          # the target box's left edge (red, 5 px wide from x = 20)
    assert b[20, 30, 2] > 150 and b[20, 30, 0] < 120          # the predicted box's left edge (blue, from x = 28)
    for head in (0, 1, 2, 7):
        a = np.asarray(Image.open(root / "att" / f"{tag}_{head}.jpg"))
        assert a.shape[:2] == (20, 30)
    with pytest.raises(ValueError):
        _vis_dirs(None, "val")


def test_box_weights_and_direct_loss_gate(monkeypatch):
    """Host side of the captured step's direct loss path: the weight tensor handed to rt_box_loss is weight_dict in the kernel's
    [layer][bbox, giou] order (last layer = the un-suffixed keys, criterion.py / engine_vg.py:43), and the path is taken only for
    box-loss-only totals of an unwrapped single-process model."""
    from types import SimpleNamespace
    from reftr_amd.engine_vg import CapturedTrainStep
    from reftr_amd.models.criterion import CriterionVGMultiPhrase, _box_weights
    wd = {"loss_bbox": 5.0, "loss_giou": 2.0, "loss_bbox_0": 0.5, "loss_giou_0": 0.25, "loss_bbox_1": 3.0, "loss_giou_1": 4.0}
    crit = CriterionVGMultiPhrase(wd, ["boxes"])
    w = _box_weights(crit, 3, torch.device("cpu"))
    assert w.tolist() == [[0.5, 0.25], [3.0, 4.0], [5.0, 2.0]]
    assert _box_weights(crit, 1, torch.device("cpu")).tolist() == [[5.0, 2.0]]           # no aux outputs: the last layer only
    losses = torch.tensor([[1.0, 2.0], [3.0, 4.0], [5.0, 6.0]])
    ld = crit._loss_dict(losses)
    assert float(ld["loss_bbox"]) == 5.0 and float(ld["loss_giou_1"]) == 4.0 and float(ld["loss_bbox_0"]) == 1.0
    total = crit.weighted_total(ld)
    assert abs(float(total) - sum(float(ld[k]) * wd[k] for k in ld)) < 1e-5

    def gate(**kw):
        inner = SimpleNamespace(seg=kw.get("seg"), dp_mode=kw.get("dp", False))
        cap = CapturedTrainStep.__new__(CapturedTrainStep)
        cap.inner = inner
        cap.model = inner if not kw.get("wrapped") else SimpleNamespace(module=inner)
        cap.criterion = kw.get("crit", crit)
        return cap._direct_loss_ok()
    monkeypatch.delenv("REFTR_LOSS_DIRECT", raising=False)
    monkeypatch.delenv("REFTR_FUSED_TOTAL", raising=False)
    assert gate() is True
    assert gate(wrapped=True) is False                       # a data-parallel wrapper drives backward itself
    assert gate(dp=True) is False
    assert gate(seg=object()) is False                       # REC+RES: the total has mask terms
    assert gate(crit=CriterionVGMultiPhrase(wd, ["boxes", "masks"])) is False
    monkeypatch.setenv("REFTR_LOSS_DIRECT", "0")
    assert gate() is False
