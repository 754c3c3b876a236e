"""GPU end-to-end parity of the RES head (RefTRSeg, cfg4) through the C ABI against the golden vectors minted from the
imported reference (tests/golden/seg_single.npz; oracle/gen_golden_seg.py).

Tolerances (bf16 operands vs the reference's fp32): mask logits / attention map rel-L2 <= 2e-2, boxes <= 5e-3, the four
losses <= 1e-2 relative; gradients of the head tensors <= 0.1 rel-L2 and cosine >= 0.995 (measured: 1-6 %, the
FPN adapters sit behind three GroupNorm+ReLU stages whose masks flip under bf16 noise); global gradient (all
tensors) rel-L2 <= 0.25, cosine >= 0.97 (noise floor of ReLU-mask flips).  Post-processed masks are
compared as decisions (> 99 % of the pixels equal: logits within bf16 noise of 0 flip; measured 99.46 %)."""
import os

import numpy as np
import pytest
import torch

from oracle import reftr_oracle as O
from oracle.shapes import param_shapes
from oracle.synth import make_inputs
from oracle.weights import formula_state
from test_model_gpu import rel, to_cuda

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def build_seg(dilation=False):
    from reftr_amd.models import layout as L
    from reftr_amd.models.criterion import CriterionVGOnePhraseSeg
    from reftr_amd.models.reftr_transformer import RefTR
    ocfg = O.Cfg(enc_layers=2, dec_layers=2, bert=O.BertCfg(layers=2), masks=True, aux_loss=False, dilation=dilation)
    cfg = L.ModelConfig(enc_layers=2, dec_layers=2, bert=L.BertConfig(layers=2), masks=True, dilation=dilation)
    P = formula_state(param_shapes(ocfg))
    model = RefTR(cfg, device="cuda", aux_loss=False)
    model.load_state_dict(P, strict=True)
    crit = CriterionVGOnePhraseSeg(O.weight_dict(ocfg), ["masks", "boxes"])
    return model, crit, P, ocfg


def seg_batch(g):
    samples, targets = make_inputs("seg_single", B=2, H=96, W=128, L=12)
    targets = [dict(t, masks=torch.from_numpy(g[f"target_mask{i}"])) for i, t in enumerate(targets)]
    return samples, targets


def test_eval_forward_is_bit_reproducible(hip):
    """No atomics anywhere in the forward (GroupNorm statistics are reduced in a fixed order): boxes, mask logits and the
    attention map of two eval runs are bit-identical, like the reference's eval forward."""
    g = np.load(os.path.join(GOLD, "seg_single.npz"))
    model, crit, P, ocfg = build_seg()
    model.eval()
    s, tg = to_cuda(*seg_batch(g))
    with torch.no_grad():
        a = {k: v.clone() for k, v in model(s).items() if torch.is_tensor(v)}
        for _ in range(3):
            b = model(s)
            for k in a:
                assert torch.equal(a[k], b[k]), k


@pytest.mark.parametrize("fixture", ["seg_single", "seg_dilation"])
def test_refer_segmentation_vs_reference_golden(hip, fixture):
    """seg_dilation (round 6): --masks with --dilation -- layer4 at stride 16 (/root/reference/models/modeling/backbone.py:117-125), the mask
    head's first FPN level the same size as its input (/root/reference/models/reftr_segmentation.py:243-280, 343-384); golden vectors
    minted from the imported reference by `oracle/gen_golden_seg.py --dilation`."""
    g = np.load(os.path.join(GOLD, fixture + ".npz"))
    model, crit, P, ocfg = build_seg(dilation=fixture == "seg_dilation")
    model.eval()
    samples, targets = seg_batch(g)
    s, tg = to_cuda(samples, targets)
    out = model(s)
    assert out["pred_masks"].shape == g["pred_masks"].shape and out["mask_att"].shape == g["mask_att"].shape
    assert rel(out["pred_masks"], g["pred_masks"]) < 2e-2
    assert rel(out["mask_att"], g["mask_att"]) < 2e-2
    assert rel(out["pred_boxes"], g["pred_boxes"]) < 5e-3
    losses = crit(out, tg)
    for k in ("loss_mask", "loss_dice", "loss_bbox", "loss_giou"):
        ref = float(g["loss." + k])
        assert abs(float(losses[k]) - ref) < 1e-2 * max(abs(ref), 0.1), (k, float(losses[k]), ref)
    wd = crit.weight_dict
    total = sum(losses[k] * wd[k] for k in losses if k in wd)
    model.store.flat_g.zero_()
    total.backward()
    G = model.store.G
    for key in g.files:
        if not key.startswith("grad."):
            continue
        k = key[5:]
        ref = torch.from_numpy(g[key])
        mine = G[k].detach().float().cpu()
        mine = mine[:8] if mine.dim() > 1 and mine.shape[0] > 8 else mine
        if k.startswith("mask_head.") or k.startswith("bbox_attention."):
            cos = float((mine * ref).sum() / (mine.norm() * ref.norm() + 1e-30))
            # the attention-map projections see the mask loss only through dP, a 3e-6-sized gradient left over from
            # cancelling sums: the oracle's own bf16-point mode differs from its fp32 mode by rel 0.18 in dP and moves the
            # norm of these two weight gradients by +-15 % (benchmarks/debug_seg_grads.py); their direction is kept
            tol = 0.25 if k.startswith("bbox_attention.") else 0.1
            assert rel(mine, ref) < tol and cos > 0.995, (k, rel(mine, ref), cos)
    names = [str(n) for n in g["grad_names"]]
    gn_ref = torch.tensor(g["grad_norms"])
    gn = torch.tensor([float(G[k].norm()) for k in names])
    assert float((gn - gn_ref).norm() / gn_ref.norm()) < 0.1
    # padding of the padded storage never receives gradient
    ph = model.store.phys("mask_head.lay1.weight", grad=True)
    assert float(ph[:, :, 520:].abs().max()) == 0.0 and float(ph[520:].abs().max()) == 0.0
    # post-processing decisions
    from reftr_amd.models.post_process import PostProcessSegm
    res = PostProcessSegm()([{} for _ in range(2)], {"pred_masks": out["pred_masks"].detach()},
                            torch.from_numpy(g["post_orig"]), torch.from_numpy(g["post_sizes"]))
    for i in range(2):
        same = (res[i]["masks"].cpu() == torch.from_numpy(g[f"post_masks{i}"])).float().mean()
        assert float(same) > 0.99


@pytest.mark.parametrize("fixture", ["seg_single", "seg_dilation"])
def test_seg_gradients_vs_oracle_global(hip, fixture):
    g = np.load(os.path.join(GOLD, fixture + ".npz"))
    model, crit, P, ocfg = build_seg(dilation=fixture == "seg_dilation")
    model.eval()
    samples, targets = seg_batch(g)
    names = [k for k in P if O.is_trainable(k) and torch.is_floating_point(P[k])]
    leaves = {k: P[k].clone().requires_grad_(True) for k in names}
    Pl = dict(P); Pl.update(leaves)
    o = O.reftr_forward(Pl, samples, ocfg, train=False, q=False)
    tot = O.total_loss(O.criterion(o, targets), O.weight_dict(ocfg))
    ref = dict(zip(names, torch.autograd.grad(tot, [leaves[k] for k in names])))
    s, tg = to_cuda(samples, targets)
    out = model(s)
    ld = crit(out, tg)
    total = sum(ld[k] * crit.weight_dict[k] for k in ld if k in crit.weight_dict)
    model.store.flat_g.zero_()
    total.backward()
    a = torch.cat([model.store.G[k].detach().float().cpu().reshape(-1) for k in names])
    b = torch.cat([ref[k].reshape(-1) for k in names])
    assert float((a - b).norm() / b.norm()) < 0.25
    assert float((a * b).sum() / (a.norm() * b.norm())) > 0.97
    # the ResNet stages that only the RES head's FPN adapters reach more directly
    for k in ("img_backbone.0.body.layer3.5.conv3.weight", "img_backbone.0.body.layer2.3.conv3.weight"):
        x, y = model.store.G[k].detach().float().cpu().reshape(-1), ref[k].reshape(-1)
        assert float((x * y).sum() / (x.norm() * y.norm())) > 0.95, k


def test_cem_block_vs_reference_golden(hip):
    """--ablation cem_loss (CEM block, reftr_segmentation.py:16-41): loss_cem, its gradients onto c2 / c3 and through
    res_feat / the last decoder output, vs the imported reference (tests/golden/seg_cem.npz) and the fp32 oracle."""
    from reftr_amd.models import layout as L
    from reftr_amd.models.criterion import CriterionVGOnePhraseSeg
    from reftr_amd.models.reftr_transformer import RefTR
    g = np.load(os.path.join(GOLD, "seg_cem.npz"))
    ocfg = O.Cfg(enc_layers=2, dec_layers=2, bert=O.BertCfg(layers=2), masks=True, aux_loss=False, cem=True)
    cfg = L.ModelConfig(enc_layers=2, dec_layers=2, bert=L.BertConfig(layers=2), masks=True, cem=True)
    P = formula_state(param_shapes(ocfg))
    model = RefTR(cfg, device="cuda", aux_loss=False)
    model.load_state_dict(P, strict=True)
    # registered after mask_head (reftr_segmentation.py:62-64): the optimizer / checkpoint order of the reference
    assert L.reference_param_order(cfg)[-6:] == [f"cem_block.c{i}.{w}" for i in (1, 2, 3) for w in ("weight", "bias")]
    crit = CriterionVGOnePhraseSeg(O.weight_dict(ocfg), ["masks", "boxes"])
    model.eval()
    samples, targets = seg_batch(g)
    s, tg = to_cuda(samples, targets)
    out = model(s)
    losses = crit(out, tg)
    e = {k: abs(float(losses[k]) - float(g["loss." + k])) / max(abs(float(g["loss." + k])), 0.1)
         for k in ("loss_cem", "loss_mask", "loss_dice", "loss_bbox", "loss_giou")}
    print("cem losses rel err", e, "loss_cem", float(losses["loss_cem"]), float(g["loss.loss_cem"]))
    # measured: loss_cem 9.2e-4, loss_mask 1.6e-4, loss_dice 3.6e-5, loss_bbox 6.5e-4, loss_giou 2.6e-4 (asserted at 1.5x)
    assert e["loss_cem"] < 1.4e-3 and e["loss_mask"] < 2.5e-4 and e["loss_dice"] < 6e-5 and e["loss_bbox"] < 1e-3 and e["loss_giou"] < 4e-4
    total = crit.weighted_total(losses)
    ref_total = sum(losses[k] * crit.weight_dict[k] for k in losses if k in crit.weight_dict)
    assert abs(float(total) - float(ref_total)) < 1e-5 * abs(float(ref_total))
    assert abs(float(total) - float(g["total_loss"])) < 1e-2 * float(g["total_loss"])
    model.store.flat_g.zero_()
    total.backward()
    G = model.store.G
    worst = {}
    for key in g.files:
        if key.startswith("grad.cem_block.") or key in ("grad.mask_head.lay5.weight", "grad.mask_head.gn5.weight"):
            k = key[5:]
            ref = torch.from_numpy(g[key]).float()
            mine = G[k].detach().float().cpu()
            mine = mine[:8] if mine.dim() > 1 and mine.shape[0] > 8 else mine
            cos = float((mine * ref).sum() / (mine.norm() * ref.norm() + 1e-30))
            worst[k] = (rel(mine, ref), cos)
    print("cem grads (rel, cos)", worst)
    # measured (rel): c2.weight 2.9e-3, c3.weight 4.1e-3, c3.bias 1.6e-3, lay5.weight 9.1e-3, gn5.weight 2.8e-3; cos >= 0.99996
    tol = {"cem_block.c2.weight": 4.4e-3, "cem_block.c3.weight": 6.2e-3, "cem_block.c3.bias": 2.5e-3,
           "mask_head.lay5.weight": 1.4e-2, "mask_head.gn5.weight": 4.2e-3}
    for k, (r, c) in worst.items():
        assert r < tol[k] and c > 0.9999, (k, r, c)
    # exact zeros: c1 (softmax over the single query is the constant 1), c2.bias (softmax shift invariance)
    for k in ("cem_block.c1.weight", "cem_block.c1.bias", "cem_block.c2.bias"):
        assert float(G[k].abs().max()) == 0.0, k
    names = [str(n) for n in g["grad_names"]]
    gn_ref = torch.tensor(g["grad_norms"])
    gn = torch.tensor([float(G[k].norm()) for k in names])
    assert float((gn - gn_ref).norm() / gn_ref.norm()) < 0.1


def test_cem_training_steps_captured(hip):
    """The multitask step with loss_cem, eager and as hipGraph replays from the same state: finite, decreasing, and the two
    loss sequences agree to the trajectory noise of the fixture."""
    from reftr_amd.engine_vg import CapturedTrainStep, train_step
    from reftr_amd.models import layout as L
    from reftr_amd.models.criterion import CriterionVGOnePhraseSeg
    from reftr_amd.models.reftr_transformer import RefTR
    from reftr_amd.optim import FusedAdamW
    g = np.load(os.path.join(GOLD, "seg_cem.npz"))
    ocfg = O.Cfg(enc_layers=2, dec_layers=2, bert=O.BertCfg(layers=2), masks=True, aux_loss=False, cem=True)
    cfg = L.ModelConfig(enc_layers=2, dec_layers=2, bert=L.BertConfig(layers=2), masks=True, cem=True)
    P = formula_state(param_shapes(ocfg))
    s, tg = to_cuda(*seg_batch(g))
    seqs = []
    for captured in (False, True):
        model = RefTR(cfg, device="cuda", aux_loss=False)
        model.load_state_dict(P, strict=True)
        model.eval()                                     # no dropout: the two runs are comparable
        crit = CriterionVGOnePhraseSeg(O.weight_dict(ocfg), ["masks", "boxes"])
        opt = FusedAdamW(model, lr=1e-4, lr_backbone=1e-5)
        if captured:
            p0, m0, v0 = model.store.flat_p.clone(), opt.m.clone(), opt.v.clone()
            cap = CapturedTrainStep(model, crit, opt, 0.1, s, tg, warmup=1)
            cap.reset_pending()
            model.store.flat_p.copy_(p0); opt.m.copy_(m0); opt.v.copy_(v0); opt.step_dev.zero_(); opt.step_count = 0
            model.mark_dirty(full=True)
        ls = []
        for _ in range(4):
            if captured:
                l, ld, _ = cap(s, tg)
                ls.append((float(l), float(ld["loss_cem"])))
            else:
                lv, sc, _, _ = train_step(model, crit, s, tg, opt, None, max_norm=0.1)
                ls.append((lv, float(sc["loss_cem"])))
        if captured:
            cap.flush()
        seqs.append(ls)
    print("cem step losses", seqs)
    assert all(np.isfinite(v) for ls in seqs for t in ls for v in t)
    # step 1 is identical; afterwards the fixture amplifies 1-ulp weight differences (atomics order in the LayerNorm / bias
    # gradients): two EAGER runs differ from each other by up to 1e-3 at step 2, 2e-3 at step 3 and 3.6e-2 at step 4 (measured over
    # five runs), so the later steps are bounded like tests/test_model_gpu.py::test_captured_step_matches_eager bounds them
    for it, (a, b) in enumerate(zip(*seqs)):
        tol = (1e-6, 3e-3, 1e-2, 8e-2)[it]
        assert abs(a[0] - b[0]) < tol * abs(a[0]) and abs(a[1] - b[1]) < tol * abs(a[1]), (it, a, b)
    assert seqs[0][-1][0] < seqs[0][0][0]         # (step 1 is compared to 1e-6 above: the mask loss is summed with atomics)


def test_seg_training_steps_run(hip):
    """Three optimiser steps of the REC+RES multitask step (train mode, dropout on) stay finite and reduce the loss."""
    from reftr_amd.engine_vg import train_step
    from reftr_amd.optim import FusedAdamW
    g = np.load(os.path.join(GOLD, "seg_single.npz"))
    model, crit, P, ocfg = build_seg()
    model.train()
    opt = FusedAdamW(model, lr=1e-4, lr_backbone=1e-5, weight_decay=1e-4)
    samples, targets = seg_batch(g)
    s, tg = to_cuda(samples, targets)
    hist = [train_step(model, crit, s, tg, opt, None, max_norm=0.1)[0] for _ in range(4)]
    assert all(np.isfinite(h) for h in hist) and hist[-1] < hist[0], hist
    ph = model.store.phys("mask_head.lay1.weight")
    assert float(ph[:, :, 520:].abs().max()) == 0.0            # padded weights stay exactly zero through AdamW


def test_evaluate_rec_and_res_metrics(hip):
    """engine_vg.evaluate (engine_vg.py:82-225): Acc@0.5 / mIoU of the boxes and the mask IoU, on a two-batch loader,
    against the same metrics recomputed from the model outputs with plain torch ops (exact decisions)."""
    import torch.nn.functional as F
    from reftr_amd.engine_vg import evaluate
    from reftr_amd.models.post_process import PostProcessSegm, PostProcessVGMultiPhrase
    from reftr_amd.util import box_ops
    from reftr_amd.util.misc import NestedTensor
    g = np.load(os.path.join(GOLD, "seg_single.npz"))
    model, crit, P, ocfg = build_seg()
    samples, targets = seg_batch(g)
    for i, t in enumerate(targets):
        h, w = t["masks"].shape[-2:]
        t.update(size=torch.tensor([h, w]), orig_size=torch.tensor([2 * h + 1, 3 * w]), image_id=torch.tensor(10 + i),
                 dataset_id=torch.tensor(i))
    s = {k: v for k, v in samples.items() if k not in ("img", "img_mask")}
    s["img"] = NestedTensor(samples["img"], samples["img_mask"])
    loader = [(s, targets), (s, targets)]
    post = {"bbox": PostProcessVGMultiPhrase(), "segm": PostProcessSegm()}
    stats, results = evaluate(model, crit, post, loader, torch.device("cuda"))
    assert {"accuracy_iou0.5", "miou", "seg_miou", "loss", "loss_mask", "loss_dice", "loss_bbox", "loss_giou"} <= set(stats)
    assert not any(k.endswith("_unscaled") for k in stats)
    # independent recomputation
    model.eval()
    cs, ct = to_cuda(samples, targets)
    out = model(cs)
    ious, sious = [], []
    pm = F.interpolate(out["pred_masks"].squeeze(2), size=tuple(torch.stack([t["size"] for t in targets]).max(0)[0].tolist()),
                       mode="bilinear", align_corners=False).sigmoid() > 0.5
    for i, t in enumerate(ct):
        b = box_ops.box_cxcywh_to_xyxy(out["pred_boxes"][i, 0])
        ious.append(torch.diag(box_ops.box_iou(box_ops.box_cxcywh_to_xyxy(t["boxes"]), b)[0]))
        h, w = t["masks"].shape[-2:]
        sious.append(box_ops.mask_iou(pm[i, :, :h, :w][0], t["masks"][0]))
    iou = torch.cat(ious)
    assert abs(stats["accuracy_iou0.5"] - float((iou > 0.5).float().mean())) < 1e-6
    assert abs(stats["miou"] - float(iou.mean())) < 1e-5
    assert abs(stats["seg_miou"] - float(torch.stack(sious).mean())) < 1e-5
    assert set(results) == {10, 11} and len(results[10][0]) == 4
    sc = torch.tensor(results[10][0]); nb = box_ops.box_cxcywh_to_xyxy(out["pred_boxes"][0, 0]).cpu()[0]
    oh, ow = [float(v) for v in targets[0]["orig_size"]]
    assert torch.allclose(sc, nb * torch.tensor([ow, oh, ow, oh]), rtol=1e-4, atol=1e-3)


def test_evaluate_visualize_end_to_end(hip, tmp_path):
    """evaluate(..., visualize=True) (engine_vg.py:84-96,157-192) on the REC+RES model: one set of image dumps per evaluated sample
    under output_dir/vis/<split>/{mask,gt,bbox,att}, the predicted mask at the ORIGINAL image size (PostProcessSegm's
    'masks_origin'), metrics unchanged by the dumps."""
    from PIL import Image
    from reftr_amd.engine_vg import evaluate
    from reftr_amd.models.post_process import PostProcessSegm, PostProcessVGMultiPhrase
    from reftr_amd.util.misc import NestedTensor
    g = np.load(os.path.join(GOLD, "seg_single.npz"))
    model, crit, P, ocfg = build_seg()
    samples, targets = seg_batch(g)
    sizes = {}
    for i, t in enumerate(targets):
        h, w = t["masks"].shape[-2:]
        t.update(size=torch.tensor([h, w]), orig_size=torch.tensor([2 * h + 1, 3 * w]), image_id=torch.tensor(10 + i),
                 dataset_id=torch.tensor(i))
        sizes[i] = (2 * h + 1, 3 * w)

    class DS:
        split = "val"
        def pull_item(self, idx):
            H_, W_ = sizes[idx]
            m = np.zeros((H_, W_), dtype=np.uint8); m[H_ // 4: H_ // 2, W_ // 4: W_ // 2] = 1
            return (np.full((H_, W_, 3), 90, dtype=np.uint8), m, "a phrase", np.array([W_ / 4, H_ / 4, W_ / 2, H_ / 2]),
                    f"images/img_{idx:03d}.jpg")

    class Loader(list):
        dataset = DS()

    s = {k: v for k, v in samples.items() if k not in ("img", "img_mask")}
    s["img"] = NestedTensor(samples["img"], samples["img_mask"])
    post = {"bbox": PostProcessVGMultiPhrase(), "segm": PostProcessSegm()}
    plain, _ = evaluate(model, crit, post, Loader([(s, targets)]), torch.device("cuda"))
    stats, _ = evaluate(model, crit, post, Loader([(s, targets)]), torch.device("cuda"), output_dir=tmp_path, visualize=True)
    assert set(stats) == set(plain) and all(abs(stats[k] - plain[k]) <= 1e-5 * max(1.0, abs(plain[k])) for k in plain), (stats, plain)
    root = tmp_path / "vis" / "val"
    for i in range(len(targets)):
        tag = f"img_{i:03d}_{i:05d}"
        for sub, name in (("mask", f"{tag}.jpg"), ("gt", f"{tag}.jpg"), ("bbox", f"{tag}.jpg"), ("att", f"{tag}_0.jpg"), ("att", f"{tag}_7.jpg")):
            assert (root / sub / name).is_file(), (sub, name)
        assert Image.open(root / "mask" / f"{tag}.jpg").size == (sizes[i][1], sizes[i][0])
        assert Image.open(root / "att" / f"{tag}_1.jpg").size == (min(320, sizes[i][1] // 2), min(320, sizes[i][0] // 2))
