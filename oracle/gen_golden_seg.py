"""Mint tests/golden/seg_single.npz from the REFERENCE's RefTRSeg (models/reftr_segmentation.py), imported read-only in
the build container (same recipe and shims as oracle/gen_golden.py), and pin oracle.seg_forward / loss_masks against it.

    python oracle/gen_golden_seg.py
"""
import io
import os
import sys
from contextlib import redirect_stdout

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle.gen_golden import GOLD, import_reference, ref_args, ref_samples, rel   # noqa: E402


def seg_targets(targets, B, H, W):
    """Per-image bool masks [1,h,w] of DIFFERENT sizes (the criterion zero-pads them to the batch maximum)."""
    out = []
    for b, t in enumerate(targets):
        h, w = (H, W) if b % 2 == 0 else ((H * 3) // 4, (W * 2) // 3)
        yy, xx = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
        cx, cy, bw, bh = [float(v) for v in t["boxes"][0]]
        m = (((xx + 0.5) / w - cx).abs() < bw / 2) & (((yy + 0.5) / h - cy).abs() < bh / 2) & (((xx + yy) % 7) != 0)
        d = dict(t); d["masks"] = m[None]
        out.append(d)
    return out


def main(dilation=False, tag="seg_single"):
    """dilation=True (python oracle/gen_golden_seg.py --dilation): the same fixture for --masks with --dilation (DC5: layer4 keeps stride 16,
    backbone.py:117-125 + reftr_segmentation.py:343-384) -> tests/golden/seg_dilation.npz."""
    torch.manual_seed(0); torch.set_num_threads(8)
    rt, crit, bb, vl, pp, misc = import_reference()
    import models.reftr_segmentation as seg
    from transformers import BertConfig, BertModel
    from oracle import reftr_oracle as O
    from oracle.shapes import param_shapes
    from oracle.synth import make_inputs
    from oracle.weights import fill_state_dict

    args = ref_args(enc_layers=2, dec_layers=2, masks=True, aux_loss=False, dilation=dilation)
    with redirect_stdout(io.StringIO()):
        model = seg.RefTRSeg(bb.build_backbone(args), BertModel(BertConfig(num_hidden_layers=2, attn_implementation="eager")),
                             vl.build_vl_transformer(args), num_feature_levels=1, num_queries_per_phrase=1)
        wd = {"loss_giou": 1.0, "loss_bbox": 1.0, "loss_dice": 1.0, "loss_mask": 1.0, "loss_cem": 1.0}
        C = seg.CriterionVGOnePhraseSeg(wd, losses=["masks", "boxes"])
    fill_state_dict(model.state_dict())
    model.eval()
    B, H, W = 2, 96, 128
    samples, targets = make_inputs("seg_single", B=B, H=H, W=W, L=12)
    targets = seg_targets(targets, B, H, W)
    out = model(ref_samples(misc, samples))
    losses = C(out, targets)
    total = sum(losses[k] * wd[k] for k in losses if k in wd)
    total.backward()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    grads = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}

    cfg = O.Cfg(enc_layers=2, dec_layers=2, bert=O.BertCfg(layers=2), masks=True, aux_loss=False, dilation=dilation)
    shp = param_shapes(cfg)
    ref_shp = {k: tuple(v.shape) for k, v in sd.items() if torch.is_floating_point(v)}
    assert shp == ref_shp, (set(shp) ^ set(ref_shp), [k for k in shp if k in ref_shp and shp[k] != ref_shp[k]])
    P = {k: v.clone() for k, v in sd.items()}
    names = [k for k in P if O.is_trainable(k) and torch.is_floating_point(P[k])]
    assert sorted(names) == sorted(grads.keys()), (set(names) ^ set(grads.keys()))
    leaves = {k: P[k].requires_grad_(True) for k in names}
    o = O.reftr_forward(P, samples, cfg, train=False, q=False)
    ol = O.criterion(o, targets)
    ot = O.total_loss(ol, O.weight_dict(cfg))
    og = torch.autograd.grad(ot, [leaves[k] for k in names])
    report = {
        "pred_masks": rel(o["pred_masks"], out["pred_masks"]), "mask_att": rel(o["mask_att"], out["mask_att"]),
        "pred_boxes": rel(o["pred_boxes"], out["pred_boxes"]),
        "losses": max(abs(float(ol[k]) - float(losses[k])) / max(abs(float(losses[k])), 1e-6) for k in losses),
        "grads_worst": max(rel(g, grads[k]) for k, g in zip(names, og)),
    }
    # post-processing of the masks (PostProcessSegm, reftr_segmentation.py:288-302): exact decisions
    sizes = torch.tensor([[t["masks"].shape[-2], t["masks"].shape[-1]] for t in targets])
    orig = torch.tensor([[150, 210], [77, 60]])
    res = seg.PostProcessSegm()([{} for _ in range(B)], out, orig, sizes)
    gkeys = ["mask_head.lay1.weight", "mask_head.lay5.weight", "mask_head.out_lay.weight", "mask_head.adapter1.weight",
             "mask_head.adapter3.weight", "mask_head.gn1.weight", "mask_head.gn5.bias", "bbox_attention.q_linear.weight",
             "bbox_attention.k_linear.weight", "bbox_embed.layers.2.weight", "img_backbone.0.body.layer3.5.conv3.weight",
             "img_backbone.0.body.layer2.0.conv1.weight", "vl_transformer.decoder.layers.1.linear2.weight"]
    fixture = {
        "pred_masks": out["pred_masks"].detach().numpy(), "mask_att": out["mask_att"].detach().numpy(),
        "pred_boxes": out["pred_boxes"].detach().numpy(), "total_loss": np.float32(float(total)),
        "grad_names": np.array(names), "grad_norms": np.array([float(grads[k].norm()) for k in names], dtype=np.float32),
        "post_sizes": sizes.numpy(), "post_orig": orig.numpy(),
    }
    for i, t in enumerate(targets):
        fixture[f"target_mask{i}"] = t["masks"].numpy()
        fixture[f"post_masks{i}"] = res[i]["masks"].numpy(); fixture[f"post_masks_origin{i}"] = res[i]["masks_origin"].numpy()
    for k in gkeys:
        g = grads[k]
        fixture["grad." + k] = (g[:8] if g.dim() > 1 and g.shape[0] > 8 else g).numpy()
    fixture.update({"loss." + k: np.float32(float(v)) for k, v in losses.items()})
    np.savez_compressed(os.path.join(GOLD, tag + ".npz"), **fixture)
    print("oracle vs imported reference RefTRSeg (rel. error):")
    for k, v in report.items():
        print(f"  {k:14s} {v:.3e}")
    assert all(v < 2e-4 for v in report.values()), report
    print("written", os.path.join(GOLD, tag + ".npz"), {k: float(v) for k, v in losses.items()})


if __name__ == "__main__":
    if "--dilation" in sys.argv:
        main(dilation=True, tag="seg_dilation")
    else:
        main()
