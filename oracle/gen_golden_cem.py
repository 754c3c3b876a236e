"""Mint tests/golden/seg_cem.npz from the REFERENCE's RefTRSeg built with cem_loss=True (--ablation cem_loss,
models/reftr_segmentation.py:16-41, 62-64, 146-147, 335-336), imported read-only in the build container, and pin
oracle.cem_forward against it.

    python oracle/gen_golden_cem.py
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
from oracle.gen_golden_seg import seg_targets                                       # noqa: E402


def main():
    torch.manual_seed(0); torch.set_num_threads(8)
    rt, crit, bb, vl, pp, misc = import_reference()
    import models.reftr_segmentation as seg
    from transformers import BertConfig, BertModel
    from oracle import reftr_oracle as O
    from oracle.shapes import param_shapes
    from oracle.synth import make_inputs
    from oracle.weights import fill_state_dict

    args = ref_args(enc_layers=2, dec_layers=2, masks=True, aux_loss=False)
    with redirect_stdout(io.StringIO()):
        model = seg.RefTRSeg(bb.build_backbone(args), BertModel(BertConfig(num_hidden_layers=2, attn_implementation="eager")),
                             vl.build_vl_transformer(args), num_feature_levels=1, num_queries_per_phrase=1, cem_loss=True)
        wd = {"loss_giou": 1.0, "loss_bbox": 1.0, "loss_dice": 1.0, "loss_mask": 1.0, "loss_cem": 1.0}
        C = seg.CriterionVGOnePhraseSeg(wd, losses=["masks", "boxes"])
    fill_state_dict(model.state_dict())
    model.eval()
    B, H, W = 2, 96, 128
    samples, targets = make_inputs("seg_single", B=B, H=H, W=W, L=12)
    targets = seg_targets(targets, B, H, W)
    out = model(ref_samples(misc, samples))
    losses = C(out, targets)
    assert "loss_cem" in losses
    total = sum(losses[k] * wd[k] for k in losses if k in wd)
    total.backward()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    grads = {n: (p.grad.detach().clone() if p.grad is not None else torch.zeros_like(p)) for n, p in model.named_parameters() if p.requires_grad}

    cfg = O.Cfg(enc_layers=2, dec_layers=2, bert=O.BertCfg(layers=2), masks=True, aux_loss=False, cem=True)
    shp = param_shapes(cfg)
    ref_shp = {k: tuple(v.shape) for k, v in sd.items() if torch.is_floating_point(v)}
    assert shp == ref_shp, (set(shp) ^ set(ref_shp), [k for k in shp if k in ref_shp and shp[k] != ref_shp[k]])
    P = {k: v.clone() for k, v in sd.items()}
    names = [k for k in P if O.is_trainable(k) and torch.is_floating_point(P[k])]
    leaves = {k: P[k].requires_grad_(True) for k in names}
    o = O.reftr_forward(P, samples, cfg, train=False, q=False)
    ol = O.criterion(o, targets)
    ot = O.total_loss(ol, O.weight_dict(cfg))
    og = torch.autograd.grad(ot, [leaves[k] for k in names], allow_unused=True)
    og = [g if g is not None else torch.zeros_like(P[k]) for g, k in zip(og, names)]
    big = [k for k in names if float(grads[k].norm()) > 1e-9]
    report = {
        "cem_loss": abs(float(ol["loss_cem"]) - float(losses["loss_cem"])) / abs(float(losses["loss_cem"])),
        "losses": max(abs(float(ol[k]) - float(losses[k])) / max(abs(float(losses[k])), 1e-6) for k in losses),
        "grads_worst": max(rel(g, grads[k]) for k, g in zip(names, og) if k in big),
    }
    # the two tensors with exactly-zero gradient: c1 (softmax over the single query is the constant 1) and c2.bias (softmax shift)
    print("zero-gradient tensors: |d c1.weight| max", float(grads["cem_block.c1.weight"].abs().max()), " |d c1.bias|", float(grads["cem_block.c1.bias"].abs().max()),
          " |d c2.bias|", float(grads["cem_block.c2.bias"].abs().max()), " |d c2.weight|", float(grads["cem_block.c2.weight"].norm()))
    gkeys = ["cem_block.c2.weight", "cem_block.c3.weight", "cem_block.c3.bias", "mask_head.lay5.weight", "mask_head.gn5.weight",
             "mask_head.out_lay.weight", "vl_transformer.decoder.layers.1.linear2.weight", "bbox_embed.layers.2.weight"]
    fixture = {"total_loss": np.float32(float(total)), "grad_names": np.array(names),
               "grad_norms": np.array([float(grads[k].norm()) for k in names], dtype=np.float32),
               "pred_masks": out["pred_masks"].detach().numpy(), "pred_boxes": out["pred_boxes"].detach().numpy()}
    for i, t in enumerate(targets):
        fixture[f"target_mask{i}"] = t["masks"].numpy()
    for k in gkeys:
        g = grads[k]
        fixture["grad." + k] = (g[:8] if g.dim() > 1 and g.shape[0] > 8 else g).numpy()
    fixture.update({"loss." + k: np.float32(float(v)) for k, v in losses.items()})
    np.savez_compressed(os.path.join(GOLD, "seg_cem.npz"), **fixture)
    print("oracle vs imported reference RefTRSeg + CEM (rel. error):")
    for k, v in report.items():
        print(f"  {k:14s} {v:.3e}")
    assert all(v < 2e-4 for v in report.values()), report
    print("written", os.path.join(GOLD, "seg_cem.npz"), {k: float(v) for k, v in losses.items()})


if __name__ == "__main__":
    main()
