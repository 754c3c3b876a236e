"""Mint tests/golden/e2e_learned_pos.npz: the reference RefTR built with --position_embedding learned
(models/modeling/position_encoding.py:59-84,91-92: Joiner[1] = PositionEmbeddingLearned, two nn.Embedding(50, 128)), multi-phrase
inputs, reduced depth.  Container-only, like oracle/gen_golden.py (the reference is imported read-only)."""
import io
import os
import sys
from contextlib import redirect_stdout

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle.gen_golden import GOLD, build_ref_model, import_reference, ref_args, ref_samples, rel   # noqa: E402
from oracle.synth import make_inputs   # noqa: E402


def main():
    torch.manual_seed(0); torch.set_num_threads(8)
    rt, crit, bb, vl, pp, misc = import_reference()
    from oracle import reftr_oracle as O
    from oracle.shapes import param_shapes
    from oracle.weights import fill_state_dict
    args = ref_args(enc_layers=2, dec_layers=2, position_embedding="learned")
    model = build_ref_model(rt, bb, vl, args, bert_layers=2)
    with redirect_stdout(io.StringIO()):
        wd = {"loss_giou": 1.0, "loss_bbox": 1.0, "loss_giou_0": 1.0, "loss_bbox_0": 1.0}
        C = crit.CriterionVGMultiPhrase(wd, ["boxes"])
    fill_state_dict(model.state_dict())
    model.eval()
    samples, targets = make_inputs("e2e_learned", B=2, H=96, W=128, L=12, n_phrase=3)
    out = model(ref_samples(misc, samples))
    losses = C(out, targets)
    total = sum(losses[k] * wd[k] for k in losses if k in wd)
    total.backward()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    grads = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
    order = [n for n, p in model.named_parameters() if p.requires_grad]
    cfg = O.Cfg(enc_layers=2, dec_layers=2, bert=O.BertCfg(layers=2), pos_learned=True)
    shp = param_shapes(cfg)
    ref_shp = {k: tuple(v.shape) for k, v in sd.items() if torch.is_floating_point(v)}
    assert shp == ref_shp, (set(shp) ^ set(ref_shp), [k for k in shp if k in ref_shp and shp[k] != ref_shp[k]])
    P = {k: v.clone() for k, v in sd.items()}
    names = [k for k in P if O.is_trainable(k) and torch.is_floating_point(P[k])]
    assert sorted(names) == sorted(grads.keys()), set(names) ^ set(grads.keys())
    leaves = {k: P[k].requires_grad_(True) for k in names}
    o = O.reftr_forward(P, samples, cfg)
    ol = O.criterion(o, targets)
    og = torch.autograd.grad(O.total_loss(ol, O.weight_dict(cfg)), [leaves[k] for k in names])
    stack = torch.stack([a["pred_boxes"] for a in out["aux_outputs"]] + [out["pred_boxes"]])
    report = {"boxes": rel(o["logits"].sigmoid(), stack),
              "loss": max(abs(float(ol[k]) - float(losses[k])) for k in losses),
              "grads_worst": max(rel(g, grads[k]) for k, g in zip(names, og))}
    rk, ck = "img_backbone.1.row_embed.weight", "img_backbone.1.col_embed.weight"
    np.savez_compressed(os.path.join(GOLD, "e2e_learned_pos.npz"), boxes=stack.detach().numpy(), total_loss=np.float32(float(total)),
                        grad_row=grads[rk].numpy(), grad_col=grads[ck].numpy(), grad_bbox2_w=grads["bbox_embed.layers.2.weight"].numpy(),
                        grad_level_embed=grads["vl_transformer.level_embed"].numpy(), param_order=np.array(order),
                        **{"loss." + k: np.float32(float(v)) for k, v in losses.items()})
    print("oracle vs imported reference (learned position embedding):", {k: "%.2e" % v for k, v in report.items()})
    assert all(v < 2e-4 for v in report.values()), report


if __name__ == "__main__":
    main()
