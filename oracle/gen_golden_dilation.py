"""Mint tests/golden/e2e_dilation.npz: the reference RefTR built with --dilation (models/modeling/backbone.py:117-125:
torchvision's replace_stride_with_dilation=[False, False, True] -- layer4 at stride 1, its 3x3 convolutions dilated by 2 from the
second block on, the c5 map at stride 16), single-phrase inputs, reduced depth.  Container-only, like oracle/gen_golden.py (the
reference is imported read-only)."""
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
    args = ref_args(enc_layers=2, dec_layers=2, dilation=True)
    model = build_ref_model(rt, bb, vl, args, bert_layers=2)
    with redirect_stdout(io.StringIO()):
        wd = {"loss_giou": 1.0, "loss_bbox": 1.0, "loss_giou_0": 1.0, "loss_bbox_0": 1.0}
        C = crit.CriterionVGMultiPhrase(wd, ["boxes"])
    fill_state_dict(model.state_dict())
    model.eval()
    samples, targets = make_inputs("e2e_dilation", B=2, H=96, W=128, L=12)
    out = model(ref_samples(misc, samples))
    losses = C(out, targets)
    total = sum(losses[k] * wd[k] for k in losses if k in wd)
    total.backward()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    grads = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
    cfg = O.Cfg(enc_layers=2, dec_layers=2, bert=O.BertCfg(layers=2), dilation=True)
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
    l4 = "img_backbone.0.body.layer4."
    keep = [l4 + "0.conv2.weight", l4 + "1.conv2.weight", l4 + "2.conv2.weight", l4 + "0.downsample.0.weight",
            "img_backbone.0.body.layer3.5.conv3.weight", "input_proj.0.0.weight", "bbox_embed.layers.2.weight"]
    np.savez_compressed(os.path.join(GOLD, "e2e_dilation.npz"), boxes=stack.detach().numpy(), total_loss=np.float32(float(total)),
                        c5_hw=np.array([96 // 16, 128 // 16]),
                        # the big convolution gradients as every 97th element (a few tens of KB each) + their norms
                        **{"grad." + k: (grads[k].reshape(-1)[::97] if grads[k].numel() > 100000 else grads[k]).numpy() for k in keep},
                        **{"gnorm." + k: np.float32(float(grads[k].norm())) for k in keep},
                        **{"loss." + k: np.float32(float(v)) for k, v in losses.items()})
    print("oracle vs imported reference (--dilation):", {k: "%.2e" % v for k, v in report.items()})
    assert all(v < 2e-4 for v in report.values()), report


if __name__ == "__main__":
    main()
