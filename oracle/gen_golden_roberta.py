"""Mint tests/golden/e2e_roberta.npz: the reference RefTR with a HF RobertaModel language backbone
(configs/flickr30k/RefTR_flickr_roberta.sh:17, models/reftr_transformer.py:315-316), multi-phrase inputs."""
import io
import os
import sys
from contextlib import redirect_stdout

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle.gen_golden import GOLD, import_reference, ref_args, ref_samples, rel   # noqa: E402


from oracle.synth import roberta_inputs   # noqa: E402


def main():
    torch.manual_seed(0); torch.set_num_threads(8)
    rt, crit, bb, vl, pp, misc = import_reference()
    from transformers import RobertaConfig, RobertaModel
    from oracle import reftr_oracle as O
    from oracle.shapes import param_shapes
    from oracle.weights import fill_state_dict
    args = ref_args(enc_layers=2, dec_layers=2)
    with redirect_stdout(io.StringIO()):
        lm = RobertaModel(RobertaConfig(num_hidden_layers=2, vocab_size=50265, max_position_embeddings=514, type_vocab_size=1,
                                        layer_norm_eps=1e-5, pad_token_id=1, attn_implementation="eager"))
        model = rt.RefTR(bb.build_backbone(args), lm, vl.build_vl_transformer(args), num_feature_levels=1,
                         num_queries_per_phrase=1, aux_loss=True)
        wd = {"loss_giou": 1.0, "loss_bbox": 1.0, "loss_giou_0": 1.0, "loss_bbox_0": 1.0}
        C = crit.CriterionVGMultiPhrase(wd, ["boxes"])
    fill_state_dict(model.state_dict())
    model.eval()
    samples, targets = roberta_inputs()
    out = model(ref_samples(misc, samples))
    losses = C(out, targets)
    total = sum(losses[k] * wd[k] for k in losses if k in wd)
    total.backward()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    grads = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
    cfg = O.Cfg(enc_layers=2, dec_layers=2, bert=O.roberta_cfg(layers=2))
    shp = param_shapes(cfg)
    ref_shp = {k: tuple(v.shape) for k, v in sd.items() if torch.is_floating_point(v)}
    assert shp == ref_shp, (set(shp) ^ set(ref_shp), [k for k in shp if k in ref_shp and shp[k] != ref_shp[k]])
    P = {k: v.clone() for k, v in sd.items()}
    names = [k for k in P if O.is_trainable(k) and torch.is_floating_point(P[k])]
    leaves = {k: P[k].requires_grad_(True) for k in names}
    o = O.reftr_forward(P, samples, cfg)
    ol = O.criterion(o, targets)
    og = torch.autograd.grad(O.total_loss(ol, O.weight_dict(cfg)), [leaves[k] for k in names], allow_unused=True)
    stack = torch.stack([a["pred_boxes"] for a in out["aux_outputs"]] + [out["pred_boxes"]])
    report = {"boxes": rel(o["logits"].sigmoid(), stack),
              "loss": max(abs(float(ol[k]) - float(losses[k])) for k in losses),
              "grads_worst": max(rel(g, grads[k]) for k, g in zip(names, og) if g is not None and k in grads)}
    pk = "lang_backbone.embeddings.position_embeddings.weight"
    np.savez_compressed(os.path.join(GOLD, "e2e_roberta.npz"), boxes=stack.detach().numpy(), total_loss=np.float32(float(total)),
                        grad_pos_emb=grads[pk][:20].numpy(), grad_word_rows=grads["lang_backbone.embeddings.word_embeddings.weight"][[101, 102]].numpy(),
                        grad_bbox2_w=grads["bbox_embed.layers.2.weight"].numpy(),
                        **{"loss." + k: np.float32(float(v)) for k, v in losses.items()})
    print("oracle vs imported reference (RoBERTa backbone):", {k: "%.2e" % v for k, v in report.items()})
    assert all(v < 2e-4 for v in report.values()), report


if __name__ == "__main__":
    main()
