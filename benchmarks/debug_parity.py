"""Stage-by-stage parity dump (GPU box): HIP model vs the CPU oracle on the reduced-depth fixture inputs.
    python benchmarks/debug_parity.py [--full] [--multi]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import reftr_oracle as O          # noqa: E402
from oracle.shapes import param_shapes        # noqa: E402
from oracle.synth import make_inputs          # noqa: E402
from oracle.weights import formula_state      # noqa: E402
from reftr_amd.models import layout as L      # noqa: E402
from reftr_amd.models.criterion import CriterionVGMultiPhrase  # noqa: E402
from reftr_amd.models.reftr_transformer import RefTR            # noqa: E402
from reftr_amd.util.misc import NestedTensor  # noqa: E402


def rel(a, b):
    a = a.detach().float().cpu(); b = b.detach().float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true")
    ap.add_argument("--multi", action="store_true")
    args = ap.parse_args()
    if args.full:
        ocfg, cfg = O.Cfg(), L.ModelConfig()
        B, Hh, Ww, Lq = 2, 320, 320, 40
    else:
        ocfg = O.Cfg(enc_layers=2, dec_layers=2, bert=O.BertCfg(layers=2))
        cfg = L.ModelConfig(enc_layers=2, dec_layers=2, bert=L.BertConfig(layers=2))
        B, Hh, Ww, Lq = 2, 96, 128, 12
    tag = "e2e_multi" if args.multi else "e2e_single"
    samples, targets = make_inputs(tag, B=B, H=Hh, W=Ww, L=Lq, n_phrase=3 if args.multi else 0)
    P = formula_state(param_shapes(ocfg))
    model = RefTR(cfg, device="cuda")
    model.load_state_dict(P, strict=True)
    model.eval()
    model._debug = True
    names = [k for k in P if O.is_trainable(k)]
    res = {}
    for q in (False, True):
        Pq = {k: v.clone() for k, v in P.items()}
        leaves = {k: Pq[k].requires_grad_(True) for k in names}
        o = O.reftr_forward(Pq, samples, ocfg, q=q)
        losses = O.criterion(o, targets)
        tot = O.total_loss(losses, O.weight_dict(ocfg))
        inter = [o["logits"], o["hs"], o["memory"], o["c5"]]
        allg = torch.autograd.grad(tot, [leaves[k] for k in names] + inter)
        grads = dict(zip(names, allg[:len(names)]))
        res[q] = (o, losses, tot, grads, allg[len(names):])
    s = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in samples.items() if k not in ("img", "img_mask")}
    s["img"] = NestedTensor(samples["img"].cuda(), samples["img_mask"].cuda())
    tg = [{k: v.cuda() for k, v in t.items()} for t in targets]
    crit = CriterionVGMultiPhrase(O.weight_dict(ocfg), ["boxes"])
    out = model(s)
    ld = crit(out, tg)
    total = sum(ld[k] * crit.weight_dict[k] for k in ld if k in crit.weight_dict)
    model.store.flat_g.zero_()
    total.backward()
    torch.cuda.synchronize()
    sv = model._saved
    for q in (False, True):
        o, losses, tot, grads, ig = res[q]
        print(f"==== vs oracle q={q}")
        Bn, C, h, w = o["c5"].shape
        print("c5      ", rel(sv["c5"].view(Bn, h, w, C).permute(0, 3, 1, 2), o["c5"]))
        print("memory  ", rel(sv["memory"].view(Bn, -1, 256).transpose(0, 1), o["memory"]))
        print("logits  ", rel(out["pred_logits"], o["logits"]))
        print("boxes   ", rel(out["pred_logits"].sigmoid(), o["logits"].sigmoid()))
        print("loss    ", float(total), float(tot), {k: (round(float(ld[k]), 5), round(float(losses[k]), 5)) for k in losses})
        errs = sorted(((rel(model.store.G[k], grads[k]), k, float(grads[k].norm())) for k in names), reverse=True)
        if hasattr(model, "_dbg"):
            d = model._dbg
            print("d_logits", rel(d["dlogits"], ig[0]))
            print("d_hs    ", rel(d["dhs"].view(ig[1].shape), ig[1]))
            print("d_memory", rel(d["dmem"].view(Bn, -1, 256).transpose(0, 1), ig[2]))
            print("d_c5    ", rel(d["g_c5"].view(Bn, h, w, C).permute(0, 3, 1, 2), ig[3] * (o["c5"] > 0)))
        print("best grads:")
        errs_a = sorted(((rel(model.store.G[k], grads[k]), k, float(grads[k].norm())) for k in names))
        for e, k, n in errs_a[:25]:
            print(f"   {e:9.3e}  |g|={n:9.3e}  {k}")
        print("worst grads:")
        for e, k, n in errs[:25]:
            print(f"   {e:9.3e}  |g|={n:9.3e}  {k}")
        tot_g = torch.cat([model.store.G[k].detach().float().cpu().reshape(-1) for k in names])
        ref_g = torch.cat([grads[k].reshape(-1) for k in names])
        print("global grad rel-L2", float((tot_g - ref_g).norm() / ref_g.norm()), " median", errs[len(errs) // 2][0])


if __name__ == "__main__":
    main()
