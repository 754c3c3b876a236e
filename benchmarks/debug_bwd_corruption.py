"""Does a backward write anything it must not?  Forward + backward at configs[0] size, then compare every persistent buffer
(parameters, frozen weights, buffers, bf16 operands) with its copy from before the backward, and the logits of a second forward
with the first's."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import reftr_oracle as O  # noqa: E402
from oracle.shapes import param_shapes  # noqa: E402
from oracle.synth import make_inputs  # noqa: E402
from oracle.weights import formula_state, formula_tensor  # noqa: E402
from reftr_amd.models import layout as L  # noqa: E402
from reftr_amd.models.reftr_transformer import RefTR  # noqa: E402
from reftr_amd.util.misc import NestedTensor  # noqa: E402


def main():
    ocfg, cfg = O.Cfg(), L.ModelConfig()
    model = RefTR(cfg, device="cuda", aux_loss=True)
    model.load_state_dict(formula_state(param_shapes(ocfg)), strict=True)
    model.eval()
    samples, targets = make_inputs("e2e_single", B=2, H=320, W=320, L=40)
    s = {k: v.cuda() for k, v in samples.items() if k not in ("img", "img_mask")}
    s["img"] = NestedTensor(samples["img"].cuda(), samples["img_mask"].cuda())
    st = model.store
    out = model(s)
    first = out["pred_logits"].detach().clone()
    W = formula_tensor("functional.w", tuple(out["pred_logits"].shape), 1.0, bf16=False).cuda()
    snap = {k: v.clone() for k, v in st.flat.items()}
    ops = {k: (l.W.clone(), l.WT.clone()) for k, l in model.net.lins.items()}
    st.flat_g.zero_()
    (out["pred_logits"] * W).sum().backward()
    torch.cuda.synchronize()
    for k, v in st.flat.items():
        d = (v != snap[k]).nonzero().flatten()
        print(f"flat[{k}]: {d.numel()} elements changed by the backward", d[:8].tolist())
        if d.numel() and k == "p":
            offs = sorted((o, n) for n, (b, o) in st.offset.items() if b == "p")
            for idx in d[:5].tolist():
                name = [n for o, n in offs if o <= idx][-1]
                print("   ", idx, name)
    bad = [k for k, l in model.net.lins.items() if not (torch.equal(l.W, ops[k][0]) and torch.equal(l.WT, ops[k][1]))]
    print("bf16 operands changed:", bad[:10])
    dump = os.environ.get("GRAD_DUMP")
    if dump and not os.path.exists(dump):
        torch.save(st.flat_g.cpu(), dump)
    elif dump:
        ref = torch.load(dump).cuda()
        g = st.flat_g
        print("whole buffer: cos %.6f  rel diff %.3e  norms %.6e %.6e" % (float((g * ref).sum() / (g.norm() * ref.norm())),
              float((g - ref).norm() / ref.norm()), float(g.norm()), float(ref.norm())))
        rows = []
        for n, (b, o) in st.offset.items():
            if b != "p":
                continue
            a, r = st.G[n].reshape(-1).float(), st.view_of(ref, n).reshape(-1).float()
            d = float((a - r).norm())
            rows.append((d, d / (float(r.norm()) + 1e-30), n))
        for d, rel, n in sorted(rows, reverse=True)[:14]:
            print("   %-70s |diff| %.3e  rel %.3e" % (n, d, rel))
    with torch.no_grad():
        second = model(s)["pred_logits"]
    print("second forward equals the first:", torch.equal(first, second), float((first - second).abs().max()))


if __name__ == "__main__":
    main()
