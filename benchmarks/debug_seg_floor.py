"""Where does the RES head leave the oracle's order floor?  Every intermediate tensor of MaskHeadSmallConv (reftr_segmentation.py:240-280):
HIP path vs the q=True oracle, beside the q=True oracle in another summation order vs itself (the floor), at 320 x 320, B = 2."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from oracle import reftr_oracle as O
from oracle.synth import make_inputs
from test_parity_fullsize_gpu import build_full
from test_model_gpu import rel, to_cuda

size = int(os.environ.get("SIZE", "320"))
samples, targets = make_inputs("seg_full", B=2, H=size, W=size, L=40)
model, crit, P, ocfg = build_full(masks=True)
s, tg = to_cuda(samples, targets)
with torch.no_grad():
    out = model(s)
    sv = model._saved["seg"]
    O.MASK_HEAD_TRACE = t0 = {}
    o = O.reftr_forward(P, samples, ocfg, q=True)
    O.MASK_HEAD_TRACE = t1 = {}
    with O.accumulate_permuted(3):
        o1 = O.reftr_forward(P, samples, ocfg, q=True)
    O.MASK_HEAD_TRACE = None
B = 2
cv = model.seg.convs


def nchw(t, hh, ww, c):            # HIP [B*HW, ld] (channels padded) -> [B, c, h, w]
    return t.view(B, hh, ww, -1)[..., :c].permute(0, 3, 1, 2).float().cpu()


h, w = sv["h"], sv["w"]
rows = [("X0 (bf16 operand)", nchw(sv["X0"], h, w, cv["lay1"].cin), "X0", True),
        ("u1 = lay1(X0)", nchw(sv["u1"], h, w, cv["lay1"].cout), "u1", False),
        ("a1 = relu(gn1)", nchw(sv["a1"], h, w, cv["lay1"].cout), "a1", True),
        ("u2 = lay2(a1)", nchw(sv["u2"], h, w, cv["lay2"].cout), "u2", False)]
for j, stg in enumerate(sv["stages"]):
    lay = cv[f"lay{j + 3}"]
    rows.append((f"x{j} = adapter + up(a) (bf16 operand)", nchw(stg["x"], stg["fh"], stg["fw"], lay.cin), f"x{j}", True))
    rows.append((f"u{j + 3} = lay{j + 3}(x{j})", nchw(stg["u"], stg["fh"], stg["fw"], lay.cout), f"u{j + 3}", False))
st5 = sv["stages"][-1]
rows.append(("a5 = relu(gn5)", nchw(sv["a5"], st5["fh"], st5["fw"], cv["lay5"].cout), "a5", True))
print("%-40s %12s %12s %8s" % ("tensor", "HIP vs q", "floor", "ratio"))
for name, mine, key, is16 in rows:
    ref, alt = t0[key], t1[key]
    if is16:
        ref, alt = ref.to(torch.bfloat16).float(), alt.to(torch.bfloat16).float()
    a, b = rel(mine, ref), rel(alt, ref)
    print("%-40s %12.3e %12.3e %8.2f" % (name, a, b, a / max(b, 1e-30)))
a, b = rel(out["pred_masks"], o["pred_masks"]), rel(o1["pred_masks"], o["pred_masks"])
print("%-40s %12.3e %12.3e %8.2f" % ("pred_masks", a, b, a / b))
a, b = rel(out["mask_att"], o["mask_att"]), rel(o1["mask_att"], o["mask_att"])
print("%-40s %12.3e %12.3e %8.2f" % ("mask_att", a, b, a / b))
feats = model._saved["seg"]["stages"]
hip_f = {2 - j: st["f16"] for j, st in enumerate(feats)}          # stages use (layer3, layer2, layer1) outputs
for li in (0, 1, 2):
    ref, alt = o["feats"][li], o1["feats"][li]
    Bn, C, fh, fw = ref.shape
    mine = hip_f[li].view(Bn, fh, fw, C).permute(0, 3, 1, 2).float().cpu()
    a, b = rel(mine, ref), rel(alt, ref)
    print("%-40s %12.3e %12.3e %8.2f" % (f"layer{li + 1} output", a, b, a / b))
for j in range(3):
    ref, alt = t0[f"fo{j}"], t1[f"fo{j}"]
    print("%-40s %12s %12.3e" % (f"oracle adapter{j + 1} output floor", "", rel(alt, ref)))
