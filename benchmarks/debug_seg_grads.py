"""Where does the +11 % on the bbox_attention q/k weight gradients come from?  Compare dP (gradient w.r.t. the attention
map) and the softmax-backward output between the HIP path and the oracle (fp32 and bf16-point)."""
import os, sys, numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from test_seg_gpu import build_seg, seg_batch, to_cuda, rel, GOLD, O
from reftr_amd import hip as H
g = np.load(os.path.join(GOLD, "seg_single.npz"))
model, crit, P, ocfg = build_seg(); model.eval()
samples, targets = seg_batch(g)
s, tg = to_cuda(samples, targets)
cap = {}
_bwd = H.attn_map_bwd
def bwd(q, k, Pm, dX0, B, HW, E, nh, *a):
    dq, dk = _bwd(q, k, Pm, dX0, B, HW, E, nh, *a)
    cap.update(P=Pm.clone(), dp=dX0[:, 2 * E:2 * E + nh].clone().view(B, HW, nh), dq=dq.clone(), dk=dk.clone(), q=q.clone())
    return dq, dk
H.attn_map_bwd = bwd
import reftr_amd.models.segmentation as SG
SG.H = H
out = model(s); losses = crit(out, tg); wd = crit.weight_dict
total = sum(losses[k] * wd[k] for k in losses if k in wd)
model.store.flat_g.zero_(); total.backward(); torch.cuda.synchronize()
for qmode in (False, True):
    box = {}
    _mh = O.mask_head
    def mh(Pp, x, bbox_mask, fpns, **kw):
        bbox_mask.retain_grad(); box["bm"] = bbox_mask
        return _mh(Pp, x, bbox_mask, fpns, **kw)
    O.mask_head = mh
    names = ["bbox_attention.q_linear.weight", "bbox_attention.k_linear.weight"]
    leaves = {k: P[k].clone().requires_grad_(True) for k in names}
    Pl = dict(P); Pl.update(leaves)
    o = O.reftr_forward(Pl, samples, ocfg, train=False, q=qmode)
    tot = O.total_loss(O.criterion(o, targets), O.weight_dict(ocfg))
    tot.backward()
    O.mask_head = _mh
    bm = box["bm"]; dbm = bm.grad                      # [B, Q=1, n, h, w]
    B, _, n, h, w = bm.shape
    dp_ref = dbm[:, 0].permute(0, 2, 3, 1).reshape(B, h * w, n)
    p_ref = bm[:, 0].permute(0, 2, 3, 1).reshape(B, h * w, n).detach()
    dp = cap["dp"].float().cpu(); Pm = cap["P"].float().cpu().view(B, n, h * w).permute(0, 2, 1)
    print("oracle q=%s:  P rel %.3e   dP rel %.4f ratio %.4f   mean(dP) hip %.4e ref %.4e   std hip %.4e ref %.4e" % (
        qmode, rel(Pm, p_ref), rel(dp, dp_ref), float(dp.norm() / dp_ref.norm()), float(dp.mean()), float(dp_ref.mean()),
        float(dp.std()), float(dp_ref.std())))
    # softmax backward recomputed on the CPU from each side's (P, dP)
    def sm_bwd(p, d):
        dot = (p * d).sum((1, 2), keepdim=True)
        return p * (d - dot)
    a, b = sm_bwd(Pm, dp), sm_bwd(p_ref, dp_ref)
    print("     dlogit (recomputed) rel %.4f ratio %.4f;   with hip dP + ref P: ratio %.4f;  with ref dP + hip P: ratio %.4f" % (
        rel(a, b), float(a.norm() / b.norm()), float(sm_bwd(p_ref, dp).norm() / b.norm()), float(sm_bwd(Pm, dp_ref).norm() / b.norm())))
    for k in names:
        print("     grad", k, "ratio %.4f" % float(model.store.G[k].detach().float().cpu().norm() / leaves[k].grad.norm()))
