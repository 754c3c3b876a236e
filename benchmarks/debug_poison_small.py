"""Small-fixture version of debug_poison.py: NaN-fill (or big-value fill, POISON=big) every torch.empty on the GPU, run
the eager loop, compare with an unpoisoned run."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
_empty, _empty_like = torch.empty, torch.empty_like
MODE = os.environ.get("POISON", "nan")
ON = [False]
def poison(t):
    if ON[0] and t.is_cuda:
        if t.is_floating_point(): t.fill_(float("nan") if MODE == "nan" else 1e4)
        else: t.fill_(-1 if t.dtype != torch.uint8 else 255)
    return t
torch.empty = lambda *a, **k: poison(_empty(*a, **k))
torch.empty_like = lambda *a, **k: poison(_empty_like(*a, **k))
from test_model_gpu import build, to_cuda, rel, make_inputs
from reftr_amd.engine_vg import train_step
from reftr_amd.optim import FusedAdamW
samples, targets = make_inputs("e2e_single", B=2, H=96, W=128, L=12)
s, tg = to_cuda(samples, targets)
out = {}
for on in (False, True):
    model, crit, P, ocfg = build(small=True)
    model.eval()
    opt = FusedAdamW(model, lr=1e-4, lr_backbone=1e-5, weight_decay=1e-4)
    ON[0] = on
    r = []
    for it in range(2):
        lv, _, _, gn = train_step(model, crit, s, tg, opt, None, max_norm=0.1)
        torch.cuda.synchronize()
        r.append((lv, float(gn), model.store.flat_g.clone()))
    ON[0] = False
    out[on] = (r, model)
    print("poison" if on else "clean ", [(a, b) for a, b, _ in r])
(r0, m0), (r1, m1) = out[False], out[True]
for it in range(2):
    G0, G1 = r0[it][2], r1[it][2]
    st = m0.store
    bad = []
    for name, shape, kind in st.table:
        if kind != "param": continue
        a = st.view_of(G0, name); b = st.view_of(G1, name)
        if not torch.isfinite(b).all() or float((a - b).abs().max()) > 1e-5 * (float(a.abs().max()) + 1e-12) + 1e-9:
            bad.append((name, float((a - b).abs().max()) if torch.isfinite(b).all() else float("nan"), float(a.abs().max())))
    print("step", it, len(bad), "params differ"); [print("   ", x) for x in bad]
