"""Step-1 of the eager loop on a fresh model (small fixture): dump forward intermediates + gradients to a file, or compare two
dumps (modes that differ between PROCESSES point at reads of never-written device memory)."""
import os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
if sys.argv[1] == "cmp":
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    print("gnorm", a["gn"], b["gn"])
    n = 0
    for (pa, ta), (pb, tb) in zip(a["fwd"], b["fwd"]):
        if ta.shape != tb.shape or not torch.equal(ta, tb):
            print("  FWD DIFF", pa, tuple(ta.shape), "max|d| %.3e" % float((ta.float() - tb.float()).abs().max())); n += 1
            if n > 12: break
    bad = []
    for k in a["g"]:
        x, y = a["g"][k], b["g"][k]
        d = float((x - y).norm() / (y.norm() + 1e-30))
        if d > 1e-5: bad.append((round(d, 5), k))
    print(len(bad), "grad tensors differ:"); [print("   ", x) for x in sorted(bad, reverse=True)[:60]]
    sys.exit(0)
from test_model_gpu import build, to_cuda, rel, make_inputs
from reftr_amd.engine_vg import train_step
from reftr_amd.optim import FusedAdamW
if os.environ.get("DIRTY"):                       # leave garbage in device memory for the next process
    x = torch.full((int(os.environ["DIRTY"]) << 18,), 3.0e4, device="cuda"); torch.cuda.synchronize(); del x; sys.exit(0)
samples, targets = make_inputs("e2e_single", B=2, H=96, W=128, L=12)
s, tg = to_cuda(samples, targets)
def walk(o, path, out):
    if torch.is_tensor(o): out.append((path, o.detach().cpu().clone()))
    elif isinstance(o, dict):
        for k, v in o.items(): walk(v, f"{path}.{k}", out)
    elif isinstance(o, (list, tuple)):
        for i, v in enumerate(o): walk(v, f"{path}[{i}]", out)
model, crit, P, ocfg = build(small=True); model.eval()
opt = FusedAdamW(model, lr=1e-4, lr_backbone=1e-5, weight_decay=1e-4)
_, _, _, gn = train_step(model, crit, s, tg, opt, None, max_norm=0.1)
torch.cuda.synchronize()
snap = []; walk(model._saved, "saved", snap)
st = model.store
torch.save({"gn": float(gn), "fwd": snap, "g": {n: st.view_of(st.flat_g, n).detach().cpu().clone() for n, sh, k in st.table if k == "param"}}, sys.argv[1])
print("gnorm %.5f" % float(gn))
