import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import reftr_oracle as O
from oracle.shapes import param_shapes
from oracle.synth import make_inputs
from oracle.weights import formula_state
from reftr_amd.models import layout as L
from reftr_amd.models.criterion import CriterionVGMultiPhrase
from reftr_amd.models.reftr_transformer import RefTR
from reftr_amd.optim import FusedAdamW
from reftr_amd.engine_vg import train_step
from reftr_amd.util.misc import NestedTensor

def rel(a, b):
    a = a.detach().float().cpu(); b = b.detach().float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))

ocfg = O.Cfg(enc_layers=2, dec_layers=2, bert=O.BertCfg(layers=2))
cfg = L.ModelConfig(enc_layers=2, dec_layers=2, bert=L.BertConfig(layers=2))
samples, targets = make_inputs("e2e_single", B=2, H=96, W=128, L=12)
P = formula_state(param_shapes(ocfg))
P0 = {k: v.clone() for k, v in P.items()}
state = {}
_, tot, gn, grads = O.train_step(P, samples, targets, ocfg, state, 1, max_norm=0.1, train=False)
model = RefTR(cfg, device="cuda"); model.load_state_dict(P0); model.eval()
crit = CriterionVGMultiPhrase(O.weight_dict(ocfg), ["boxes"])
opt = FusedAdamW(model, lr=1e-4, lr_backbone=1e-5, weight_decay=1e-4)
s = {k: v.cuda() for k, v in samples.items() if k not in ("img", "img_mask")}
s["img"] = NestedTensor(samples["img"].cuda(), samples["img_mask"].cuda())
tg = [{k: v.cuda() for k, v in t.items()} for t in targets]
lv, _, _, gnorm = train_step(model, crit, s, tg, opt, None, max_norm=0.1)
print("loss0", lv, tot, "gnorm", float(gnorm), gn)
sd = model.state_dict()
names = [k for k in P if O.is_trainable(k)]
rows = []
for k in names:
    d_mine = sd[k].cpu() - P0[k]; d_ref = P[k] - P0[k]
    lr = O.lr_group(k)
    rows.append((rel(d_mine, d_ref), k, float(d_mine.abs().mean()) / lr, float(d_ref.abs().mean()) / lr,
                 float((torch.sign(d_mine) == torch.sign(d_ref)).float().mean())))
rows.sort(reverse=True)
for r in rows[:12]: print("%.3f %-70s mine|d|/lr %.3f ref %.3f signagree %.3f" % r)
print("...")
for r in rows[-5:]: print("%.3f %-70s mine|d|/lr %.3f ref %.3f signagree %.3f" % r)
# loss of the oracle evaluated at MY updated parameters
with torch.no_grad():
    Pm = {k: sd[k].cpu() for k in P}
    o = O.reftr_forward(Pm, samples, ocfg)
    print("oracle loss at my params", float(O.total_loss(O.criterion(o, targets), O.weight_dict(ocfg))))
    o = O.reftr_forward(P, samples, ocfg)
    print("oracle loss at ref params", float(O.total_loss(O.criterion(o, targets), O.weight_dict(ocfg))))
out = model(s)
ld = crit(out, tg)
print("my loss at my params", float(sum(ld[k] * crit.weight_dict[k] for k in ld)))
m2 = RefTR(cfg, device="cuda"); m2.load_state_dict(sd); m2.eval()
ld2 = crit(m2(s), tg)
print("fresh model at my params", float(sum(ld2[k] * crit.weight_dict[k] for k in ld2)))
# which operands differ between stepped model and fresh model?
bad = []
for k, l in model.net.lins.items():
    e = rel(l.W, m2.net.lins[k].W)
    if e > 0: bad.append((e, k))
for k, w in model.body.W.items():
    e = rel(w, m2.body.W[k])
    if e > 0: bad.append((e, "conv " + k))
print("stale operands:", sorted(bad, reverse=True)[:10])
with torch.no_grad():
    for q in (False, True):
        o = O.reftr_forward(Pm, samples, ocfg, q=q)
        print("oracle q=%s at my params" % q, float(O.total_loss(O.criterion(o, targets), O.weight_dict(ocfg))),
              "logits rel", rel(m2(s)["pred_logits"], o["logits"]))
