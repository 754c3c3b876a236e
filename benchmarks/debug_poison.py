"""Find reads of uninitialised memory: every torch.empty / empty_like is filled with NaN (float) or 0xFF."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
_empty, _empty_like = torch.empty, torch.empty_like
def poison(t):
    if t.is_cuda:
        if t.is_floating_point(): t.fill_(float("nan"))
        else: t.fill_(-1 if t.dtype != torch.uint8 else 255)
    return t
torch.empty = lambda *a, **k: poison(_empty(*a, **k))
torch.empty_like = lambda *a, **k: poison(_empty_like(*a, **k))
from reftr_amd.engine_vg import train_step
from reftr_amd.models import layout as Lm
from reftr_amd.models.criterion import CriterionVGMultiPhrase
from reftr_amd.models.reftr_transformer import RefTR
from reftr_amd.optim import FusedAdamW
from reftr_amd.util.misc import NestedTensor
dev = torch.device("cuda")
cfg = Lm.ModelConfig()
model = RefTR(cfg, device=dev)
wd = {"loss_giou": 1.0, "loss_bbox": 1.0}
wd.update({f"{k}_{i}": v for i in range(5) for k, v in list(wd.items())})
crit = CriterionVGMultiPhrase(wd, ["boxes"])
torch.manual_seed(1234)
model.store.P["bbox_embed.layers.2.weight"].normal_(0, 0.02); model.mark_dirty()
opt = FusedAdamW(model)
model.train()
samples, targets = bench.synth_batch(8, 640, 640, 40, dev, 1234)
s = {k: v.to(dev) for k, v in samples.items() if k not in ("img", "img_mask")}
s["img"] = NestedTensor(samples["img"].to(dev), samples["img_mask"].to(dev))
tg = [{k: v.to(dev) for k, v in t.items()} for t in targets]
for i in range(3):
    r = train_step(model, crit, s, tg, opt, None, 0.1)
    G = model.store.G
    bad = [k for k in G if not torch.isfinite(G[k]).all()]
    print("step", i, "loss", r[0], "gnorm", float(opt.grad_norm), "nonfinite grads:", len(bad), bad[:12])
    sv = model._saved
    for k in ("c5", "mem32", "hs16", "y1", "y2"):
        print("   ", k, bool(torch.isfinite(sv[k].float()).all()))
    if bad: break
