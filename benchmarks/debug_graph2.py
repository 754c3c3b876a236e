import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from reftr_amd.engine_vg import train_step, CapturedTrainStep
from reftr_amd.models import layout as Lm
from reftr_amd.models.criterion import CriterionVGMultiPhrase
from reftr_amd.models.reftr_transformer import RefTR
from reftr_amd.optim import FusedAdamW
from reftr_amd.util.misc import NestedTensor
dev = torch.device("cuda")
junk = torch.full((1 << 28,), float("nan"), device=dev); del junk      # poison the allocator's free memory
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
mode = sys.argv[1]
if mode == "graph":
    cap = CapturedTrainStep(model, crit, opt, 0.1, s, tg)
out = []
for i in range(26):
    if mode == "graph":
        l, ld, gn = cap(s, tg)
        out.append("%.3g" % l.item())
        if not torch.isfinite(model.store.flat_p).all(): out.append("<-nonfinite p"); break
    else:
        r = train_step(model, crit, s, tg, opt, None, 0.1)
        out.append("%.3g" % r[0])
print(mode, " ".join(out))
sv = model._saved
print("logits finite", bool(torch.isfinite(sv["hs16"].float()).all()), "gnorm", float(opt.grad_norm))
