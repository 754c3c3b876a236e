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
cfg = Lm.ModelConfig() if "--full" in sys.argv else Lm.ModelConfig(enc_layers=2, dec_layers=2, bert=Lm.BertConfig(layers=2))
NAUX = cfg.dec_layers - 1
def make():
    torch.manual_seed(0)
    model = RefTR(cfg, device=dev)
    wd = {"loss_giou": 1.0, "loss_bbox": 1.0}
    wd.update({f"{k}_{i}": v for i in range(NAUX) for k, v in list(wd.items())})
    crit = CriterionVGMultiPhrase(wd, ["boxes"])
    model.store.P["bbox_embed.layers.2.weight"].normal_(0, 0.02); model.mark_dirty()
    opt = FusedAdamW(model)
    model.train()
    return model, crit, opt
samples, targets = bench.synth_batch(8, 640, 640, 40, dev, 1234) if '--full' in sys.argv else bench.synth_batch(4, 320, 320, 40, dev, 1234)
s = {k: v.to(dev) for k, v in samples.items() if k not in ("img", "img_mask")}
s["img"] = NestedTensor(samples["img"].to(dev), samples["img_mask"].to(dev))
tg = [{k: v.to(dev) for k, v in t.items()} for t in targets]
m1, c1, o1 = make()
e = [(lambda r: (r[0], 0, 0, float(r[3])))(train_step(m1, c1, s, tg, o1, None, 0.1)) for _ in range(40)]
print("eager ", ["%.4f" % x[0] for x in e], ["%.3f" % float(x[3]) for x in e])
m2, c2, o2 = make()
cap = CapturedTrainStep(m2, c2, o2, 0.1, s, tg, warmup=2)
g = []
for _ in range(38):
    l, ld, gn = cap(s, tg)
    g.append((l.item(), float(gn)))
    if not torch.isfinite(m2.store.flat_g).all(): print('nonfinite grad at', len(g))
print("graph ", ["%.4f" % x[0] for x in g], ["%.3f" % x[1] for x in g], "(first graph step = eager step 3)")
print("step_dev", int(o2.step_dev), "seed_dev", int(m2.seed_dev), "finite p", bool(torch.isfinite(m2.store.flat_p).all()))
