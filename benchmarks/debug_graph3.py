import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from reftr_amd.engine_vg import train_step, CapturedTrainStep
from reftr_amd.models import layout as Lm
from reftr_amd.models.criterion import CriterionVGMultiPhrase
from reftr_amd.models.reftr_transformer import RefTR
from reftr_amd.optim import FusedAdamW
from reftr_amd.util.misc import NestedTensor
dev = torch.device("cuda", 0)
cfg = Lm.ModelConfig()
wd = {"loss_giou": 1.0, "loss_bbox": 1.0}
wd.update({f"{k}_{i}": v for i in range(5) for k, v in list(wd.items())})
samples, targets = bench.synth_batch(8, 640, 640, 40, dev, 1234)
s = {k: v.to(dev) for k, v in samples.items() if k not in ("img", "img_mask")}
s["img"] = NestedTensor(samples["img"].to(dev), samples["img_mask"].to(dev))
tg = [{k: v.to(dev) for k, v in t.items()} for t in targets]
for trial in range(6):
    model = RefTR(cfg, device=dev)
    crit = CriterionVGMultiPhrase(wd, ["boxes"])
    torch.manual_seed(1234)
    model.store.P["bbox_embed.layers.2.weight"].normal_(0, 0.02); model.mark_dirty()
    opt = FusedAdamW(model, lr=0.0, lr_backbone=0.0, weight_decay=0.0)      # frozen weights: grads must repeat exactly
    model.eval()
    cap = CapturedTrainStep(model, crit, opt, 0.1, s, tg)
    ref = None
    msg = []
    for i in range(4):
        l, ld, gn = cap(s, tg)
        g = model.store.flat_g.clone()
        fin = bool(torch.isfinite(g).all())
        if ref is None: ref = g
        msg.append("%.4f/%s/%.2e" % (l.item(), fin, float((g - ref).norm() / ref.norm())))
    # eager reference on the same weights
    r = train_step(model, crit, s, tg, opt, None, 0.1)
    ge = model.store.flat_g
    print("trial", trial, " ".join(msg), "| eager", "%.4f" % r[0], "rel(graph, eager) %.2e" % float((ref - ge).norm() / ge.norm()), flush=True)
    del cap, model, opt
