import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from reftr_amd.engine_vg import train_step, CapturedTrainStep, _copy_batch
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
for trial in range(5):
    model = RefTR(cfg, device=dev)
    crit = CriterionVGMultiPhrase(wd, ["boxes"])
    torch.manual_seed(1234)
    model.store.P["bbox_embed.layers.2.weight"].normal_(0, 0.02); model.mark_dirty()
    opt = FusedAdamW(model, lr=1e-4, lr_backbone=1e-5, weight_decay=1e-4)
    model.train()
    cap = CapturedTrainStep(model, crit, opt, 0.1, s, tg)
    print("trial", trial, "after capture: p finite", bool(torch.isfinite(model.store.flat_p).all()), "m finite", bool(torch.isfinite(opt.m).all()),
          "step_dev", int(opt.step_dev), "gn", float(opt.grad_norm), flush=True)
    for i in range(8):
        _copy_batch(cap.s, cap.t, s, tg)
        cap.g_fb.replay()
        torch.cuda.synchronize()
        G = model.store.G
        bad = [k for k in G if not torch.isfinite(G[k]).all()]
        loss = cap.out[0].item()
        cap.g_opt.replay()
        torch.cuda.synchronize()
        pf = bool(torch.isfinite(model.store.flat_p).all())
        if bad or not pf or not (loss == loss) or abs(loss) > 100:
            print("   step", i, "loss", loss, "bad grads", len(bad), bad[:6], "p finite", pf, "sq", float(opt.sq), "gn", float(opt.grad_norm), "step", int(opt.step_dev))
            sv = model._saved
            for k in ("c5", "mem32", "hs16", "y1", "y2", "ip"):
                print("      ", k, bool(torch.isfinite(sv[k].float()).all()))
            break
    else:
        print("   ok, last loss", loss)
    del cap, model, opt
