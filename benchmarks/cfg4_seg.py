"""configs[3] shape (RefCOCO + segmentation head, ResNet-50, 640x640, batch 8/GPU): the REC+RES multitask training step.
Prints ms/step eager and under hipGraph replay plus the per-kernel-family time of the RES head."""
import os, sys, time, argparse, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from reftr_amd import hip
from reftr_amd.engine_vg import CapturedTrainStep, train_step
from reftr_amd.models import layout as Lm
from reftr_amd.models.criterion import CriterionVGOnePhraseSeg
from reftr_amd.models.reftr_transformer import RefTR
from reftr_amd.optim import FusedAdamW
from reftr_amd.util.misc import NestedTensor

ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=10); a = ap.parse_args()
dev = torch.device("cuda")
B, S_ = 8, 640
cfg = Lm.ModelConfig(masks=True)
model = RefTR(cfg, device=dev, aux_loss=False)
wd = {"loss_giou": 1.0, "loss_bbox": 1.0, "loss_dice": 1.0, "loss_mask": 1.0, "loss_cem": 1.0}
crit = CriterionVGOnePhraseSeg(wd, ["masks", "boxes"])
model.store.P["bbox_embed.layers.2.weight"].normal_(0, 0.02); model.mark_dirty()
opt = FusedAdamW(model, lr=1e-4, lr_backbone=1e-5, weight_decay=1e-4)
model.train()
samples, targets = bench.synth_batch(B, S_, S_, 40, dev, 1234)
g = torch.Generator().manual_seed(3)
s = {k: v.to(dev) for k, v in samples.items() if k not in ("img", "img_mask")}
s["img"] = NestedTensor(samples["img"].to(dev), samples["img_mask"].to(dev))
tg = []
for t in targets:
    cx, cy, bw, bh = [float(v) for v in t["boxes"][0]]
    yy, xx = torch.meshgrid(torch.arange(S_), torch.arange(S_), indexing="ij")
    m = (((xx + 0.5) / S_ - cx).abs() < bw / 2) & (((yy + 0.5) / S_ - cy).abs() < bh / 2)
    tg.append({"boxes": t["boxes"].to(dev), "labels": t["labels"].to(dev), "masks": m[None].to(dev)})
for _ in range(2):
    lv = train_step(model, crit, s, tg, opt, None, 0.1)[0]
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(a.steps):
    lv = train_step(model, crit, s, tg, opt, None, 0.1)[0]
torch.cuda.synchronize(); te = (time.perf_counter() - t0) / a.steps
print("eager   : %.2f ms/step, %.1f img/s, loss %.4f" % (te * 1e3, B / te, lv))
cap = CapturedTrainStep(model, crit, opt, 0.1, s, tg)
for _ in range(2): cap(s, tg)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(a.steps):
    l, ld, gn = cap(s, tg); lv = float(l)
torch.cuda.synchronize(); tg_ = (time.perf_counter() - t0) / a.steps
assert torch.isfinite(model.store.flat_g).all() and lv == lv
print("hipgraph: %.2f ms/step, %.1f img/s, loss %.4f (%s), peak mem %.1f GB" % (
    tg_ * 1e3, B / tg_, lv, " ".join("%s %.3f" % (k, float(v)) for k, v in ld.items()), torch.cuda.max_memory_allocated() / 2**30))
