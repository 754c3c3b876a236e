"""--dilation (DC5) at the headline size: the multi-phrase step of cfg5_stress.py's shape on ResNet-50, 640 x 640, c5 at stride 16 -> 40 x 40 image
tokens, S = 90 + 1600 = 1690 per image: every attention launch of the encoder / decoder walks its inner axis in chunks
(rt_attention.hip, long inner axes).  Prints ms/step eager and under hipGraph replay; asserts finite loss / gradients."""
import os, sys, time, argparse
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reftr_amd.engine_vg import CapturedTrainStep, train_step
from reftr_amd.models import layout as Lm
from reftr_amd.models.criterion import CriterionVGMultiPhrase
from reftr_amd.models.reftr_transformer import RefTR
from reftr_amd.optim import FusedAdamW
from reftr_amd.util.misc import NestedTensor

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8); ap.add_argument("--size", type=int, default=640)
ap.add_argument("--phrases", type=int, default=4); ap.add_argument("--steps", type=int, default=10)
a = ap.parse_args()
dev = torch.device("cuda")
B, S_, L, P, Lp = a.batch, a.size, 90, a.phrases, 22
g = torch.Generator().manual_seed(5)
img = torch.randn(B, 3, S_, S_, generator=g); mask = torch.zeros(B, S_, S_, dtype=torch.bool)
for b in range(1, B, 2):
    mask[b, :, (S_ * 3) // 4:] = True; img[b, :, :, (S_ * 3) // 4:] = 0
ids = torch.zeros(B, L, dtype=torch.long); sm = torch.zeros(B, L, dtype=torch.long)
ph = torch.zeros(B, P, Lp, dtype=torch.long); pm = torch.zeros(B, P, Lp, dtype=torch.long)
pl = torch.zeros(B, P, dtype=torch.long); pr = torch.ones(B, P, dtype=torch.long)
targets = []
for b in range(B):
    n = int(torch.randint(40, L + 1, (1,), generator=g))
    ids[b, :n] = torch.randint(1000, 30000, (n,), generator=g); ids[b, 0] = 101; ids[b, n - 1] = 102; sm[b, :n] = 1
    nv = max(1, P - 2 * b)
    for j in range(P):
        if j < nv:
            k = 3 + (j % 5)
            ph[b, j, :k] = torch.randint(1000, 30000, (k,), generator=g); ph[b, j, 0] = 101; ph[b, j, k - 1] = 102; pm[b, j, :k] = 1
            pl[b, j] = 1 + 2 * j; pr[b, j] = 1 + 2 * j + (k - 2)
        else:
            ph[b, j, 0] = 101; ph[b, j, 1] = 102; pm[b, j, :2] = 1
    u = torch.rand(nv, 4, generator=g)
    targets.append({"boxes": torch.stack([0.3 + 0.4 * u[:, 0], 0.3 + 0.4 * u[:, 1], 0.1 + 0.4 * u[:, 2], 0.1 + 0.4 * u[:, 3]], -1).to(dev),
                    "labels": torch.zeros(nv, dtype=torch.long, device=dev)})
s = {"img": NestedTensor(img.to(dev), mask.to(dev)), "sentence": ids.to(dev), "sentence_mask": sm.to(dev), "phrase": ph.to(dev),
     "phrase_mask": pm.to(dev), "phrase_pos_l": pl.to(dev), "phrase_pos_r": pr.to(dev)}
cfg = Lm.ModelConfig(dilation=True)
model = RefTR(cfg, device=dev, aux_loss=True)
wd = {"loss_giou": 1.0, "loss_bbox": 1.0}
wd.update({f"{k}_{i}": v for i in range(cfg.dec_layers - 1) for k, v in list(wd.items())})
crit = CriterionVGMultiPhrase(wd, ["boxes"])
model.store.P["bbox_embed.layers.2.weight"].normal_(0, 0.02); model.mark_dirty()
opt = FusedAdamW(model, lr=1e-4, lr_backbone=1e-5, weight_decay=1e-4)
model.train()
for _ in range(2):
    lv, _, _, gn = train_step(model, crit, s, targets, opt, None, 0.1)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(a.steps):
    lv, _, _, gn = train_step(model, crit, s, targets, opt, None, 0.1)
torch.cuda.synchronize(); te = (time.perf_counter() - t0) / a.steps
assert torch.isfinite(model.store.flat_g).all() and lv == lv
print("eager   : %.2f ms/step, %.1f img/s, loss %.4f gnorm %.3f" % (te * 1e3, B / te, lv, float(gn)))
cap = CapturedTrainStep(model, crit, opt, 0.1, s, targets)
for _ in range(2): cap(s, targets)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(a.steps):
    l, _, gn = cap(s, targets); lv = float(l)
torch.cuda.synchronize(); tg_ = (time.perf_counter() - t0) / a.steps
assert torch.isfinite(model.store.flat_g).all() and lv == lv
print("hipgraph: %.2f ms/step, %.1f img/s, loss %.4f gnorm %.3f, peak mem %.1f GB" % (tg_ * 1e3, B / tg_, lv, float(gn), torch.cuda.max_memory_allocated() / 2**30))
