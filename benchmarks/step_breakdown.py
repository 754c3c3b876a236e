"""Per-shape time of the GEMM-family launches inside ONE real training step (HIP events per launch)."""
import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from reftr_amd import hip
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
model.store.P["bbox_embed.layers.2.weight"].normal_(0, 0.02); model.mark_dirty()
opt = FusedAdamW(model)
model.train()
samples, targets = bench.synth_batch(8, 640, 640, 40, dev, 1234)
s = {k: v.to(dev) for k, v in samples.items() if k not in ("img", "img_mask")}
s["img"] = NestedTensor(samples["img"].to(dev), samples["img_mask"].to(dev))
tg = [{k: v.to(dev) for k, v in t.items()} for t in targets]
for _ in range(3):
    train_step(model, crit, s, tg, opt, None, 0.1)
recs = []
hip.set_launch_timer(recs)
train_step(model, crit, s, tg, opt, None, 0.1)
torch.cuda.synchronize()
hip.set_launch_timer(None)
agg = collections.OrderedDict()
for r in recs:
    t = r["start"].elapsed_time(r["end"]) * 1e3
    a = agg.setdefault(r["tag"], [0, 0.0, 0.0])
    a[0] += 1; a[1] += t; a[2] += r["flops"]
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for v in agg.values())
print("total GEMM-family time %.2f ms, %d launches" % (tot / 1e3, len(recs)))
def hbm_us(tag):
    """the launch's algorithmic bytes (operands once, bf16 in / bf16 out, fp32 dw) at 5 TB/s"""
    if tag is None or len(tag) != 12:
        return float("nan")
    B, SH, SW, SC, DH, DW, N, KH, KW, st, pd = tag[1:]
    a, o, w = B * SH * SW * SC * 2, B * DH * DW * N * 2, N * KH * KW * SC
    return (a + o + w * (4 if tag[0] == "W" else 2)) / 5e12 * 1e6
print("%-4s %-44s %5s %9s %8s %8s %8s %6s" % ("kind", "B,SH,SW,SC,DH,DW,N,KH,KW,s,p", "n", "total us", "avg us", "TF", "hbm us", "x hbm"))
for tag, (n, t, f) in rows[:int(os.environ.get("ROWS", "70"))]:
    h = hbm_us(tag)
    tag = tag or ("grouped",)
    print("%-4s %-44s %5d %9.1f %8.1f %8.1f %8.1f %6.1f" % (tag[0], ",".join(map(str, tag[1:])), n, t, t / n, f / t / 1e6, h, t / n / h))
