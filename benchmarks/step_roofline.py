"""Per-shape time of every GEMM-family launch inside one real training step against its own roofline
(max(FLOPs / 2.5 PF, compulsory bytes / 6.3 TB/s)), measured with the stream backlogged as in bench.py."""
import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from reftr_amd import hip
from reftr_amd.engine_vg import CapturedTrainStep
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
cap = CapturedTrainStep(model, crit, opt, 0.1, s, tg)
model.net.side.enabled = False
recs = []
torch.cuda.synchronize(); torch.cuda._sleep(int(2.4e9 * 0.12))
hip.set_launch_timer(recs)
cap._fwd_bwd(); cap._opt()
torch.cuda.synchronize()
hip.set_launch_timer(None)
agg = collections.OrderedDict()
for r in recs:
    t = r["start"].elapsed_time(r["end"]) * 1e3
    a = agg.setdefault(r["tag"] or ("grp", r["kind"]), [0, 0.0, 0.0, 0.0])
    a[0] += 1; a[1] += t; a[2] += r["flops"]; a[3] += r["bytes"]
def roof(f, b): return max(f / 2.5e15, b / 6.3e12) * 1e6
rows = sorted(agg.items(), key=lambda kv: -(kv[1][1] - roof(kv[1][2], kv[1][3])))
tot = sum(v[1] for v in agg.values()); troof = sum(roof(v[2], v[3]) for v in agg.values())
print("total GEMM-family time %.2f ms, roofline %.2f ms, %d launches" % (tot / 1e3, troof / 1e3, len(recs)))
print("%-4s %-44s %4s %9s %8s %8s %7s %7s" % ("kind", "B,SH,SW,SC,DH,DW,N,KH,KW,s,p", "n", "total us", "avg us", "roof us", "TF", "TB/s"))
for tag, (n, t, f, b) in rows[:70]:
    print("%-4s %-44s %4d %9.1f %8.1f %8.1f %7.1f %7.2f" % (tag[0], ",".join(map(str, tag[1:])), n, t, t / n, roof(f, b) / n, f / t / 1e6, b / t / 1e6))
