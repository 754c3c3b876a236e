"""Where do the small aten fill / copy / elementwise kernels of one step come from? (torch.profiler with python stacks)"""
import os, sys, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
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
opt = FusedAdamW(model); model.train()
samples, targets = bench.synth_batch(8, 640, 640, 40, dev, 1234)
s = {k: v.to(dev) for k, v in samples.items() if k not in ("img", "img_mask")}
s["img"] = NestedTensor(samples["img"].to(dev), samples["img_mask"].to(dev))
tg = [{k: v.to(dev) for k, v in t.items()} for t in targets]
cap = CapturedTrainStep(model, crit, opt, 0.1, s, tg)
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    cap._fwd_bwd(); cap._opt()
torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    if e.name in ("aten::fill_", "aten::zero_", "aten::copy_", "aten::add", "aten::mul", "aten::add_", "aten::clone", "aten::contiguous", "aten::to", "aten::eq", "aten::sum", "aten::cat"):
        st = [f for f in e.stack if "reftr_amd" in f or "bench" in f]
        cnt[(e.name, st[0] if st else "(autograd/other)")] += 1
for (n, w), c in cnt.most_common(60):
    print("%4d %-16s %s" % (c, n, w[-110:]))
