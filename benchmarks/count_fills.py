"""Which host-side calls launch torch fill kernels during ONE eager training step (size histogram by call site)."""
import collections, os, sys, traceback, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("REFTR_STREAMS", "0")
import bench
from reftr_amd.engine_vg import train_step
from reftr_amd.models import layout as Lm
from reftr_amd.models.criterion import CriterionVGMultiPhrase
from reftr_amd.models.reftr_transformer import RefTR
from reftr_amd.optim import FusedAdamW
from reftr_amd.util.misc import NestedTensor
dev = torch.device("cuda")
model = RefTR(Lm.ModelConfig(), device=dev)
wd = {"loss_giou": 1.0, "loss_bbox": 1.0}; wd.update({f"{k}_{i}": v for i in range(5) for k, v in list(wd.items())})
crit = CriterionVGMultiPhrase(wd, ["boxes"]); opt = FusedAdamW(model); model.train()
samples, targets = bench.synth_batch(8, 640, 640, 40, dev, 1234)
s = {k: v.to(dev) for k, v in samples.items() if k not in ("img", "img_mask")}
s["img"] = NestedTensor(samples["img"].to(dev), samples["img_mask"].to(dev))
tg = [{k: v.to(dev) for k, v in t.items()} for t in targets]
for _ in range(2): train_step(model, crit, s, tg, opt, None, 0.1)
sites = collections.Counter(); nbytes = collections.Counter()
def site():
    for f in reversed(traceback.extract_stack()[:-2]):
        if "reftr_amd" in f.filename or "bench" in f.filename: return f"{os.path.basename(f.filename)}:{f.lineno}"
    return "?"
def wrap(fn, name):
    def w(*a, **k):
        r = fn(*a, **k)
        t = r if torch.is_tensor(r) else (a[0] if a and torch.is_tensor(a[0]) else None)
        if t is not None and t.is_cuda:
            key = name + " @ " + site(); sites[key] += 1; nbytes[key] += t.numel() * t.element_size()
        return r
    return w
torch.zeros = wrap(torch.zeros, "zeros"); torch.zeros_like = wrap(torch.zeros_like, "zeros_like"); torch.full = wrap(torch.full, "full")
torch.Tensor.zero_ = wrap(torch.Tensor.zero_, "zero_"); torch.Tensor.fill_ = wrap(torch.Tensor.fill_, "fill_"); torch.ones = wrap(torch.ones, "ones")
torch.Tensor.new_zeros = wrap(torch.Tensor.new_zeros, "new_zeros")
train_step(model, crit, s, tg, opt, None, 0.1); torch.cuda.synchronize()
for k, n in sites.most_common(40): print("%4d  %8.2f MB  %s" % (n, nbytes[k] / 1e6, k))
print("total", sum(sites.values()), "calls", sum(nbytes.values()) / 1e6, "MB")
