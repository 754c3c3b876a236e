"""Which intermediate of the eval forward is not bit-reproducible run to run?  Walks model._saved of two runs."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from reftr_amd.models import layout as L
from reftr_amd.models.reftr_transformer import RefTR
from reftr_amd.util.misc import NestedTensor

def walk(o, path, out):
    if torch.is_tensor(o):
        out.append((path, o.detach().clone()))
    elif isinstance(o, dict):
        for k, v in o.items(): walk(v, f"{path}.{k}", out)
    elif isinstance(o, (list, tuple)):
        for i, v in enumerate(o): walk(v, f"{path}[{i}]", out)

torch.manual_seed(7)
model = RefTR(L.ModelConfig(), device="cuda", aux_loss=True)
model.store.P["bbox_embed.layers.2.weight"].normal_(0, 0.05); model.mark_dirty()
model.eval()
samples, targets = bench.synth_batch(8, 640, 640, 40, "cuda", 1234)
s = {k: v.cuda() for k, v in samples.items() if k not in ("img", "img_mask")}
s["img"] = NestedTensor(samples["img"].cuda(), samples["img_mask"].cuda())
runs = []
for it in range(3):
    with torch.no_grad():
        out = model(s)
    torch.cuda.synchronize()
    l = []; walk(model._saved, "saved", l); walk(out["pred_logits"], "logits", l)
    runs.append(l)
for a, b in ((0, 1), (1, 2)):
    n = 0
    print("run", a, "vs", b, len(runs[a]), "tensors")
    for (pa, ta), (pb, tb) in zip(runs[a], runs[b]):
        if ta.shape != tb.shape or not torch.equal(ta, tb):
            d = (ta.float() - tb.float()).abs().max().item() if ta.shape == tb.shape else -1
            print("  DIFF", pa, tuple(ta.shape), ta.dtype, "max|d| %.3e" % d)
            n += 1
            if n > 25: break
