"""Host-side split of the time between two replays of the captured step (bench.py's loop): how long the launch call takes, how long the
read-out (`.item()`) blocks, and the same with the read-out replaced by an event the host spins on.  Prints microseconds (medians)."""
import os, statistics, sys, time
import torch
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
model = RefTR(cfg, device=dev, aux_loss=True)
wd = {"loss_giou": 1.0, "loss_bbox": 1.0}
wd.update({f"{k}_{i}": v for i in range(cfg.dec_layers - 1) for k, v in list(wd.items())})
crit = CriterionVGMultiPhrase(wd, ["boxes"])
torch.manual_seed(1234)
model.store.P["bbox_embed.layers.2.weight"].normal_(0, 0.02); model.mark_dirty()
opt = FusedAdamW(model, lr=1e-4, lr_backbone=1e-5, weight_decay=1e-4)
model.train()
samples, targets = bench.synth_batch(8, 640, 640, 40, dev, 1234)
s = {k: v.to(dev) for k, v in samples.items() if k not in ("img", "img_mask")}
s["img"] = NestedTensor(samples["img"].to(dev), samples["img_mask"].to(dev))
tg = [{k: v.to(dev) for k, v in t.items()} for t in targets]
cap = CapturedTrainStep(model, crit, opt, 0.1, s, tg)
sb, tb = cap.batch
for _ in range(10):
    cap(sb, tb)[0].item()

def loop(read):
    tl, tr, tt = [], [], []
    torch.cuda.synchronize(); t_prev = time.perf_counter()
    for _ in range(60):
        t0 = time.perf_counter(); out = cap(sb, tb); t1 = time.perf_counter(); read(out); t2 = time.perf_counter()
        tl.append(t1 - t0); tr.append(t2 - t1); tt.append(t2 - t_prev); t_prev = t2
    med = lambda v: statistics.median(v) * 1e6
    return med(tl), med(tr), med(tt)

def read_item(out):
    return out[0].item()

host = torch.empty(1, dtype=torch.float32).pin_memory()
ev = torch.cuda.Event()
def read_event_query(out):                       # async copy into pinned memory + spin on the event from Python
    host.copy_(out[0].reshape(1), non_blocking=True); ev.record()
    while not ev.query():
        pass
    return float(host[0])

def read_event_sync(out):
    host.copy_(out[0].reshape(1), non_blocking=True); ev.record(); ev.synchronize()
    return float(host[0])

for name, fn in (("item()", read_item), ("pinned copy + event.query spin", read_event_query), ("pinned copy + event.synchronize", read_event_sync), ("item()", read_item)):
    a, b, c = loop(fn)
    print("%-34s launch call %6.1f us   read-out blocks %7.1f us   iteration %7.1f us" % (name, a, b, c))
