"""Inference side of the path (engine_vg.evaluate's loop body: forward + criterion + box post-process), cfg2 shapes:
eager launches vs one hipGraph of the forward."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from reftr_amd.models import layout as Lm
from reftr_amd.models.criterion import CriterionVGMultiPhrase
from reftr_amd.models.post_process import PostProcessVGMultiPhrase
from reftr_amd.models.reftr_transformer import RefTR
from reftr_amd.util.misc import NestedTensor
dev = torch.device("cuda")
B = int(os.environ.get("B", "8"))
model = RefTR(Lm.ModelConfig(), device=dev, aux_loss=True)
wd = {"loss_giou": 1.0, "loss_bbox": 1.0}
wd.update({f"{k}_{i}": v for i in range(5) for k, v in list(wd.items())})
crit = CriterionVGMultiPhrase(wd, ["boxes"]); post = PostProcessVGMultiPhrase()
model.store.P["bbox_embed.layers.2.weight"].normal_(0, 0.02); model.mark_dirty()
model.eval(); crit.eval()
samples, targets = bench.synth_batch(B, 640, 640, 40, dev, 1234)
s = {k: v.to(dev) for k, v in samples.items() if k not in ("img", "img_mask")}
s["img"] = NestedTensor(samples["img"].to(dev), samples["img_mask"].to(dev))
tg = [{k: v.to(dev) for k, v in t.items()} for t in targets]
sizes = torch.tensor([[640, 640]] * B, device=dev)
def body():
    out = model(s)
    ld = crit(out, tg)
    res = post(out, sizes)
    return out, ld, res
with torch.no_grad():
    for _ in range(3): body()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): body()
    torch.cuda.synchronize(); te = (time.perf_counter() - t0) / 20
    print("eager   : %.2f ms/batch  %.0f img/s" % (te * 1e3, B / te))
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        body()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    with torch.cuda.graph(g):
        out = model(s)
    for _ in range(3): g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50):
        g.replay(); ld = crit(out, tg); res = post(out, sizes)
    torch.cuda.synchronize(); tgph = (time.perf_counter() - t0) / 50
    print("hipgraph: %.2f ms/batch  %.0f img/s (forward replayed, criterion + post-process eager)" % (tgph * 1e3, B / tgph))
