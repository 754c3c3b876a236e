"""True marginal cost (hipGraph replay, 2 streams) of the model's parts: ms/step of the cfg2 step with fewer decoder /
encoder / BERT layers.  rocprofv3 inflates ~2 us kernels, so this -- not the kernel trace -- sizes the small-kernel chains."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from reftr_amd.engine_vg import CapturedTrainStep
from reftr_amd.models import layout as Lm
from reftr_amd.models.criterion import CriterionVGMultiPhrase
from reftr_amd.models.reftr_transformer import RefTR
from reftr_amd.optim import FusedAdamW
from reftr_amd.util.misc import NestedTensor
dev = torch.device("cuda")
samples, targets = bench.synth_batch(8, 640, 640, 40, dev, 1234)
s = {k: v.to(dev) for k, v in samples.items() if k not in ("img", "img_mask")}
s["img"] = NestedTensor(samples["img"].to(dev), samples["img_mask"].to(dev))
tg = [{k: v.to(dev) for k, v in t.items()} for t in targets]
def run(**kw):
    bl = kw.pop("bert", 12)
    cfg = Lm.ModelConfig(bert=Lm.BertConfig(layers=bl), **kw)
    model = RefTR(cfg, device=dev)
    wd = {"loss_giou": 1.0, "loss_bbox": 1.0}
    wd.update({f"{k}_{i}": v for i in range(cfg.dec_layers - 1) for k, v in list(wd.items())})
    crit = CriterionVGMultiPhrase(wd, ["boxes"])
    model.store.P["bbox_embed.layers.2.weight"].normal_(0, 0.02); model.mark_dirty()
    opt = FusedAdamW(model); model.train()
    cap = CapturedTrainStep(model, crit, opt, 0.1, s, tg)
    for _ in range(3): cap(s, tg)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): cap(s, tg)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 20 * 1e3
base = run()
print("full model                 %.2f ms" % base)
for name, kw, n in (("decoder 6 -> 1 layers", dict(dec_layers=1), 5), ("encoder 6 -> 1 layers", dict(enc_layers=1), 5),
                    ("BERT 12 -> 1 layers", dict(bert=1), 11)):
    t = run(**kw)
    print("%-26s %.2f ms  -> %.3f ms per layer" % (name, t, (base - t) / n))
