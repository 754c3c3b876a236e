"""Stage timeline of the cooperative decoder launch (REFTR_DEC_TRACE=1): microseconds between the stage boundaries of workgroup 0
(a core workgroup: every stage) and of the last workgroup (attention + linear1 only), inside a captured training step.
    REFTR_DEC_TRACE=1 python benchmarks/dec_trace.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("REFTR_DEC_TRACE", "1")
import bench  # noqa: E402
from reftr_amd import hip as H  # noqa: E402
from reftr_amd.engine_vg import CapturedTrainStep  # noqa: E402
from reftr_amd.models import layout as Lm  # noqa: E402
from reftr_amd.models.criterion import CriterionVGMultiPhrase  # noqa: E402
from reftr_amd.models.reftr_transformer import RefTR  # noqa: E402
from reftr_amd.optim import FusedAdamW  # noqa: E402


def main():
    dev = torch.device("cuda")
    torch.zeros(1, device=dev)
    H.decoder_trace(readback=False)          # allocates the stamp buffer outside any stream capture
    cfg = Lm.ModelConfig()
    model = RefTR(cfg, device=dev, aux_loss=True)
    wd = {"loss_giou": 1.0, "loss_bbox": 1.0}
    wd.update({f"{k}_{i}": v for i in range(cfg.dec_layers - 1) for k, v in list(wd.items())})
    crit = CriterionVGMultiPhrase(wd, ["boxes"])
    opt = FusedAdamW(model, lr=1e-4, lr_backbone=1e-5, weight_decay=1e-4)
    samples, targets = bench.synth_batch(8, 640, 640, 40, dev, 1234)
    from reftr_amd.util.misc import NestedTensor
    s = {k: v.to(dev) for k, v in samples.items() if k not in ("img", "img_mask")}
    s["img"] = NestedTensor(samples["img"].to(dev), samples["img_mask"].to(dev))
    tg = [{k: v.to(dev) for k, v in t.items()} for t in targets]
    cap = CapturedTrainStep(model, crit, opt, 0.1, s, tg)
    for _ in range(10):
        cap(s, tg)[0].item()
    torch.cuda.synchronize()
    for name, st in zip(("workgroup 0", "last workgroup"), H.decoder_trace()):
        n = max(i for i, v in enumerate(st) if v) + 1
        d = [((st[i + 1] - st[i]) & 0xFFFFFFFF) / 100.0 for i in range(n - 1)]
        print(f"{name}: {n} stamps, total {sum(d):.1f} us")
        print("  " + " ".join(f"{x:.2f}" for x in d))


if __name__ == "__main__":
    main()
