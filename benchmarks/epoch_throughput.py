"""The loop the north star names, measured as a loop: engine_vg.train_one_epoch (reference engine_vg.py:22-78) over a
synthetic fixed-shape loader at configs[1] (R50, 640 x 640, B = 8, L = 40, aux loss, dropout on, clip 0.1, AdamW) -- pinned
host batches as a DataLoader(pin_memory=True) would hand them over, H2D on the prefetch stream, staging into the graph's
input buffers, hipGraph replay, meters, lr scheduler -- next to the bench.py number (the loop BODY on a resident batch).

    python benchmarks/epoch_throughput.py [--batches 200] [--warm 20]

Prints one JSON line: epoch img/s, the resident-batch img/s of the same process, and their ratio (VERDICT r02 item 3:
>= 0.97)."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from reftr_amd.engine_vg import CapturedTrainStep, train_one_epoch  # noqa: E402
from reftr_amd.models import layout as Lm  # noqa: E402
from reftr_amd.models.criterion import CriterionVGMultiPhrase  # noqa: E402
from reftr_amd.models.reftr_transformer import RefTR  # noqa: E402
from reftr_amd.optim import FusedAdamW  # noqa: E402
from reftr_amd.util.misc import NestedTensor  # noqa: E402


class SyntheticLoader:
    """`n` batches per epoch drawn round-robin from `distinct` seeded synthetic batches held in pinned host memory."""

    def __init__(self, n, B, size, L, distinct=4):
        self.n = n
        self.batches = []
        for i in range(distinct):
            samples, targets = bench.synth_batch(B, size, size, L, "cpu", 1234 + i)
            s = {k: v.pin_memory() for k, v in samples.items() if k not in ("img", "img_mask")}
            s["img"] = NestedTensor(samples["img"].pin_memory(), samples["img_mask"].pin_memory())
            t = [{k: v.pin_memory() for k, v in tg.items()} for tg in targets]
            self.batches.append((s, t))

    def __len__(self):
        return self.n

    def __iter__(self):
        for i in range(self.n):
            yield self.batches[i % len(self.batches)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, default=200)
    ap.add_argument("--warm", type=int, default=20)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--size", type=int, default=640)
    a = ap.parse_args()
    dev = torch.device("cuda")
    cfg = Lm.ModelConfig()
    model = RefTR(cfg, device=dev, aux_loss=True)
    wd = {"loss_giou": 1.0, "loss_bbox": 1.0}
    wd.update({f"{k}_{i}": v for i in range(cfg.dec_layers - 1) for k, v in list(wd.items())})
    crit = CriterionVGMultiPhrase(wd, ["boxes"])
    torch.manual_seed(1234)
    model.store.P["bbox_embed.layers.2.weight"].normal_(0, 0.02)
    model.mark_dirty()
    opt = FusedAdamW(model, lr=1e-4, lr_backbone=1e-5, weight_decay=1e-4)
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=100000)

    # warm-up epoch: captures the graph for this shape (its warm-up updates are taken back by the engine)
    real_print = print
    import builtins
    builtins.print = lambda *x, **k: None            # the loop's progress lines
    try:
        train_one_epoch(model, crit, SyntheticLoader(a.warm, a.batch, a.size, 40), opt, sched, dev, 0, max_norm=0.1)
        torch.cuda.synchronize()
        loader = SyntheticLoader(a.batches, a.batch, a.size, 40)
        t0 = time.perf_counter()
        stats = train_one_epoch(model, crit, loader, opt, sched, dev, 1, max_norm=0.1)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
    finally:
        builtins.print = real_print
    epoch_ips = a.batches * a.batch / el

    # the resident-batch loop body of bench.py, same process, same captured graphs
    cap = next(iter(model._captured_steps.values()))
    sb, tb = cap.batch
    for _ in range(10):
        cap(sb, tb)[0].item()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 100
    for _ in range(K):
        cap(sb, tb)[0].item()
    torch.cuda.synchronize()
    body = (time.perf_counter() - t0) / K
    body_ips = a.batch / body
    real_print(json.dumps({"workload": f"configs[1]: R50 {a.size}x{a.size} B={a.batch} L=40, train_one_epoch over {a.batches} pinned synthetic batches",
                           "epoch_images_per_s": epoch_ips, "epoch_ms_per_iteration": el / a.batches * 1e3,
                           "body_images_per_s": body_ips, "body_ms_per_step": body * 1e3, "epoch_over_body": epoch_ips / body_ips,
                           "loss_avg": stats.get("loss"), "grad_norm_avg": stats.get("grad_norm")}))


if __name__ == "__main__":
    main()
