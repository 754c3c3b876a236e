"""Train / eval loops with the reference's signatures (engine_vg.py:22-25,82): the loop body of
train_one_epoch is the hot path this library accelerates (engine_vg.py:40-72).

Differences that are deliberate and documented in INTEGRATION.md:
  * `optimizer` is reftr_amd.optim.FusedAdamW (clip + AdamW in one kernel); a plain torch optimizer also works;
  * the prefetcher only uses a side HIP stream when the device is a GPU (the reference constructs
    torch.cuda.Stream() unconditionally, engine_vg.py:240).
"""
import math
import sys

import torch

from .util import misc as utils
from .util.box_ops import box_cxcywh_to_xyxy, box_iou


def _to_device(samples, targets, device, non_blocking=True):
    s = {k: (v.to(device, non_blocking=non_blocking) if hasattr(v, "to") else v) for k, v in samples.items()}
    t = [{k: (v.to(device, non_blocking=non_blocking) if hasattr(v, "to") else v) for k, v in tg.items()} for tg in targets]
    return s, t


class data_prefetcher:
    """H2D copies on a side stream, one batch ahead (engine_vg.py:234-291)."""

    def __init__(self, loader, device, prefetch=True):
        self.loader = iter(loader)
        self.device = torch.device(device)
        self.prefetch = prefetch and self.device.type == "cuda"
        if self.prefetch:
            self.stream = torch.cuda.Stream()
            self.preload()

    def preload(self):
        try:
            s, t = next(self.loader)
        except StopIteration:
            self.next_samples = self.next_targets = None
            return
        with torch.cuda.stream(self.stream):
            self.next_samples, self.next_targets = _to_device(s, t, self.device)

    def next(self):
        if not self.prefetch:
            try:
                s, t = next(self.loader)
            except StopIteration:
                return None, None
            return _to_device(s, t, self.device, non_blocking=False)
        torch.cuda.current_stream().wait_stream(self.stream)
        s, t = self.next_samples, self.next_targets
        if s is not None:
            for v in s.values():
                if hasattr(v, "record_stream"):
                    v.record_stream(torch.cuda.current_stream())
            for tg in t:
                for v in tg.values():
                    if hasattr(v, "record_stream"):
                        v.record_stream(torch.cuda.current_stream())
        self.preload()
        return s, t


def train_step(model, criterion, samples, targets, optimizer, lr_scheduler=None, max_norm=0.0):
    """The loop body, engine_vg.py:40-72.  Returns (loss_value, reduced scaled dict, reduced unscaled dict,
    grad_norm tensor)."""
    outputs = model(samples)
    loss_dict = criterion(outputs, targets)
    weight_dict = criterion.weight_dict
    losses = sum(loss_dict[k] * weight_dict[k] for k in loss_dict.keys() if k in weight_dict)
    loss_dict_reduced = utils.reduce_dict(loss_dict)
    unscaled = {f"{k}_unscaled": v for k, v in loss_dict_reduced.items()}
    scaled = {k: v * weight_dict[k] for k, v in loss_dict_reduced.items() if k in weight_dict}
    loss_value = sum(scaled.values()).item()
    if not math.isfinite(loss_value):
        print("Loss is {}, stopping training".format(loss_value))
        print(loss_dict_reduced)
        sys.exit(1)
    optimizer.zero_grad()
    losses.backward()
    if hasattr(optimizer, "clip_grad_norm_"):
        grad_total_norm = optimizer.clip_grad_norm_(max_norm)
    elif max_norm > 0:
        grad_total_norm = torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm)
    else:
        grad_total_norm = torch.zeros(())
    optimizer.step()
    if lr_scheduler is not None:
        lr_scheduler.step()
    return loss_value, scaled, unscaled, grad_total_norm


def train_one_epoch(model, criterion, data_loader, optimizer, lr_scheduler, device, epoch, max_norm=0):
    model.train()
    criterion.train()
    metric_logger = utils.MetricLogger(delimiter="  ")
    metric_logger.add_meter("lr", utils.SmoothedValue(window_size=1, fmt="{value:.6f}"))
    metric_logger.add_meter("grad_norm", utils.SmoothedValue(window_size=1, fmt="{value:.2f}"))
    header = "Epoch: [{}]".format(epoch)
    prefetcher = data_prefetcher(data_loader, device, prefetch=True)
    samples, targets = prefetcher.next()
    for _ in metric_logger.log_every(range(len(data_loader)), 50, header):
        loss_value, scaled, unscaled, gnorm = train_step(model, criterion, samples, targets, optimizer, lr_scheduler, max_norm)
        metric_logger.update(loss=loss_value, **scaled, **unscaled)
        metric_logger.update(lr=optimizer.param_groups[0]["lr"])
        metric_logger.update(grad_norm=gnorm)
        samples, targets = prefetcher.next()
    metric_logger.synchronize_between_processes()
    print("Averaged stats:", metric_logger)
    return {k: meter.global_avg for k, meter in metric_logger.meters.items()}


@torch.no_grad()
def evaluate(model, criterion, postprocessors, data_loader, device, output_dir=None, visualize=False):
    """REC part of engine_vg.evaluate (engine_vg.py:82-225): Acc@0.5 and mean IoU of the predicted boxes."""
    model.eval()
    criterion.eval()
    metric_logger = utils.MetricLogger(delimiter="  ")
    sum_accu = torch.zeros(1, device=device); sum_iou = torch.zeros(1, device=device); cnt = torch.zeros(1, device=device)
    results_dict = {}
    prefetcher = data_prefetcher(data_loader, device, prefetch=True)
    samples, targets = prefetcher.next()
    for _ in metric_logger.log_every(range(len(data_loader)), 50, "Test:"):
        outputs = model(samples)
        loss_dict = criterion(outputs, targets)
        weight_dict = criterion.weight_dict
        red = utils.reduce_dict(loss_dict)
        metric_logger.update(loss=sum(v * weight_dict[k] for k, v in red.items() if k in weight_dict))
        sizes = torch.stack([t["size"] for t in targets], dim=0)
        results = postprocessors["bbox"](outputs, sizes, scale_to_original_shape=False)
        for res, tg in zip(results, targets):
            gt = box_cxcywh_to_xyxy(tg["boxes"])
            iou = torch.diag(box_iou(res["boxes"], gt)[0])
            sum_accu += (iou > 0.5).float().sum(); sum_iou += iou.sum(); cnt += iou.numel()      # engine_vg.py:131-140
            if "image_id" in tg:
                results_dict[int(tg["image_id"])] = res["boxes"].cpu()
        samples, targets = prefetcher.next()
    if utils.is_dist_avail_and_initialized():
        for t in (sum_accu, sum_iou, cnt):
            torch.distributed.all_reduce(t)
    metric_logger.synchronize_between_processes()
    stats = {k: meter.global_avg for k, meter in metric_logger.meters.items()}
    stats["accuracy_iou0.5"] = float(sum_accu / cnt.clamp(min=1))
    stats["miou"] = float(sum_iou / cnt.clamp(min=1))
    return stats, results_dict
