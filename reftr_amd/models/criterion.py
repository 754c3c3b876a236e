"""CriterionVGMultiPhrase on the fused box-loss kernel.

Same constructor / forward contract and loss keys as the reference (models/criterion.py:100-202):
`criterion(outputs, targets) -> {'loss_bbox', 'loss_giou', 'loss_bbox_i', 'loss_giou_i', ...}`, `.weight_dict`.
All decoder layers are evaluated by ONE rt_box_loss launch over outputs['pred_logits'] (the model's
pre-sigmoid stack); its backward is a second launch of the same kernel weighted by the incoming gradient of
every loss entry, so any weight_dict the training loop applies (engine_vg.py:43) is honoured.
"""
import torch
from torch import nn

from .. import hip as H
from ..util.misc import get_world_size, is_dist_avail_and_initialized


class _BoxLossFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, valid_u8, targets, tgt_off, num_boxes):
        losses, _, _ = H.box_loss(logits, valid_u8, targets, tgt_off, num_boxes, want_grad=False)
        ctx.save_for_backward(logits, valid_u8, targets, tgt_off, num_boxes)
        return losses

    @staticmethod
    def backward(ctx, g):
        logits, valid_u8, targets, tgt_off, num_boxes = ctx.saved_tensors
        _, _, dl = H.box_loss(logits, valid_u8, targets, tgt_off, num_boxes, want_grad=True,
                              weights=g.contiguous().to(torch.float32))
        return dl, None, None, None, None


class CriterionVGMultiPhrase(nn.Module):
    def __init__(self, weight_dict, losses):
        super().__init__()
        self.weight_dict = weight_dict
        self.losses = losses
        self._off_cache = {}
        self._nb_cache = {}
        self.num_boxes_static = None      # CapturedTrainStep: device scalar it refreshes (all-reduced) before every replay

    def _targets(self, targets, device):
        lens = tuple(int(t["boxes"].shape[0]) for t in targets)
        if lens not in self._off_cache:
            off = [0]
            for n in lens:
                off.append(off[-1] + n)
            self._off_cache[lens] = torch.tensor(off, dtype=torch.int32, device=device)
        boxes = torch.cat([t["boxes"].to(device, torch.float32) for t in targets], dim=0).contiguous()
        if boxes.shape[0] == 0:
            # a batch without any target box (criterion.py:176-180 clamps num_boxes to 1, every loss is an empty sum = 0): the
            # kernel needs a non-null pointer, the all-zero offsets make every image's target range empty
            boxes = torch.zeros(1, 4, dtype=torch.float32, device=device)
        return boxes, self._off_cache[lens]

    def num_boxes(self, targets, device):
        """criterion.py:176-180: the number of target boxes of the GLOBAL batch averaged over the ranks (one sum all-reduce of
        a device scalar; the clamp(min=1) is applied by the kernel).  Kept on the device: no host sync."""
        nb = float(sum(len(t["labels"]) for t in targets))
        if self.num_boxes_static is not None:
            return self.num_boxes_static
        if is_dist_avail_and_initialized():
            num_boxes = torch.tensor([nb], dtype=torch.float32, device=device)
            torch.distributed.all_reduce(num_boxes)
            return num_boxes / get_world_size()
        if nb not in self._nb_cache:               # cached device scalar: no H2D copy in the steady state / under capture
            self._nb_cache[nb] = torch.tensor([nb], dtype=torch.float32, device=device)
        return self._nb_cache[nb]

    def forward(self, outputs, targets):
        assert "boxes" in self.losses and "pred_logits" in outputs, \
            "the HIP criterion consumes the model's pre-sigmoid logits (outputs['pred_logits'])"
        logits = outputs["pred_logits"]                       # [NL, B, P, K, 4]
        if "aux_outputs" not in outputs:
            logits = logits[-1:]
        logits = logits.contiguous()
        device = logits.device
        num_boxes = self.num_boxes(targets, device)
        boxes, off = self._targets(targets, device)
        valid = H.as_u8(outputs["phrase_mask"])
        losses = _BoxLossFunction.apply(logits, valid, boxes, off, num_boxes)     # [NL, 2]
        return self._loss_dict(losses)

    def _loss_dict(self, losses):
        self._last_box = losses
        nl = losses.shape[0]
        out = {"loss_bbox": losses[nl - 1, 0], "loss_giou": losses[nl - 1, 1]}
        for i in range(nl - 1):
            out[f"loss_bbox_{i}"] = losses[i, 0]
            out[f"loss_giou_{i}"] = losses[i, 1]
        return out

    def prepare(self, targets, device):
        """The part of `forward` that only reads the targets: (boxes [sum n_i, 4] fp32, offsets, num_boxes).  The captured
        training step issues it at the head of the step on the side stream, off the chain the model's forward runs on."""
        num_boxes = self.num_boxes(targets, device)
        boxes, off = self._targets(targets, device)
        return boxes, off, num_boxes

    def loss_and_grad(self, logits, phrase_mask, prepared, aux):
        """Losses AND d total / d logits in one rt_box_loss launch, for a total that is the weighted sum of the box losses only
        (`weighted_total` without mask / CEM terms): the gradient of that sum with respect to losses[i, j] IS weight[i, j], so the
        launch is handed the weights the autograd path would have delivered (`_BoxLossFunction.backward`) -- same kernel, same
        arguments, same bits -- without the forward-only launch and the five autograd kernels between the two.
        Returns (loss_dict, losses [NL, 2], dlogits [as logits])."""
        full = logits
        if not aux:
            logits = logits[-1:]
        logits = logits.contiguous()
        boxes, off, num_boxes = prepared
        valid = H.as_u8(phrase_mask)
        w = _box_weights(self, logits.shape[0], logits.device)
        losses, _, dl = H.box_loss(logits, valid, boxes, off, num_boxes, want_grad=True, weights=w)
        if not aux and full.shape[0] > 1:          # only the last layer carries a loss: the other layers' gradient is zero
            dl = torch.cat([torch.zeros_like(full[:-1]), dl], dim=0)
        return self._loss_dict(losses), losses, dl


def _box_weights(crit, nl, device):
    """weight_dict's entries for the box losses as a device tensor [NL, 2] (cached per (NL, device))."""
    key = (nl, device)
    if getattr(crit, "_wkey", None) != key:
        wd = crit.weight_dict
        w = torch.zeros(nl, 2)
        for i in range(nl):
            sfx = "" if i == nl - 1 else f"_{i}"
            w[i, 0] = wd.get("loss_bbox" + sfx, 0.0); w[i, 1] = wd.get("loss_giou" + sfx, 0.0)
        crit._wbox = w.to(device)
        crit._wmask = torch.tensor([wd.get("loss_mask", 0.0), wd.get("loss_dice", 0.0)], device=device)
        crit._wkey = key
    return crit._wbox


def _weighted_total(crit, loss_dict):
    """engine_vg.py:43 (`sum(loss_dict[k] * weight_dict[k] ...)`) as ONE weighted reduction over the loss tensors the
    kernels produced, instead of ~4 tiny autograd kernels per loss key (12 keys with aux losses)."""
    box = crit._last_box                                  # [NL, 2] = (loss_bbox, loss_giou) per decoder layer
    total = (box * _box_weights(crit, box.shape[0], box.device)).sum()
    if getattr(crit, "_last_mask", None) is not None and "loss_mask" in loss_dict:
        total = total + (crit._last_mask * crit._wmask).sum()
    if "loss_cem" in loss_dict and "loss_cem" in crit.weight_dict:
        total = total + loss_dict["loss_cem"] * crit.weight_dict["loss_cem"]
    return total


CriterionVGMultiPhrase.weighted_total = _weighted_total


class _MaskLossFunction(torch.autograd.Function):
    """Bilinear upsample + sigmoid focal + dice (rt_mask_loss); returns losses[2] = {loss_mask, loss_dice}."""

    @staticmethod
    def forward(ctx, pred, target_u8, norm):
        B, _, h, w = pred.shape
        Ht, Wt = target_u8.shape[-2:]
        pred = pred.contiguous()
        losses, sums = H.mask_loss(pred.view(-1, 1), target_u8, B, h, w, Ht, Wt, 1, norm)
        ctx.save_for_backward(pred, target_u8, sums)
        ctx.norm = norm
        return losses

    @staticmethod
    def backward(ctx, g):
        pred, target_u8, sums = ctx.saved_tensors
        B, _, h, w = pred.shape
        Ht, Wt = target_u8.shape[-2:]
        g = g.contiguous().float()
        dpred = torch.zeros(B * h * w, 1, dtype=torch.float32, device=pred.device)
        H.mask_loss(pred.view(-1, 1), target_u8, B, h, w, Ht, Wt, 1, ctx.norm, sums=sums, dpred=dpred,
                    g_focal=g[0:1].contiguous(), g_dice=g[1:2].contiguous())
        return dpred.view_as(pred), None, None


class CriterionVGOnePhraseSeg(CriterionVGMultiPhrase):
    """models/reftr_segmentation.py:305-337: losses = ['masks', 'boxes']; the mask losses are normalised by bs * num_q
    and computed against the zero-padded batch of target masks (nested_tensor_from_tensor_list, util/misc.py:288-305)."""

    def _padded_masks(self, targets, device):
        ms = [t["masks"] for t in targets]
        Ht = max(m.shape[-2] for m in ms); Wt = max(m.shape[-1] for m in ms)
        out = torch.zeros(len(ms), ms[0].shape[0], Ht, Wt, dtype=torch.uint8, device=device)
        for i, m in enumerate(ms):
            out[i, :, :m.shape[-2], :m.shape[-1]] = m.to(device, torch.uint8)
        return out

    def forward(self, outputs, targets):
        losses = {}
        if "masks" in self.losses:
            assert "pred_masks" in outputs
            pm = outputs["pred_masks"]
            bs, nq = pm.shape[:2]
            assert nq == 1, "RefTRSeg predicts one mask per image (n_ph = n_q = 1)"
            tgt = self._padded_masks(targets, pm.device)
            lm = _MaskLossFunction.apply(pm, tgt, float(bs * nq))
            self._last_mask = lm
            losses["loss_mask"], losses["loss_dice"] = lm[0], lm[1]
        if "cem_loss" in outputs:                                       # reftr_segmentation.py:237-238
            losses["loss_cem"] = outputs["cem_loss"]
        losses.update(super().forward(outputs, targets))
        return losses
