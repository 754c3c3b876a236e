"""Evaluation post-processors on the HIP kernels of csrc/rt_post.hip.

Same call contracts and result dictionaries as the reference's `postprocessors['bbox']` / `['segm']`
(models/post_process.py:45-83, models/reftr_segmentation.py:282-302) so `engine_vg.evaluate` and a reference-side
caller see the same keys, shapes and dtypes -- but each is ONE kernel launch over the whole batch plus one small
device->host copy of the per-image counts / sizes, instead of a per-image loop of masked_select / interpolate calls.
There is no CPU path: host tensors are rejected like everywhere else in the package.
"""
import torch
from torch import nn

from .. import hip as H


def _sizes_to_host(t):
    """[B, 2] integer sizes as python ints (ONE device->host copy if the caller handed a device tensor, as the reference's
    .tolist() calls do per image)."""
    return [[int(a), int(b)] for a, b in torch.as_tensor(t).tolist()]


class PostProcessVGMultiPhrase(nn.Module):
    """outputs['pred_boxes'] [B, P, K, 4] cxcywh + outputs['phrase_mask'] [B, P(*K)] -> [{'boxes': xyxy [n_valid_b, 4]}]."""

    @torch.no_grad()
    def forward(self, outputs, target_sizes, scale_to_original_shape=False):
        boxes = outputs["pred_boxes"]
        B, P, K, _ = boxes.shape
        assert B == len(target_sizes) and target_sizes.shape[1] == 2
        valid = outputs["phrase_mask"].reshape(B, P, K).to(torch.uint8).contiguous()
        sizes = None
        if scale_to_original_shape:          # boxes * [img_w, img_h, img_w, img_h]: the integer sizes enter as fp32, as torch promotes them
            sizes = target_sizes.to(device=boxes.device, dtype=torch.float32).contiguous()
        xyxy, counts = H.box_postprocess(boxes.to(torch.float32).contiguous(), valid, sizes)
        return [{"boxes": xyxy[b, :n]} for b, n in enumerate(counts.tolist())]


class PostProcessSegm(nn.Module):
    """outputs['pred_masks'] [B, Q, 1, h, w] (or [B, Q, h, w]) mask logits -> results[i]['masks'] bool [Q, 1, img_h, img_w]
    (decision on the logits resized to the padded frame, cropped to the image's own size) and results[i]['masks_origin']
    uint8 [Q, 1, orig_h, orig_w] (nearest resize of it to the original image size)."""

    def __init__(self, threshold=0.5):
        super().__init__()
        self.threshold = threshold

    @torch.no_grad()
    def forward(self, results, outputs, orig_target_sizes, max_target_sizes):
        assert len(orig_target_sizes) == len(max_target_sizes)
        pm = outputs["pred_masks"]
        if pm.dim() == 5:
            pm = pm.squeeze(2)
        pm = pm.to(torch.float32).contiguous()
        B, Q = pm.shape[:2]
        dev = pm.device
        sizes, orig = _sizes_to_host(max_target_sizes), _sizes_to_host(orig_target_sizes)
        max_h, max_w = max(s[0] for s in sizes), max(s[1] for s in sizes)
        off = [0]
        for oh, ow in orig:
            off.append(off[-1] + Q * oh * ow)
        table = torch.tensor([v for s in sizes for v in s] + [v for s in orig for v in s], dtype=torch.int32).to(dev)
        masks, packed = H.mask_postprocess(pm, table[:2 * B].view(B, 2), (max_h, max_w), self.threshold,
                                           orig_i32=table[2 * B:].view(B, 2), origin_off=torch.tensor(off, dtype=torch.int64).to(dev),
                                           origin_total=off[-1], max_origin=max(oh * ow for oh, ow in orig))
        masks = masks.view(torch.bool)
        for i, ((ih, iw), (oh, ow)) in enumerate(zip(sizes, orig)):
            results[i]["masks"] = masks[i, :, :ih, :iw].unsqueeze(1)
            results[i]["masks_origin"] = packed[off[i]:off[i + 1]].view(Q, 1, oh, ow)
        return results
