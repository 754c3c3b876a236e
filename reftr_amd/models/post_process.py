"""PostProcessVGMultiPhrase — same contract as the reference (models/post_process.py:41-83): select the valid
phrases of every image (phrase order preserved), cxcywh -> xyxy, optionally scale to the original size."""
import torch
from torch import nn

from ..util import box_ops


class PostProcessVGMultiPhrase(nn.Module):
    @torch.no_grad()
    def forward(self, outputs, target_sizes, scale_to_original_shape=False):
        out_bbox = outputs["pred_boxes"]
        bsz, num_phrase, k, _ = out_bbox.shape
        mask = outputs["phrase_mask"].view(bsz, num_phrase, k)
        assert bsz == len(target_sizes) and target_sizes.shape[1] == 2
        results = []
        for i in range(bsz):
            pred_i = out_bbox[i][mask[i]].view(-1, k, 4)
            boxes = box_ops.box_cxcywh_to_xyxy(pred_i[:, 0, :])
            if scale_to_original_shape:
                img_h, img_w = target_sizes[i:i + 1].unbind(1)
                boxes = boxes * torch.stack([img_w, img_h, img_w, img_h], dim=1)
            results.append({"boxes": boxes})
        return results


class PostProcessSegm(nn.Module):
    """models/reftr_segmentation.py:282-302 (eval only: bilinear resize -> sigmoid > threshold -> crop to the padded-free
    size -> nearest resize to the original size).  Plain torch ops: exact decisions on float inputs, not a hot path."""

    def __init__(self, threshold=0.5):
        super().__init__()
        self.threshold = threshold

    @torch.no_grad()
    def forward(self, results, outputs, orig_target_sizes, max_target_sizes):
        import torch.nn.functional as F
        assert len(orig_target_sizes) == len(max_target_sizes)
        max_h, max_w = max_target_sizes.max(0)[0].tolist()
        masks = outputs["pred_masks"].squeeze(2)
        masks = F.interpolate(masks, size=(max_h, max_w), mode="bilinear", align_corners=False)
        masks = masks.sigmoid() > self.threshold
        for i, (cur, t, tt) in enumerate(zip(masks, max_target_sizes, orig_target_sizes)):
            img_h, img_w = int(t[0]), int(t[1])
            results[i]["masks"] = cur[:, :img_h, :img_w].unsqueeze(1)
            results[i]["masks_origin"] = F.interpolate(results[i]["masks"].float(), size=tuple(tt.tolist()), mode="nearest").byte()
        return results
