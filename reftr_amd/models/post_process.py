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
