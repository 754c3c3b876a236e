"""GPU: the evaluation post-processors (csrc/rt_post.hip through the C ABI) against the reference's golden vectors
(exact) and against the oracle's restatement at evaluation sizes.

  PostProcessVGMultiPhrase (models/post_process.py:45-83): selection order is integer logic, the box arithmetic is issued in
  the reference's order un-fused -> bit-identical floats.
  PostProcessSegm (models/reftr_segmentation.py:282-302): outputs are decisions (bool / uint8) of an fp32 bilinear value; the
  kernel uses the reference kernel's index / weight rule and association, so decisions are identical except where the
  interpolated logit is within an ulp of the threshold's pre-image (fused vs un-fused multiply-add on the host) -- none on
  the golden vectors, and at most a 1e-6 fraction is tolerated on the 3.3 M-pixel random case (measured: 0).
"""
import os

import numpy as np
import pytest
import torch

from oracle import reftr_oracle as O

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_box_postprocess_matches_reference_golden_exactly(hip):
    from reftr_amd.models.post_process import PostProcessVGMultiPhrase
    g = np.load(os.path.join(GOLD, "postprocess.npz"))
    out = {"pred_boxes": torch.from_numpy(g["pred"]).cuda(), "phrase_mask": torch.from_numpy(g["mask"]).cuda()}
    sizes = torch.from_numpy(g["sizes"]).cuda()
    res = PostProcessVGMultiPhrase()(out, sizes, scale_to_original_shape=True)
    for i, r in enumerate(res):
        assert r["boxes"].dtype == torch.float32
        assert torch.equal(r["boxes"].cpu(), torch.from_numpy(g[f"boxes{i}"])), i           # bit-identical
    # unscaled form against the oracle (the evaluate loop's IoU inputs, engine_vg.py:127-140)
    res = PostProcessVGMultiPhrase()(out, sizes)
    ref = O.postprocess_boxes(torch.from_numpy(g["pred"]), torch.from_numpy(g["mask"]), torch.from_numpy(g["sizes"]))
    for r, b in zip(res, ref):
        assert torch.equal(r["boxes"].cpu(), b)


def test_box_postprocess_ragged_multi_query_integer_sizes(hip):
    """16 phrase slots, K = 2 predictions per phrase, ragged validity incl. an image without any valid phrase; int64 sizes."""
    from reftr_amd.models.post_process import PostProcessVGMultiPhrase
    g = torch.Generator().manual_seed(3)
    B, P, K = 5, 16, 2
    pred = torch.rand(B, P, K, 4, generator=g)
    valid = torch.rand(B, P, generator=g) < 0.5
    valid[2] = False
    valid[4] = True
    mask = valid[:, :, None].expand(B, P, K).reshape(B, P * K)
    sizes = torch.tensor([[480, 640], [333, 500], [640, 427], [1, 1], [799, 1333]])
    res = PostProcessVGMultiPhrase()({"pred_boxes": pred.cuda(), "phrase_mask": mask.cuda()}, sizes.cuda(), True)
    ref = O.postprocess_boxes(pred, mask, sizes, True)
    for i, (r, b) in enumerate(zip(res, ref)):
        assert r["boxes"].shape == b.shape == (int(valid[i].sum()), 4)
        assert torch.equal(r["boxes"].cpu(), b), i


def test_box_postprocess_more_than_64_phrase_slots(hip):
    """The reference handles any number of phrase slots (post_process.py:62-70); the kernel walks them 64 at a time with an
    ordered running rank (P > 64 used to be rejected).  P = 150, ragged validity, exact against the oracle."""
    from reftr_amd.models.post_process import PostProcessVGMultiPhrase
    g = torch.Generator().manual_seed(11)
    B, P, K = 3, 150, 1
    pred = torch.rand(B, P, K, 4, generator=g)
    valid = torch.rand(B, P, generator=g) < 0.4
    valid[1] = True
    mask = valid[:, :, None].expand(B, P, K).reshape(B, P * K)
    sizes = torch.tensor([[480, 640], [333, 500], [640, 427]])
    for scale in (False, True):
        res = PostProcessVGMultiPhrase()({"pred_boxes": pred.cuda(), "phrase_mask": mask.cuda()}, sizes.cuda(), scale)
        ref = O.postprocess_boxes(pred, mask, sizes, scale)
        for i, (r, b) in enumerate(zip(res, ref)):
            assert r["boxes"].shape == b.shape == (int(valid[i].sum()), 4)
            assert torch.equal(r["boxes"].cpu(), b), (scale, i)


def test_mask_postprocess_matches_reference_golden_exactly(hip):
    from reftr_amd.models.post_process import PostProcessSegm
    g = np.load(os.path.join(GOLD, "seg_single.npz"))
    res = PostProcessSegm()([{} for _ in range(2)], {"pred_masks": torch.from_numpy(g["pred_masks"]).cuda()},
                            torch.from_numpy(g["post_orig"]), torch.from_numpy(g["post_sizes"]))
    for i in range(2):
        want, want_o = torch.from_numpy(g[f"post_masks{i}"]), torch.from_numpy(g[f"post_masks_origin{i}"])
        assert res[i]["masks"].dtype == torch.bool and res[i]["masks_origin"].dtype == torch.uint8
        assert res[i]["masks"].shape == want.shape and res[i]["masks_origin"].shape == want_o.shape
        assert torch.equal(res[i]["masks"].cpu(), want)
        assert torch.equal(res[i]["masks_origin"].cpu(), want_o)


def test_mask_postprocess_eval_sizes_vs_oracle(hip):
    """configs[3] evaluation shapes: 8 images, stride-4 logits 160x160 -> 640x640 frame, ragged own sizes; original sizes that
    hit every branch of torch's nearest index rule (equal, exactly 2x, up, down)."""
    from reftr_amd.models.post_process import PostProcessSegm
    g = torch.Generator().manual_seed(11)
    pred = torch.randn(8, 1, 1, 160, 160, generator=g) * 2.0
    sizes = torch.tensor([[640, 640], [640, 480], [480, 640], [427, 640], [640, 427], [333, 500], [512, 512], [97, 131]])
    orig = torch.tensor([[640, 640], [1280, 960], [375, 500], [427, 640], [1000, 667], [333, 500], [300, 300], [194, 262]])
    res = PostProcessSegm()([{} for _ in range(8)], {"pred_masks": pred.cuda()}, orig, sizes)
    ref = O.postprocess_segm(pred, orig, sizes)
    bad = tot = 0
    for r, (m, mo) in zip(res, ref):
        assert r["masks"].shape == m.shape and r["masks_origin"].shape == mo.shape
        bad += int((r["masks"].cpu() != m).sum()) + int((r["masks_origin"].cpu() != mo).sum())
        tot += m.numel() + mo.numel()
        assert 0.2 < float(m.float().mean()) < 0.8                       # a non-degenerate case
    print(f"\n[mask postprocess] {bad} of {tot} decisions differ from the torch-CPU restatement")
    assert bad <= 1e-6 * tot, (bad, tot)
