"""GPU: edge cases of the path against the CPU oracle (same formula weights, eval mode): batch of one, image sizes that are
not multiples of the stride (every stage rounds (n-1)//2+1), an image that is almost all padding, the longest sentence the
model accepts (max_lang_seq = 128, models/reftr.py:81) and sentences of minimum length, a multi-phrase batch in which one
image has NO valid phrase, and the all-empty batch (num_boxes clamps to 1, every loss is exactly 0).

An image without a valid phrase is outside the reference's domain: its decoder self-attention runs over a fully masked key
set (`tgt_key_padding_mask` all True, transformer.py:231-252), the reference's own logits for that image are NaN and so are
191 of its gradient tensors (checked with the oracle = the reference's arithmetic).  For that input only the forward quantities
that the reference defines are compared: the validity mask (exact), the boxes of the valid queries and the losses."""
import numpy as np
import pytest
import torch

from oracle import reftr_oracle as O
from oracle.synth import make_inputs
from test_model_gpu import build, rel, to_cuda

pytestmark = pytest.mark.gpu


def compare(samples, targets, tol_box=3.1e-3, tol_loss=3.4e-3, backward=True):      # 1.5 x the worst measured (2.05e-3 / 2.22e-3)
    model, crit, P, ocfg = build(small=True)
    model.eval()
    s, tg = to_cuda(samples, targets)
    out = model(s)
    ld = crit(out, tg)
    ref = O.reftr_forward(P, samples, ocfg, train=False, q=False)
    rl = O.criterion(ref, targets)
    boxes = torch.cat([torch.stack([a["pred_boxes"] for a in out["aux_outputs"]]), out["pred_boxes"][None]])
    valid = ref["phrase_mask"].reshape(-1)
    assert np.array_equal(out["phrase_mask"].reshape(-1).cpu().numpy(), valid.numpy())          # bool: exact
    got = boxes.detach().float().cpu().reshape(boxes.shape[0], -1, 4)[:, valid]
    want = ref["logits"].sigmoid().reshape(boxes.shape[0], -1, 4)[:, valid]
    rb = rel(got, want) if valid.any() else 0.0
    rl_max = max(abs(float(v) - float(rl[k])) / max(1.0, abs(float(rl[k]))) for k, v in ld.items())
    print(f"\n[edge case] boxes rel {rb:.2e} (tol {tol_box:.1e})  worst loss rel {rl_max:.2e} (tol {tol_loss:.1e})")
    assert rb < tol_box, rb
    for k, v in ld.items():
        assert abs(float(v) - float(rl[k])) < tol_loss * max(1.0, abs(float(rl[k]))), (k, float(v), float(rl[k]))
    if backward:
        total = sum(ld[k] * crit.weight_dict[k] for k in ld if k in crit.weight_dict)
        model.store.flat_g.zero_()
        total.backward()
        torch.cuda.synchronize()
        assert torch.isfinite(model.store.flat_g).all()
    return model, ld


@pytest.mark.parametrize("B,H,W,L", [(1, 75, 101, 5), (3, 33, 64, 12), (2, 130, 97, 128), (2, 64, 64, 4)])
def test_odd_shapes_single_phrase(hip, B, H, W, L):
    samples, targets = make_inputs(f"edge_{B}_{H}_{W}_{L}", B=B, H=H, W=W, L=L)
    compare(samples, targets)


def test_image_that_is_almost_all_padding(hip):
    samples, targets = make_inputs("edge_pad", B=2, H=96, W=128, L=12)
    samples["img_mask"][1] = True
    samples["img_mask"][1, :20, :20] = False            # 20 x 20 valid pixels: a single stride-32 cell
    samples["img"][1] = samples["img"][1] * (~samples["img_mask"][1]).float()
    compare(samples, targets)


def test_multi_phrase_with_an_image_without_valid_phrases(hip):
    samples, targets = make_inputs("edge_multi", B=3, H=96, W=128, L=14, n_phrase=2)   # image 2 has 0 valid phrases
    assert targets[2]["boxes"].shape[0] == 0 and int(samples["phrase_mask"][2, :, 2].sum()) == 0
    model, ld = compare(samples, targets, backward=False)
    assert float(ld["loss_bbox"]) > 0


def test_batch_without_any_box(hip):
    samples, targets = make_inputs("edge_none", B=2, H=64, W=96, L=8, n_phrase=1)
    # make every phrase the padding phrase "[CLS] [SEP]" and drop the boxes
    samples["phrase"][:, :, 2:] = 0; samples["phrase"][:, :, 1] = 102; samples["phrase_mask"][:, :, 2:] = 0
    samples["phrase_pos_l"][:] = 0; samples["phrase_pos_r"][:] = 1
    targets = [{"boxes": torch.zeros(0, 4), "labels": torch.zeros(0, dtype=torch.long)} for _ in range(2)]
    model, ld = compare(samples, targets, backward=False)      # (every query masked: the same out-of-domain NaN as above)
    assert all(float(v) == 0.0 for v in ld.values())


def test_box_count_mismatch_poisons_the_loss(hip):
    """The reference asserts that every image has as many valid phrases as target boxes (criterion.py:127).  The device
    criterion cannot raise without a host sync: it must not read another image's targets and must make the mismatch loud
    -- the losses come back NaN, which stops the training loop exactly like the assert would (engine_vg.py:55-58)."""
    from reftr_amd import hip as H
    logits = torch.zeros(2, 2, 3, 1, 4, device="cuda")
    valid = torch.tensor([[1, 1, 0], [1, 0, 0]], dtype=torch.uint8, device="cuda")
    boxes = torch.tensor([[0.5, 0.5, 0.2, 0.2]] * 3, device="cuda")
    nb = torch.tensor([3.0], device="cuda")
    ok, _, _ = H.box_loss(logits, valid, boxes, torch.tensor([0, 2, 3], dtype=torch.int32, device="cuda"), nb, want_grad=False)
    assert torch.isfinite(ok).all()
    bad, _, _ = H.box_loss(logits, valid, boxes, torch.tensor([0, 1, 3], dtype=torch.int32, device="cuda"), nb, want_grad=False)
    assert torch.isnan(bad[:, 0]).all()                                # image 0: 2 valid phrases, 1 box; image 1: 1 phrase, 2 boxes


@pytest.mark.parametrize("B,H,W,L", [(1, 75, 101, 6), (3, 64, 130, 9)])
def test_odd_shapes_refer_segmentation(hip, B, H, W, L):
    """RefTRSeg on sizes where the FPN levels do not halve evenly (19x26 -> 10x13 -> 5x7 -> 3x4 for 75x101): nearest upsampling
    to the next level's size, bilinear loss upsampling to the padded target size."""
    from test_seg_gpu import build_seg
    samples, targets = make_inputs(f"edge_seg_{B}_{H}_{W}", B=B, H=H, W=W, L=L)
    g = torch.Generator().manual_seed(3)
    targets = [dict(t, masks=(torch.rand(1, H, W, generator=g) < 0.3) & ~samples["img_mask"][i][None]) for i, t in enumerate(targets)]
    model, crit, P, ocfg = build_seg()
    model.eval()
    s, tg = to_cuda(samples, targets)
    out = model(s)
    ld = crit(out, tg)
    ref = O.reftr_forward(P, samples, ocfg, train=False, q=False)
    rl = O.criterion(ref, targets)
    assert out["pred_masks"].shape == ref["pred_masks"].shape
    assert rel(out["pred_masks"], ref["pred_masks"]) < 3e-2 and rel(out["mask_att"], ref["mask_att"]) < 3e-2
    assert rel(out["pred_boxes"], ref["logits"][-1].sigmoid().reshape(out["pred_boxes"].shape)) < 8e-3
    for k in ("loss_mask", "loss_dice", "loss_bbox", "loss_giou"):
        assert abs(float(ld[k]) - float(rl[k])) < 1.5e-2 * max(abs(float(rl[k])), 0.1), (k, float(ld[k]), float(rl[k]))
    total = sum(ld[k] * crit.weight_dict[k] for k in ld if k in crit.weight_dict)
    model.store.flat_g.zero_()
    total.backward()
    torch.cuda.synchronize()
    assert torch.isfinite(model.store.flat_g).all() and float(model.store.G["mask_head.lay1.weight"].abs().max()) > 0
