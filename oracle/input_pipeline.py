"""CPU restatement of the reference's host input pipeline: the deterministic part of
datasets/transforms.py (resize :81-137 incl. PIL's antialiased bilinear resampling of uint8 images, ToTensor :233-235,
Normalize :247-263) and the batch padding of util/collate_fn.py:24-41.

TEST INFRASTRUCTURE (oracle/).  `pil_bilinear_resize_u8` restates Pillow's ImagingResample 8-bit path (Resample.c:
precompute_coeffs / normalize_coeffs_8bpc / ImagingResampleHorizontal_8bpc / Vertical_8bpc; Pillow is a third-party
dependency of the reference through torchvision.transforms.functional.resize on PIL images, installed here as 12.2.0) in
integer arithmetic; tests/test_input_pipeline.py pins it bit-exactly against PIL.Image.resize itself.
"""
import math

import numpy as np
import torch

PRECISION_BITS = 32 - 8 - 2
MEAN = (0.485, 0.456, 0.406)      # datasets/refer_resc.py:103
STD = (0.229, 0.224, 0.225)


def get_size_with_aspect_ratio(image_size, size, max_size=None):
    """datasets/transforms.py:84-104; image_size = (w, h); returns (oh, ow)."""
    w, h = image_size
    if max_size is not None:
        mn, mx = float(min(w, h)), float(max(w, h))
        if mx / mn * size > max_size:
            size = int(round(max_size * mn / mx))
    if (w <= h and w == size) or (h <= w and h == size):
        return (h, w)
    if w < h:
        ow = size; oh = int(size * h / w)
    else:
        oh = size; ow = int(size * w / h)
    return (oh, ow)


def resample_coeffs(in_size, out_size):
    """precompute_coeffs + normalize_coeffs_8bpc for the bilinear (triangle, support 1) filter: per output index
    (xmin, n taps, int32 coefficients scaled by 2**22)."""
    scale = in_size / out_size
    fscale = max(scale, 1.0)
    support = 1.0 * fscale
    out = []
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        ss = 1.0 / fscale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        n = xmax - xmin
        w = []
        for x in range(n):
            a = abs((x + xmin - center + 0.5) * ss)
            w.append(1.0 - a if a < 1.0 else 0.0)
        ww = sum(w)
        if ww != 0.0:
            w = [v / ww for v in w]
        k = [int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS)) for v in w]
        out.append((xmin, n, k))
    return out


def _clip8(v):
    return np.clip(v >> PRECISION_BITS, 0, 255).astype(np.uint8)


def pil_bilinear_resize_u8(img, oh, ow):
    """img uint8 [H, W, C] -> uint8 [oh, ow, C], bit-exact with PIL.Image.resize((ow, oh), BILINEAR) (horizontal pass
    first, 8-bit intermediate, each pass skipped when the size does not change)."""
    H, W, C = img.shape
    cur = img
    if ow != W:
        co = resample_coeffs(W, ow)
        tmp = np.empty((H, ow, C), dtype=np.uint8)
        src = cur.astype(np.int64)
        for xx, (xmin, n, k) in enumerate(co):
            acc = np.full((H, C), 1 << (PRECISION_BITS - 1), dtype=np.int64)
            for j in range(n):
                acc += src[:, xmin + j, :] * k[j]
            tmp[:, xx, :] = _clip8(acc)
        cur = tmp
    if oh != H:
        co = resample_coeffs(H, oh)
        tmp = np.empty((oh, cur.shape[1], C), dtype=np.uint8)
        src = cur.astype(np.int64)
        for yy, (ymin, n, k) in enumerate(co):
            acc = np.full((cur.shape[1], C), 1 << (PRECISION_BITS - 1), dtype=np.int64)
            for j in range(n):
                acc += src[ymin + j, :, :] * k[j]
            tmp[yy] = _clip8(acc)
        cur = tmp
    return cur


def resize_target(target, in_hw, out_hw):
    """datasets/transforms.py:118-137 (boxes xyxy in pixels, masks nearest)."""
    (H, W), (oh, ow) = in_hw, out_hw
    rw, rh = float(ow) / float(W), float(oh) / float(H)
    t = dict(target)
    if "boxes" in t:
        t["boxes"] = t["boxes"] * torch.as_tensor([rw, rh, rw, rh])
    if "area" in t:
        t["area"] = t["area"] * (rw * rh)
    t["size"] = torch.tensor([oh, ow])
    if "masks" in t:
        t["masks"] = torch.nn.functional.interpolate(t["masks"][:, None].float(), (oh, ow), mode="nearest")[:, 0] > 0.5
    return t


def to_tensor_normalize(img_u8, target=None, mean=MEAN, std=STD):
    """ToTensor + Normalize (datasets/transforms.py:233-235, 247-263): uint8 HWC -> fp32 CHW, (x/255 - mean)/std; boxes
    xyxy pixels -> cxcywh normalised by the (resized) image size."""
    x = torch.from_numpy(np.ascontiguousarray(img_u8)).permute(2, 0, 1).float().div(255)
    x = (x - torch.tensor(mean).view(-1, 1, 1)) / torch.tensor(std).view(-1, 1, 1)
    if target is None:
        return x, None
    t = dict(target)
    h, w = x.shape[-2:]
    if "boxes" in t:
        b = t["boxes"]
        x0, y0, x1, y1 = b.unbind(-1)
        b = torch.stack([(x0 + x1) / 2, (y0 + y1) / 2, x1 - x0, y1 - y0], dim=-1)
        t["boxes"] = b / torch.tensor([w, h, w, h], dtype=torch.float32)
    return x, t


def collate(images):
    """nested_tensor_from_tensor_list (util/collate_fn.py:24-41): zero-pad to the batch maximum, mask True = padding."""
    hm = max(i.shape[1] for i in images); wm = max(i.shape[2] for i in images)
    out = torch.zeros(len(images), images[0].shape[0], hm, wm)
    mask = torch.ones(len(images), hm, wm, dtype=torch.bool)
    for i, im in enumerate(images):
        out[i, :, :im.shape[1], :im.shape[2]] = im
        mask[i, :im.shape[1], :im.shape[2]] = False
    return out, mask


def preprocess_batch(images_u8, targets, size, max_size):
    """The test-time transform chain of datasets/refer_resc.py:100-121 + the collate: list of uint8 HWC arrays ->
    (batch fp32 [B,3,H,W], mask bool [B,H,W], targets)."""
    xs, ts = [], []
    for img, tg in zip(images_u8, targets):
        H, W = img.shape[:2]
        oh, ow = get_size_with_aspect_ratio((W, H), size, max_size)
        r = pil_bilinear_resize_u8(img, oh, ow)
        t = resize_target(tg, (H, W), (oh, ow)) if tg is not None else None
        x, t = to_tensor_normalize(r, t)
        xs.append(x); ts.append(t)
    b, m = collate(xs)
    return b, m, ts
