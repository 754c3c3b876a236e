"""Filter taps of Pillow's 8-bit bilinear resampler, computed on the host exactly as Pillow computes them (Resample.c:
precompute_coeffs with the triangle filter of support 1, then normalize_coeffs_8bpc to 22-bit fixed point).  Python floats
are IEEE doubles and the operation order is Pillow's, so the integer taps — and therefore the device kernel's output bytes —
equal Pillow's.  Cached per (input length, output length)."""
import functools

import torch

PRECISION_BITS = 32 - 8 - 2


@functools.lru_cache(maxsize=512)
def taps(in_size, out_size):
    """(bounds int32 [out, 2] = (first input index, number of taps), coeffs int32 [out, ksize])."""
    scale = in_size / out_size
    fscale = scale if scale > 1.0 else 1.0
    support = 1.0 * fscale
    ksize = int(-(-support // 1)) * 2 + 1
    bounds = torch.zeros(out_size, 2, dtype=torch.int32)
    coeffs = torch.zeros(out_size, ksize, dtype=torch.int32)
    ss = 1.0 / fscale
    one = 1 << PRECISION_BITS
    for xx in range(out_size):
        center = (xx + 0.5) * scale
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
        ww = 0.0
        for v in w:
            ww += v
        bounds[xx, 0] = xmin; bounds[xx, 1] = n
        for x, v in enumerate(w):
            if ww != 0.0:
                v = v / ww
            coeffs[xx, x] = int(-0.5 + v * one) if v < 0 else int(0.5 + v * one)
    return bounds, coeffs


def size_with_aspect_ratio(w, h, size, max_size=None):
    """datasets/transforms.py:84-104: (oh, ow) of the resized image."""
    if max_size is not None:
        mn, mx = float(min(w, h)), float(max(w, h))
        if mx / mn * size > max_size:
            size = int(round(max_size * mn / mx))
    if (w <= h and w == size) or (h <= w and h == size):
        return h, w
    if w < h:
        return int(size * h / w), size
    return size, int(size * w / h)
