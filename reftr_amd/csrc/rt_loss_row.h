// One (layer, image, phrase, query) row of CriterionVGMultiPhrase's box loss (models/criterion.py:113-153,
// util/box_ops.py:17-69): sigmoid, L1, diag GIoU and the gradient of the weighted total with respect to the four
// pre-sigmoid logits.  Shared by rt_box_loss (rt_loss.hip) and the fused head launch rt_head_loss (rt_qregion.hip) so that both
// evaluate the same expression tree on a row.
#pragma once
#include "rt_common.h"

// lg[4] logits, tg[4] target (cx, cy, w, h); wb / wg the layer's loss weights, nb = max(num_boxes, 1) * K.
// Adds |s - t| and (1 - giou) to l1_sum / gi_sum; g_out[4] = d total / d logits.
__device__ __forceinline__ void rt_box_loss_row(const float* lg, const float* tg, float wb, float wg, float nb,
                                                float& l1_sum, float& gi_sum, float* g_out) {
    float s[4], t[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { s[c] = 1.f / (1.f + __expf(-lg[c])); t[c] = tg[c]; }
    float g[4];      // d(total)/d(sigmoid output)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float d = s[c] - t[c];
        l1_sum += fabsf(d);
        g[c] = wb * ((d > 0.f) ? 1.f : ((d < 0.f) ? -1.f : 0.f)) / nb;
    }
    // GIoU on xyxy
    const float x0 = s[0] - 0.5f * s[2], y0 = s[1] - 0.5f * s[3], x1 = s[0] + 0.5f * s[2], y1 = s[1] + 0.5f * s[3];
    const float X0 = t[0] - 0.5f * t[2], Y0 = t[1] - 0.5f * t[3], X1 = t[0] + 0.5f * t[2], Y1 = t[1] + 0.5f * t[3];
    const float ap = (x1 - x0) * (y1 - y0), at = (X1 - X0) * (Y1 - Y0);
    const float iw = fmaxf(fminf(x1, X1) - fmaxf(x0, X0), 0.f), ih = fmaxf(fminf(y1, Y1) - fmaxf(y0, Y0), 0.f);
    const float inter = iw * ih;
    const float uni = ap + at - inter;
    const float iou = inter / uni;
    const float ew = fmaxf(fmaxf(x1, X1) - fminf(x0, X0), 0.f), eh = fmaxf(fmaxf(y1, Y1) - fminf(y0, Y0), 0.f);
    const float C = ew * eh;
    const float giou = iou - (C - uni) / C;
    gi_sum += 1.f - giou;
    // gradients of inter / area / C wrt (x0, y0, x1, y1)
    const float di[4] = {(x0 > X0 && iw > 0.f) ? -ih : 0.f, (y0 > Y0 && ih > 0.f) ? -iw : 0.f,
                         (x1 < X1 && iw > 0.f) ? ih : 0.f, (y1 < Y1 && ih > 0.f) ? iw : 0.f};
    const float da[4] = {-(y1 - y0), -(x1 - x0), (y1 - y0), (x1 - x0)};
    const float dC[4] = {(x0 < X0 && ew > 0.f) ? -eh : 0.f, (y0 < Y0 && eh > 0.f) ? -ew : 0.f,
                         (x1 > X1 && ew > 0.f) ? eh : 0.f, (y1 > Y1 && eh > 0.f) ? ew : 0.f};
    float dg[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float du = da[c] - di[c];
        const float diou = (di[c] * uni - inter * du) / (uni * uni);
        dg[c] = diou + (du * C - uni * dC[c]) / (C * C);     // d giou / d coord
    }
    const float sc = -wg / nb;                          // loss = (1 - giou) / nb
    g[0] += sc * (dg[0] + dg[2]);
    g[1] += sc * (dg[1] + dg[3]);
    g[2] += sc * 0.5f * (dg[2] - dg[0]);
    g[3] += sc * 0.5f * (dg[3] - dg[1]);
#pragma unroll
    for (int c = 0; c < 4; ++c) g_out[c] = g[c] * s[c] * (1.f - s[c]);
}
