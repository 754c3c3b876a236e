// Box loss of CriterionVGMultiPhrase (models/criterion.py:113-202, util/box_ops.py:17-69) for ALL decoder
// layers in one launch: valid-phrase selection (exact, integer), sigmoid, L1, diag GIoU, their sums
// normalised by num_boxes * k, and the gradient of the weighted total w.r.t. the pre-sigmoid logits.
#include "rt_common.h"

namespace {

__global__ __launch_bounds__(256) void box_loss_kernel(const rt_box_loss_desc p) {
    const int total = p.NL * p.B * p.P * p.K;
    const float nb = fmaxf(p.num_boxes[0], 1.f) * (float)p.K;
    float l1_sum = 0.f, gi_sum = 0.f;
    int layer = -1;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx < total) {
        const int k = idx % p.K;
        const int ph = (idx / p.K) % p.P;
        const int b = (idx / (p.K * p.P)) % p.B;
        const int l = idx / (p.K * p.P * p.B);
        const float* lg = p.logits + (size_t)idx * 4;
        float* dl = p.dlogits ? p.dlogits + (size_t)idx * 4 : nullptr;
        const bool valid = p.valid[(size_t)b * p.P * p.K + ph * p.K + k] != 0;
        if (!valid) {
            if (dl) { dl[0] = 0.f; dl[1] = 0.f; dl[2] = 0.f; dl[3] = 0.f; }
        } else {
            layer = l;
            const float wb = p.weights ? p.weights[l * 2] : p.w_bbox;
            const float wg = p.weights ? p.weights[l * 2 + 1] : p.w_giou;
            int rank = 0, nvalid = 0;      // masked_select keeps phrase order (criterion.py:126)
            for (int q = 0; q < p.P; ++q) {
                const int v = p.valid[(size_t)b * p.P * p.K + q * p.K] ? 1 : 0;
                nvalid += v;
                if (q < ph) rank += v;
            }
            // the reference asserts pred_i.shape[0] == target_i.shape[0] per image (criterion.py:127).  A device kernel
            // cannot raise: a mismatch poisons the losses with NaN (the training loop stops on a non-finite loss,
            // engine_vg.py:55-58) and never reads outside this image's target rows.
            const int count = p.tgt_off[b + 1] - p.tgt_off[b];
            const bool mismatch = nvalid != count;
            const float* tg = p.targets + ((size_t)p.tgt_off[b] + (mismatch ? 0 : rank)) * 4;
            if (mismatch) { l1_sum = __builtin_nanf(""); tg = lg; }
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
            if (dl) {
#pragma unroll
                for (int c = 0; c < 4; ++c) dl[c] = g[c] * s[c] * (1.f - s[c]);
            }
        }
    }
    if (layer >= 0) {
        const float wb = p.weights ? p.weights[layer * 2] : p.w_bbox;
        const float wg = p.weights ? p.weights[layer * 2 + 1] : p.w_giou;
        atomicAdd(p.losses + layer * 2, l1_sum / nb);
        atomicAdd(p.losses + layer * 2 + 1, gi_sum / nb);
        atomicAdd(p.total, (wb * l1_sum + wg * gi_sum) / nb);
    }
}

}  // namespace

extern "C" int rt_box_loss(const rt_box_loss_desc* d, rt_stream_t stream) {
    if (!d || !d->logits || !d->valid || !d->targets || !d->tgt_off || !d->num_boxes || !d->losses || !d->total)
        return RT_ERR_BADARG;
    hipStream_t s = (hipStream_t)stream;
    // both outputs are accumulated with atomics; a caller that hands them over as ONE buffer ([NL][2] losses | total) pays one clear
    const bool joined = d->total == d->losses + 2 * (size_t)d->NL;
    hipError_t e = rt_zero_f32(d->losses, 2 * (size_t)d->NL + (joined ? 1 : 0), s);
    if (e != hipSuccess) return (int)e;
    if (!joined) {
        e = rt_zero_f32(d->total, 1, s);
        if (e != hipSuccess) return (int)e;
    }
    const int total = d->NL * d->B * d->P * d->K;
    hipLaunchKernelGGL(box_loss_kernel, dim3((total + 255) / 256), dim3(256), 0, s, *d);
    RT_CHECK_LAUNCH();
    return RT_OK;
}
