// Box loss of CriterionVGMultiPhrase (models/criterion.py:113-202, util/box_ops.py:17-69) for ALL decoder
// layers in one launch: valid-phrase selection (exact, integer), sigmoid, L1, diag GIoU, their sums
// normalised by num_boxes * k, and the gradient of the weighted total w.r.t. the pre-sigmoid logits.
#include "rt_common.h"
#include "rt_loss_row.h"

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
            float g[4];
            rt_box_loss_row(lg, tg, wb, wg, nb, l1_sum, gi_sum, g);
            if (dl) {
#pragma unroll
                for (int c = 0; c < 4; ++c) dl[c] = g[c];
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
