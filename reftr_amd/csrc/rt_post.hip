// Evaluation-side post-processing of RefTR as two kernels:
//   rt_mask_postprocess  PostProcessSegm.forward (models/reftr_segmentation.py:282-302): bilinear resize of the mask logits
//                        to the padded frame (align_corners=False) -> sigmoid > threshold -> crop to the image's own size
//                        -> nearest resize to the original size.  Both outputs are produced from the logits in ONE pass
//                        (a masks_origin pixel recomputes the decision of the cropped-mask pixel it samples), so neither
//                        the fp32 up-sampled frame (B x max_h x max_w x 4 B) nor the float copy of the boolean mask that
//                        the reference materialises ever exists in HBM: traffic = logits once (L2-resident, 16x smaller
//                        than the frame) + 1 B per output pixel.
//   rt_box_postprocess   PostProcessVGMultiPhrase.forward (models/post_process.py:45-83): ordered selection of every
//                        image's valid phrases (exact integer logic, = torch.masked_select order), cxcywh -> xyxy, optional
//                        scaling to the image size.  fp32 operations are issued un-fused in the reference's order
//                        (x - 0.5*w, then * scale) so the boxes are bit-identical to its CPU result.
// HBM-bound byte kernels: one thread per output pixel / box, coalesced 1-B stores along the row.
#include "rt_common.h"

namespace {

// torch's index / weight rule for mode='bilinear', align_corners=False (ATen UpSample.h: area_pixel_compute_source_index,
// guard_index_and_lambda): src = max(scale * (dst + 0.5) - 0.5, 0), i0 = min(int(src), in - 1), lambda1 = clamp(src - i0, 0, 1)
__device__ __forceinline__ void bilinear_axis(int dst, float scale, int in, int& i0, int& i1, float& l0, float& l1) {
    float src = __fsub_rn(__fmul_rn(scale, __fadd_rn((float)dst, 0.5f)), 0.5f);
    if (src < 0.f) src = 0.f;
    i0 = min((int)floorf(src), in - 1);
    l1 = fminf(fmaxf(__fsub_rn(src, (float)i0), 0.f), 1.f);
    i1 = i0 + (i0 < in - 1 ? 1 : 0);
    l0 = __fsub_rn(1.f, l1);
}

// decision of frame pixel (y, x): sigmoid(bilinear(logits)) > threshold
__device__ __forceinline__ bool mask_decision(const float* __restrict__ lg, int h, int w, int y, int x, float sh, float sw, float thr) {
    int y0, y1, x0, x1; float ly0, ly1, lx0, lx1;
    bilinear_axis(y, sh, h, y0, y1, ly0, ly1);
    bilinear_axis(x, sw, w, x0, x1, lx0, lx1);
    const float a = lg[y0 * w + x0], b = lg[y0 * w + x1], c = lg[y1 * w + x0], d = lg[y1 * w + x1];
    // ly0 * (lx0 * a + lx1 * b) + ly1 * (lx0 * c + lx1 * d), the reference kernel's association
    const float top = __fadd_rn(__fmul_rn(lx0, a), __fmul_rn(lx1, b));
    const float bot = __fadd_rn(__fmul_rn(lx0, c), __fmul_rn(lx1, d));
    const float v = __fadd_rn(__fmul_rn(ly0, top), __fmul_rn(ly1, bot));
    const float s = 1.f / (1.f + expf(-v));
    return s > thr;
}

// torch 'nearest' (legacy) source index: out == in -> dst; out == 2 in -> dst >> 1; else min(int(floorf(dst * (float)in/out)), in - 1)
__device__ __forceinline__ int nearest_src(int dst, int in, int out) {
    if (out == in) return dst;
    if (out == 2 * in) return dst >> 1;
    const float scale = (float)in / (float)out;
    return min((int)floorf(__fmul_rn((float)dst, scale)), in - 1);
}

__global__ __launch_bounds__(256) void mask_postprocess_kernel(const rt_mask_post_desc p) {
    const int img = blockIdx.y / p.Q, q = blockIdx.y - img * p.Q;
    const float* lg = p.pred + (size_t)blockIdx.y * p.h * p.w;
    const float sh = (float)p.h / (float)p.max_h, sw = (float)p.w / (float)p.max_w;
    const int ih = p.sizes[img * 2], iw = p.sizes[img * 2 + 1];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < p.max_h * p.max_w) {                    // frame pixel: the cropped mask, zero outside the image's own size
        const int y = i / p.max_w, x = i - y * p.max_w;
        const bool in = y < ih && x < iw;
        p.masks[((size_t)blockIdx.y * p.max_h + y) * p.max_w + x] = (in && mask_decision(lg, p.h, p.w, y, x, sh, sw, p.threshold)) ? 1 : 0;
    }
    if (p.masks_origin) {
        const int oh = p.orig[img * 2], ow = p.orig[img * 2 + 1];
        if (i < oh * ow) {
            const int y = i / ow, x = i - y * ow;
            const int sy = nearest_src(y, ih, oh), sx = nearest_src(x, iw, ow);
            p.masks_origin[p.origin_off[img] + (size_t)q * oh * ow + i] = mask_decision(lg, p.h, p.w, sy, sx, sh, sw, p.threshold) ? 1 : 0;
        }
    }
}

__global__ __launch_bounds__(64) void box_postprocess_kernel(const rt_box_post_desc p) {
    const int b = blockIdx.x;
    const int lane = threadIdx.x;
    // phrase_mask is [B, P, K]; the reference masked_selects the (p, k) entries of [P, K, 4] whose mask is set and keeps
    // prediction 0 of each selected phrase (post_process.py:62-70).  The K entries of a phrase are equal by construction
    // (reftr_transformer.py:237-238), so a phrase is selected iff its entry 0 is set; rank = number of selected phrases in
    // front of it.  Any P: the wave walks the phrases 64 at a time, ranks inside a pass come from a ballot + population count
    // of the lower lanes (ordered, exact), `base` carries the count of the passes before.
    int base = 0;
    for (int j0 = 0; j0 < p.P; j0 += 64) {
        const int ph = j0 + lane;
        const bool mine = ph < p.P && p.valid[((size_t)b * p.P + ph) * p.K] != 0;
        const unsigned long long bal = __ballot(mine);
        const int rank = base + __popcll(bal & ((1ull << lane) - 1ull));
        base += __popcll(bal);
        if (!mine) continue;
        const float* s = p.boxes + ((size_t)(b * p.P + ph) * p.K) * 4;
        const float cx = s[0], cy = s[1], w = s[2], h = s[3];
        float x0 = __fsub_rn(cx, __fmul_rn(0.5f, w)), y0 = __fsub_rn(cy, __fmul_rn(0.5f, h));
        float x1 = __fadd_rn(cx, __fmul_rn(0.5f, w)), y1 = __fadd_rn(cy, __fmul_rn(0.5f, h));
        if (p.sizes) {                                  // scale_to_original_shape: boxes * [img_w, img_h, img_w, img_h]
            const float ih = p.sizes[b * 2], iw = p.sizes[b * 2 + 1];
            x0 = __fmul_rn(x0, iw); y0 = __fmul_rn(y0, ih); x1 = __fmul_rn(x1, iw); y1 = __fmul_rn(y1, ih);
        }
        float* o = p.out + ((size_t)b * p.P + rank) * 4;
        o[0] = x0; o[1] = y0; o[2] = x1; o[3] = y1;
    }
    if (lane == 0) p.counts[b] = base;
}

}  // namespace

extern "C" int rt_mask_postprocess(const rt_mask_post_desc* d, rt_stream_t stream) {
    if (!d || !d->pred || !d->sizes || !d->masks) return RT_ERR_BADARG;
    if (d->B <= 0 || d->Q <= 0 || d->h <= 0 || d->w <= 0 || d->max_h <= 0 || d->max_w <= 0) return RT_ERR_BADARG;
    if (d->masks_origin && (!d->orig || !d->origin_off || d->max_origin <= 0)) return RT_ERR_BADARG;
    if ((long long)d->max_h * d->max_w >= 0x7fffffffLL || (long long)d->max_origin >= 0x7fffffffLL) return RT_ERR_UNSUPPORTED;
    const long long n = d->masks_origin && d->max_origin > (long long)d->max_h * d->max_w ? d->max_origin : (long long)d->max_h * d->max_w;
    hipLaunchKernelGGL(mask_postprocess_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)(d->B * d->Q)), dim3(256), 0,
                       (hipStream_t)stream, *d);
    RT_CHECK_LAUNCH();
    return RT_OK;
}

extern "C" int rt_box_postprocess(const rt_box_post_desc* d, rt_stream_t stream) {
    if (!d || !d->boxes || !d->valid || !d->out || !d->counts) return RT_ERR_BADARG;
    if (d->B <= 0 || d->P <= 0 || d->K <= 0) return RT_ERR_BADARG;
    hipLaunchKernelGGL(box_postprocess_kernel, dim3((unsigned)d->B), dim3(64), 0, (hipStream_t)stream, *d);
    RT_CHECK_LAUNCH();
    return RT_OK;
}
