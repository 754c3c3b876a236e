// RES head (RefTRSeg, models/reftr_segmentation.py:151-280, 314-337) — the kernels around the implicit-GEMM convolutions:
// GroupNorm(8) + ReLU over NHWC rows with padded channel strides, nearest-upsample + FPN add, the joint-softmax
// attention map of MHAttentionMap, the concat that builds the mask head's input, and the bilinear + focal + dice loss.
// All HBM-bound streaming kernels: 16-B row accesses where the channel count allows, fp32 statistics.
#include "rt_common.h"

namespace {

// ---------------------------------------------------------------------------------------------------------------
// GroupNorm over x[b][p][c] (row stride ldx), C real channels in G groups; stats[b][g] = {sum, sumsq}.
// ---------------------------------------------------------------------------------------------------------------
// Deterministic: a thread adds into its own LDS slot per group, the block folds the 256 slots per group with a fixed tree and
// writes part[b][block][g]; gn_nhwc_finalize_kernel adds the blocks in a fixed order.  (G <= 16: 2 * G * 256 floats of LDS.)
__global__ __launch_bounds__(256) void gn_nhwc_stats_kernel(const float* __restrict__ x, float* __restrict__ part,
                                                            int HW, int C, int ldx, int G, int pix_per_block) {
    extern __shared__ float sm[];                  // [G][2][256]
    const int b = blockIdx.y, tid = threadIdx.x;
    const int p0 = blockIdx.x * pix_per_block, p1 = min(p0 + pix_per_block, HW);
    const int cpg = C / G;
    for (int i = tid; i < 2 * G * 256; i += 256) sm[i] = 0.f;      // only the thread's own slots: no barrier needed
    const int total = (p1 - p0) * C;
    // a thread walks elements e = t, t+256, ...: consecutive threads -> consecutive channels of a pixel (coalesced)
    float s = 0.f, ss = 0.f; int cur = -1;
    for (int e = tid; e < total; e += 256) {
        const int pix = e / C, c = e - pix * C;
        const int g = c / cpg;
        if (g != cur) {
            if (cur >= 0) { sm[(2 * cur) * 256 + tid] += s; sm[(2 * cur + 1) * 256 + tid] += ss; }
            cur = g; s = 0.f; ss = 0.f;
        }
        const float v = x[((size_t)b * HW + p0 + pix) * ldx + c];
        s += v; ss += v * v;
    }
    if (cur >= 0) { sm[(2 * cur) * 256 + tid] += s; sm[(2 * cur + 1) * 256 + tid] += ss; }
    __syncthreads();
    // wave w folds rows w, w+4, ... of the [2G][256] table: 4 slots per lane, then an xor butterfly
    const int lane = tid & 63, wave = tid >> 6;
    for (int r = wave; r < 2 * G; r += 4) {
        float v = (sm[r * 256 + lane] + sm[r * 256 + 64 + lane]) + (sm[r * 256 + 128 + lane] + sm[r * 256 + 192 + lane]);
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) part[((size_t)b * gridDim.x + blockIdx.x) * 2 * G + r] = v;
    }
}

// stats[b][g][k] = sum over blocks of part[b][block][g][k], one wave per (b, g, k) row, fixed order
__global__ __launch_bounds__(64) void gn_nhwc_finalize_kernel(const float* __restrict__ part, float* __restrict__ stats,
                                                              int nblk, int G2) {
    const int b = blockIdx.y, r = blockIdx.x, lane = threadIdx.x;
    float v = 0.f;
    for (int k = lane; k < nblk; k += 64) v += part[((size_t)b * nblk + k) * G2 + r];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if (lane == 0) stats[(size_t)b * G2 + r] = v;
}

__global__ __launch_bounds__(256) void gn_nhwc_apply_kernel(const rt_gn_nhwc_desc p) {
    const size_t rows = (size_t)p.B * p.HW;
    const int cpg = p.C / p.G;
    const float inv_n = 1.f / ((float)cpg * (float)p.HW);
    bf16_t* yb = (bf16_t*)p.y_bf16;
    const size_t total = rows * p.ldy;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t row = i / p.ldy; const int c = (int)(i - row * p.ldy);
        float y = 0.f;
        if (c < p.C) {
            const int b = (int)(row / p.HW), g = c / cpg;
            const float mean = p.stats[((size_t)b * p.G + g) * 2] * inv_n;
            const float var = fmaxf(p.stats[((size_t)b * p.G + g) * 2 + 1] * inv_n - mean * mean, 0.f);
            y = (p.x[row * p.ldx + c] - mean) * rsqrtf(var + p.eps) * p.gamma[c] + p.beta[c];
            if (p.act == RT_ACT_RELU) y = fmaxf(y, 0.f);
        }
        yb[i] = (bf16_t)y;          // channels C..ldy-1 are the zero padding the next convolution's K tiles expect
    }
}

// backward, pass 1: bstats[b][g] = {sum g, sum g*xhat} with g = dy*gamma (dy through the ReLU mask); dgamma / dbeta.
// Round 5: a thread OWNS its channel(s) -- lane = channel (coalesced rows), 256 / CP pixel rows per pass for narrow layers -- and keeps
// the four sums of a channel in registers over the block's pixels; LDS sees one atomic per thread and sum at the end (round 4 issued
// four LDS atomics per ELEMENT, two of them onto the group's pair of words: 93-103 us per layer whatever its size, 0.49 ms per step
// of configs[3]).  CP = channels per pass: C when C divides 256, else 256 (wide layers walk their channels in passes of 256).
__global__ __launch_bounds__(256) void gn_nhwc_bstats_kernel(const rt_gn_nhwc_bwd_desc p, int pix_per_block, int CP) {
    __shared__ float sm[2 * 64];
    extern __shared__ float dgb[];         // [2][C]
    const int b = blockIdx.y;
    const int p0 = blockIdx.x * pix_per_block, p1 = min(p0 + pix_per_block, p.HW);
    const int cpg = p.C / p.G;
    const float inv_n = 1.f / ((float)cpg * (float)p.HW);
    for (int i = threadIdx.x; i < 2 * p.G; i += 256) sm[i] = 0.f;
    for (int i = threadIdx.x; i < 2 * p.C; i += 256) dgb[i] = 0.f;
    __syncthreads();
    const int cl = threadIdx.x % CP, prow = threadIdx.x / CP, rows = 256 / CP;
    for (int c = cl; c < p.C; c += CP) {
        const int g = c / cpg;
        const float mean = p.stats[((size_t)b * p.G + g) * 2] * inv_n;
        const float var = fmaxf(p.stats[((size_t)b * p.G + g) * 2 + 1] * inv_n - mean * mean, 0.f);
        const float rstd = rsqrtf(var + p.eps), gam = p.gamma[c], bet = p.beta[c];
        float a_dg = 0.f, a_db = 0.f, a_g = 0.f, a_gx = 0.f;
        for (int pix = p0 + prow; pix < p1; pix += rows) {
            const size_t row = (size_t)b * p.HW + pix;
            const float xh = (p.x[row * p.ldx + c] - mean) * rstd;
            float d = p.dy[row * p.lddy + c];
            if (p.act == RT_ACT_RELU && xh * gam + bet <= 0.f) d = 0.f;
            a_dg += d * xh; a_db += d;
            const float gg = d * gam;
            a_g += gg; a_gx += gg * xh;
        }
        atomicAdd(&dgb[c], a_dg); atomicAdd(&dgb[p.C + c], a_db);
        atomicAdd(&sm[2 * g], a_g); atomicAdd(&sm[2 * g + 1], a_gx);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * p.G; i += 256) atomicAdd(p.bstats + (size_t)b * p.G * 2 + i, sm[i]);
    for (int c = threadIdx.x; c < p.C; c += 256) {
        if (p.dgamma) atomicAdd(p.dgamma + c, dgb[c]);
        if (p.dbeta) atomicAdd(p.dbeta + c, dgb[p.C + c]);
    }
}

__global__ __launch_bounds__(256) void gn_nhwc_bwd_apply_kernel(const rt_gn_nhwc_bwd_desc p) {
    const size_t rows = (size_t)p.B * p.HW;
    const int cpg = p.C / p.G;
    const float inv_n = 1.f / ((float)cpg * (float)p.HW);
    bf16_t* dxb = (bf16_t*)p.dx_bf16;
    const size_t total = rows * p.lddx;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t row = i / p.lddx; const int c = (int)(i - row * p.lddx);
        float dx = 0.f;
        if (c < p.C) {
            const int b = (int)(row / p.HW), g = c / cpg;
            const float mean = p.stats[((size_t)b * p.G + g) * 2] * inv_n;
            const float var = fmaxf(p.stats[((size_t)b * p.G + g) * 2 + 1] * inv_n - mean * mean, 0.f);
            const float rstd = rsqrtf(var + p.eps);
            const float xh = (p.x[row * p.ldx + c] - mean) * rstd;
            float d = p.dy[row * p.lddy + c];
            if (p.act == RT_ACT_RELU && xh * p.gamma[c] + p.beta[c] <= 0.f) d = 0.f;
            const float m1 = p.bstats[((size_t)b * p.G + g) * 2] * inv_n, m2 = p.bstats[((size_t)b * p.G + g) * 2 + 1] * inv_n;
            dx = rstd * (d * p.gamma[c] - m1 - xh * m2);
        }
        dxb[i] = (bf16_t)dx;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// out[b][y][x][c] = fpn[b][y][x][c] + a[b][y*h/H][x*w/W][c]   (F.interpolate nearest: src = floor(dst * in / out))
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void upsample_add_kernel(const rt_upsample_add_desc p) {
    bf16_t* out = (bf16_t*)p.out_bf16; const bf16_t* a = (const bf16_t*)p.a_bf16;
    const size_t total = (size_t)p.B * p.H * p.W * p.ldo;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t row = i / p.ldo; const int c = (int)(i - row * p.ldo);
        float v = 0.f;
        if (c < p.C) {
            const int x = (int)(row % p.W); const size_t t = row / p.W; const int y = (int)(t % p.H); const int b = (int)(t / p.H);
            const int ys = min((int)((long long)y * p.h / p.H), p.h - 1), xs = min((int)((long long)x * p.w / p.W), p.w - 1);
            v = p.fpn[row * p.ldf + c] + (float)a[(((size_t)b * p.h + ys) * p.w + xs) * p.lda + c];
        }
        out[i] = (bf16_t)v;
    }
}

// backward: dyb = bf16(dy) (the FPN adapter's output gradient), da[b][ys][xs][c] = sum of dy over the pixels mapped there
__global__ __launch_bounds__(256) void upsample_add_bwd_kernel(const rt_upsample_add_bwd_desc p) {
    bf16_t* dyb = (bf16_t*)p.dy_bf16;
    const size_t nsrc = (size_t)p.B * p.h * p.w * p.C, ndst = (size_t)p.B * p.H * p.W * p.lddyb;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < nsrc + ndst; i += (size_t)gridDim.x * 256) {
        if (i < nsrc) {
            const int c = (int)(i % p.C); size_t t = i / p.C;
            const int xs = (int)(t % p.w); t /= p.w; const int ys = (int)(t % p.h); const int b = (int)(t / p.h);
            const int y0 = (int)(((long long)ys * p.H + p.h - 1) / p.h), y1 = (int)(((long long)(ys + 1) * p.H + p.h - 1) / p.h);
            const int x0 = (int)(((long long)xs * p.W + p.w - 1) / p.w), x1 = (int)(((long long)(xs + 1) * p.W + p.w - 1) / p.w);
            float s = 0.f;
            for (int y = y0; y < y1 && y < p.H; ++y)
                for (int x = x0; x < x1 && x < p.W; ++x) s += p.dy[(((size_t)b * p.H + y) * p.W + x) * p.lddy + c];
            p.da[(((size_t)b * p.h + ys) * p.w + xs) * p.ldda + c] = s;
        } else if (dyb) {
            const size_t j = i - nsrc;
            const size_t row = j / p.lddyb; const int c = (int)(j - row * p.lddyb);
            dyb[j] = (bf16_t)(c < p.C ? p.dy[row * p.lddy + c] : 0.f);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// MHAttentionMap (reftr_segmentation.py:196-208): P[b][n][p] = softmax over (n, p) JOINTLY of
// norm * <q[b][n*dh..], k[b][p][n*dh..]>, padded pixels -> -inf.  One workgroup per image.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_map_fwd_kernel(const rt_attn_map_desc p) {
    extern __shared__ float lg[];            // [nh * HW] logits, then probabilities
    __shared__ float red[8];
    const int b = blockIdx.x, t = threadIdx.x;
    const int dh = p.E / p.nh, n_ent = p.nh * p.HW;
    const float* q = p.q + (size_t)b * p.E;
    float mx = -INFINITY;
    for (int e = t; e < n_ent; e += 256) {
        const int n = e / p.HW, pix = e - n * p.HW;
        float v = -INFINITY;
        if (!p.mask[(size_t)b * p.HW + pix]) {
            const float* kr = p.k + ((size_t)b * p.k_rows_per_img + p.k_row_off + pix) * p.ldk + n * dh;
            float s = 0.f;
            for (int c = 0; c < dh; ++c) s += q[n * dh + c] * kr[c];
            v = s * p.norm;
        }
        lg[e] = v; mx = fmaxf(mx, v);
    }
    mx = rt_wave_max(mx);
    if ((t & 63) == 0) red[t >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    for (int e = t; e < n_ent; e += 256) { const float v = __expf(lg[e] - mx); lg[e] = v; sum += v; }
    sum = rt_wave_sum(sum);
    __syncthreads();
    if ((t & 63) == 0) red[4 + (t >> 6)] = sum;
    __syncthreads();
    const float inv = 1.f / (red[4] + red[5] + red[6] + red[7]);
    bf16_t* xc = (bf16_t*)p.concat_bf16;
    for (int e = t; e < n_ent; e += 256) {
        const int n = e / p.HW, pix = e - n * p.HW;
        const float pr = lg[e] * inv;
        p.P[(size_t)b * n_ent + e] = pr;
        if (xc) xc[((size_t)b * p.HW + pix) * p.ld_concat + p.concat_col + n] = (bf16_t)pr;
    }
}

// dlogit = norm * P * (dP - sum(P * dP));  dq[n*dh+c] = sum_p dlogit[n][p] k[p][n*dh+c];  dk[p][n*dh+c] = dlogit[n][p] q[n*dh+c]
__global__ __launch_bounds__(256) void attn_map_bwd_kernel(const rt_attn_map_bwd_desc p) {
    extern __shared__ float dl[];            // [nh * HW]
    __shared__ float red[4];
    const int b = blockIdx.x, t = threadIdx.x;
    const int dh = p.E / p.nh, n_ent = p.nh * p.HW;
    float dot = 0.f;
    for (int e = t; e < n_ent; e += 256) {
        const int n = e / p.HW, pix = e - n * p.HW;
        const float pr = p.P[(size_t)b * n_ent + e];
        const float dp = p.dconcat[((size_t)b * p.HW + pix) * p.ld_dconcat + p.concat_col + n];
        dl[e] = dp; dot += pr * dp;
    }
    dot = rt_wave_sum(dot);
    if ((t & 63) == 0) red[t >> 6] = dot;
    __syncthreads();
    dot = red[0] + red[1] + red[2] + red[3];
    for (int e = t; e < n_ent; e += 256) dl[e] = p.norm * p.P[(size_t)b * n_ent + e] * (dl[e] - dot);
    __syncthreads();
    // thread t <-> feature f = n*dh + c (E <= 256)
    if (t < p.E) {
        const int n = t / dh;
        const float qv = p.q[(size_t)b * p.E + t];
        float acc = 0.f;
        for (int pix = 0; pix < p.HW; ++pix) {
            const float d = dl[n * p.HW + pix];
            const size_t kr = ((size_t)b * p.k_rows_per_img + p.k_row_off + pix) * p.ldk + t;
            acc += d * p.k[kr];
            p.dk[kr] = d * qv;
        }
        p.dq[(size_t)b * p.E + t] = acc;
    }
}

// X0[b*HW+p] = [ bf16(src[b][p][0:E]) | bf16(mem[b*S + off + p][0:E]) | (attention map, written by attn_map_fwd) | 0 ]
__global__ __launch_bounds__(256) void seg_concat_kernel(const rt_seg_concat_desc p) {
    bf16_t* out = (bf16_t*)p.out_bf16;
    const size_t total = (size_t)p.B * p.HW * p.ldo;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t row = i / p.ldo; const int c = (int)(i - row * p.ldo);
        const int b = (int)(row / p.HW), pix = (int)(row - (size_t)b * p.HW);
        if (c < p.E) out[i] = (bf16_t)p.src[((size_t)b * p.src_rows_per_img + p.src_row_off + pix) * p.E + c];
        else if (c < 2 * p.E) out[i] = (bf16_t)p.mem[((size_t)b * p.mem_rows_per_img + p.mem_row_off + pix) * p.E + (c - p.E)];
        else if (c >= 2 * p.E + p.nh) out[i] = (bf16_t)0.f;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// loss_masks (reftr_segmentation.py:314-337): bilinear upsample (align_corners=False) of the mask logits to the
// padded target size, sigmoid focal loss (alpha 0.25, gamma 2) and dice loss.
// sums[b] = {focal sum, sum p*t, sum p, sum t}
// ---------------------------------------------------------------------------------------------------------------
struct Bilin { int y0, y1, x0, x1; float wy, wx; };
__device__ __forceinline__ Bilin bilin(int y, int x, int h, int w, float sy, float sx) {
    Bilin r;
    float fy = fmaxf(((float)y + 0.5f) * sy - 0.5f, 0.f), fx = fmaxf(((float)x + 0.5f) * sx - 0.5f, 0.f);
    r.y0 = min((int)fy, h - 1); r.x0 = min((int)fx, w - 1);
    r.y1 = min(r.y0 + 1, h - 1); r.x1 = min(r.x0 + 1, w - 1);
    r.wy = fy - (float)r.y0; r.wx = fx - (float)r.x0;
    return r;
}

__global__ __launch_bounds__(256) void mask_loss_fwd_kernel(const rt_mask_loss_desc p) {
    __shared__ float sm[4][4];
    const int b = blockIdx.y;
    const float sy = (float)p.h / (float)p.Ht, sx = (float)p.w / (float)p.Wt;
    const float* z = p.pred + (size_t)b * p.h * p.w * p.ldp;
    float f = 0.f, it = 0.f, ps = 0.f, ts = 0.f;
    const int total = p.Ht * p.Wt;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int y = i / p.Wt, x = i - y * p.Wt;
        const Bilin bl = bilin(y, x, p.h, p.w, sy, sx);
        const float v = (1.f - bl.wy) * ((1.f - bl.wx) * z[((size_t)bl.y0 * p.w + bl.x0) * p.ldp] + bl.wx * z[((size_t)bl.y0 * p.w + bl.x1) * p.ldp])
                      + bl.wy * ((1.f - bl.wx) * z[((size_t)bl.y1 * p.w + bl.x0) * p.ldp] + bl.wx * z[((size_t)bl.y1 * p.w + bl.x1) * p.ldp]);
        const float t = p.target[(size_t)b * total + i] ? 1.f : 0.f;
        const float pr = 1.f / (1.f + __expf(-v));
        const float ce = fmaxf(v, 0.f) - v * t + log1pf(__expf(-fabsf(v)));
        const float pt = pr * t + (1.f - pr) * (1.f - t);
        const float at = 0.25f * t + 0.75f * (1.f - t);
        f += at * ce * (1.f - pt) * (1.f - pt);
        it += pr * t; ps += pr; ts += t;
    }
    f = rt_wave_sum(f); it = rt_wave_sum(it); ps = rt_wave_sum(ps); ts = rt_wave_sum(ts);
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sm[wv][0] = f; sm[wv][1] = it; sm[wv][2] = ps; sm[wv][3] = ts; }
    __syncthreads();
    if (threadIdx.x < 4) atomicAdd(p.sums + b * 4 + threadIdx.x, sm[0][threadIdx.x] + sm[1][threadIdx.x] + sm[2][threadIdx.x] + sm[3][threadIdx.x]);
}

// losses[0] = focal, losses[1] = dice (both already divided by the normaliser B*Q)
__global__ void mask_loss_final_kernel(const float* __restrict__ sums, float* __restrict__ losses, int B, int npix, float inv_norm) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float f = 0.f, d = 0.f;
    for (int b = 0; b < B; ++b) {
        f += sums[b * 4] / (float)npix;
        d += 1.f - (2.f * sums[b * 4 + 1] + 1.f) / (sums[b * 4 + 2] + sums[b * 4 + 3] + 1.f);
    }
    losses[0] = f * inv_norm; losses[1] = d * inv_norm;
}

// dpred[b][ys][xs] += sum over target pixels of (w_focal * dfocal/dz + w_dice * ddice/dz) * bilinear weight of (ys, xs) in that pixel.
// Round 5: two passes without atomics.  Pass 1 (one thread per TARGET pixel, coalesced): g = w_focal * dfocal/dz + w_dice * ddice/dz
// -> gbuf.  Pass 2 (one thread per SOURCE pixel): a gather over the target pixels whose bilinear footprint contains it -- a
// (2 / scale + 2)^2 window, corners and weights recomputed with the forward's `bilin`, so exactly the scatter's terms -- and ONE add
// to dpred.  Round 4 scattered four global atomics per target pixel (13 M atomics, sixteen-fold contended at the 4x upsample: 275 us
// per step of configs[3]); a single-pass gather that re-evaluated the pixel function per corner took 143 us.
__global__ __launch_bounds__(256) void mask_loss_grad_kernel(const rt_mask_loss_desc p) {
    const int b = blockIdx.y;
    const float sy = (float)p.h / (float)p.Ht, sx = (float)p.w / (float)p.Wt;
    const float* z = p.pred + (size_t)b * p.h * p.w * p.ldp;
    const int total = p.Ht * p.Wt;
    const float num = 2.f * p.sums[b * 4 + 1] + 1.f, den = p.sums[b * 4 + 2] + p.sums[b * 4 + 3] + 1.f;
    const float gf = p.g_focal[0] * p.inv_norm / (float)total, gd = p.g_dice[0] * p.inv_norm;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int y = i / p.Wt, x = i - y * p.Wt;
        const Bilin bl = bilin(y, x, p.h, p.w, sy, sx);
        const float v = (1.f - bl.wy) * ((1.f - bl.wx) * z[((size_t)bl.y0 * p.w + bl.x0) * p.ldp] + bl.wx * z[((size_t)bl.y0 * p.w + bl.x1) * p.ldp])
                      + bl.wy * ((1.f - bl.wx) * z[((size_t)bl.y1 * p.w + bl.x0) * p.ldp] + bl.wx * z[((size_t)bl.y1 * p.w + bl.x1) * p.ldp]);
        const float t = p.target[(size_t)b * total + i] ? 1.f : 0.f;
        const float pr = 1.f / (1.f + __expf(-v));
        const float ce = fmaxf(v, 0.f) - v * t + log1pf(__expf(-fabsf(v)));
        const float pt = pr * t + (1.f - pr) * (1.f - t);
        const float at = 0.25f * t + 0.75f * (1.f - t);
        const float dpr = pr * (1.f - pr);
        // d/dz [ce * (1-pt)^2] = (pr - t) * (1-pt)^2 - 2 ce (1-pt) * dpt/dz,  dpt/dz = (2t-1) * pr(1-pr)
        const float dfocal = at * ((pr - t) * (1.f - pt) * (1.f - pt) - 2.f * ce * (1.f - pt) * (2.f * t - 1.f) * dpr);
        // dice = 1 - num/den: d/dz = -(2 t den - num) / den^2 * pr(1-pr)
        const float ddice = -(2.f * t * den - num) / (den * den) * dpr;
        p.gbuf[(size_t)b * total + i] = gf * dfocal + gd * ddice;
    }
}
__global__ __launch_bounds__(256) void mask_loss_gather_kernel(const rt_mask_loss_desc p) {
    const int b = blockIdx.y;
    const int src = blockIdx.x * 256 + threadIdx.x;
    if (src >= p.h * p.w) return;
    const int ys = src / p.w, xs = src - ys * p.w;
    const float sy = (float)p.h / (float)p.Ht, sx = (float)p.w / (float)p.Wt;
    const float* g = p.gbuf + (size_t)b * p.Ht * p.Wt;
    // target rows / columns whose y0 or y1 (x0 or x1) can be ys (xs): floor(f) in {s - 1, s}, one pixel of slack on both sides;
    // the last source row / column also collects every target pixel clamped onto it
    int ylo = (int)floorf(((float)ys - 0.5f) / sy - 0.5f) - 1, yhi = (int)ceilf(((float)ys + 1.5f) / sy - 0.5f) + 1;
    int xlo = (int)floorf(((float)xs - 0.5f) / sx - 0.5f) - 1, xhi = (int)ceilf(((float)xs + 1.5f) / sx - 0.5f) + 1;
    ylo = max(ylo, 0); xlo = max(xlo, 0);
    yhi = ys == p.h - 1 ? p.Ht - 1 : min(yhi, p.Ht - 1); xhi = xs == p.w - 1 ? p.Wt - 1 : min(xhi, p.Wt - 1);
    float acc = 0.f;
    for (int y = ylo; y <= yhi; ++y) {
        const Bilin by = bilin(y, 0, p.h, p.w, sy, sx);
        const float wyv = (by.y0 == ys ? 1.f - by.wy : 0.f) + (by.y1 == ys ? by.wy : 0.f);
        if (wyv == 0.f) continue;
        float row = 0.f;
        for (int x = xlo; x <= xhi; ++x) {
            const Bilin bx = bilin(0, x, p.h, p.w, sy, sx);
            const float wxv = (bx.x0 == xs ? 1.f - bx.wx : 0.f) + (bx.x1 == xs ? bx.wx : 0.f);
            if (wxv != 0.f) row += g[(size_t)y * p.Wt + x] * wxv;
        }
        acc += row * wyv;
    }
    p.dpred[((size_t)b * p.h * p.w + src) * p.lddp] += acc;          // one owner per element
}

static inline int grid_for(size_t total, int cap = 4096) {
    size_t b = (total + 255) / 256;
    return (int)(b > (size_t)cap ? cap : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" int rt_gn_nhwc_fwd(const rt_gn_nhwc_desc* d, rt_stream_t stream) {
    if (!d || !d->x || !d->gamma || !d->beta || !d->stats || !d->y_bf16) return RT_ERR_BADARG;
    if (d->C <= 0 || d->G <= 0 || d->G > 64 || (d->C % d->G) || d->ldx < d->C || d->ldy < d->C || d->B <= 0 || d->HW <= 0) return RT_ERR_UNSUPPORTED;
    if (d->G > 16) return RT_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    int ppb = (int)((4096 + d->C - 1) / d->C); if (ppb < 1) ppb = 1;
    const int nblk = (d->HW + ppb - 1) / ppb;
    if (!d->partials || d->partial_blocks < nblk) return RT_ERR_BADARG;
    hipLaunchKernelGGL(gn_nhwc_stats_kernel, dim3(nblk, d->B), dim3(256), 2 * d->G * 256 * sizeof(float), s,
                       d->x, d->partials, d->HW, d->C, d->ldx, d->G, ppb);
    RT_CHECK_LAUNCH();
    hipLaunchKernelGGL(gn_nhwc_finalize_kernel, dim3(2 * d->G, d->B), dim3(64), 0, s, d->partials, d->stats, nblk, 2 * d->G);
    RT_CHECK_LAUNCH();
    hipLaunchKernelGGL(gn_nhwc_apply_kernel, dim3(grid_for((size_t)d->B * d->HW * d->ldy)), dim3(256), 0, s, *d);
    RT_CHECK_LAUNCH();
    return RT_OK;
}

extern "C" int rt_gn_nhwc_bwd(const rt_gn_nhwc_bwd_desc* d, rt_stream_t stream) {
    if (!d || !d->x || !d->dy || !d->gamma || !d->beta || !d->stats || !d->bstats || !d->dx_bf16) return RT_ERR_BADARG;
    if (d->C <= 0 || d->G <= 0 || d->G > 64 || (d->C % d->G) || d->ldx < d->C || d->lddy < d->C || d->lddx < d->C || d->C > 4096) return RT_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = rt_zero_f32(d->bstats, 2 * (size_t)d->B * d->G, s);
    if (e != hipSuccess) return (int)e;
    int ppb = (int)((4096 + d->C - 1) / d->C); if (ppb < 1) ppb = 1;
    const int CP = (d->C <= 256 && 256 % d->C == 0) ? d->C : 256;
    hipLaunchKernelGGL(gn_nhwc_bstats_kernel, dim3((d->HW + ppb - 1) / ppb, d->B), dim3(256), 2 * d->C * sizeof(float), s, *d, ppb, CP);
    RT_CHECK_LAUNCH();
    hipLaunchKernelGGL(gn_nhwc_bwd_apply_kernel, dim3(grid_for((size_t)d->B * d->HW * d->lddx)), dim3(256), 0, s, *d);
    RT_CHECK_LAUNCH();
    return RT_OK;
}

extern "C" int rt_upsample_add(const rt_upsample_add_desc* d, rt_stream_t stream) {
    if (!d || !d->fpn || !d->a_bf16 || !d->out_bf16) return RT_ERR_BADARG;
    if (d->C <= 0 || d->ldf < d->C || d->lda < d->C || d->ldo < d->C || d->h <= 0 || d->w <= 0 || d->H < d->h || d->W < d->w) return RT_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(upsample_add_kernel, dim3(grid_for((size_t)d->B * d->H * d->W * d->ldo)), dim3(256), 0, (hipStream_t)stream, *d);
    RT_CHECK_LAUNCH();
    return RT_OK;
}

extern "C" int rt_upsample_add_bwd(const rt_upsample_add_bwd_desc* d, rt_stream_t stream) {
    if (!d || !d->dy || !d->da) return RT_ERR_BADARG;
    if (d->C <= 0 || d->lddy < d->C || d->ldda < d->C || (d->dy_bf16 && d->lddyb < d->C)) return RT_ERR_UNSUPPORTED;
    const size_t total = (size_t)d->B * d->h * d->w * d->C + (d->dy_bf16 ? (size_t)d->B * d->H * d->W * d->lddyb : 0);
    hipLaunchKernelGGL(upsample_add_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, *d);
    RT_CHECK_LAUNCH();
    return RT_OK;
}

extern "C" int rt_attn_map_fwd(const rt_attn_map_desc* d, rt_stream_t stream) {
    if (!d || !d->q || !d->k || !d->mask || !d->P) return RT_ERR_BADARG;
    if (d->E <= 0 || d->nh <= 0 || (d->E % d->nh) || d->HW <= 0 || (size_t)d->nh * d->HW * 4 > 150000) return RT_ERR_UNSUPPORTED;
    const size_t smem = (size_t)d->nh * d->HW * sizeof(float);
    if (smem > 65536) (void)hipFuncSetAttribute((const void*)attn_map_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL(attn_map_fwd_kernel, dim3(d->B), dim3(256), smem, (hipStream_t)stream, *d);
    RT_CHECK_LAUNCH();
    return RT_OK;
}

extern "C" int rt_attn_map_bwd(const rt_attn_map_bwd_desc* d, rt_stream_t stream) {
    if (!d || !d->q || !d->k || !d->P || !d->dconcat || !d->dq || !d->dk) return RT_ERR_BADARG;
    if (d->E <= 0 || d->E > 256 || d->nh <= 0 || (d->E % d->nh) || (size_t)d->nh * d->HW * 4 > 150000) return RT_ERR_UNSUPPORTED;
    const size_t smem = (size_t)d->nh * d->HW * sizeof(float);
    if (smem > 65536) (void)hipFuncSetAttribute((const void*)attn_map_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL(attn_map_bwd_kernel, dim3(d->B), dim3(256), smem, (hipStream_t)stream, *d);
    RT_CHECK_LAUNCH();
    return RT_OK;
}

extern "C" int rt_seg_concat(const rt_seg_concat_desc* d, rt_stream_t stream) {
    if (!d || !d->src || !d->mem || !d->out_bf16) return RT_ERR_BADARG;
    if (d->ldo < 2 * d->E + d->nh || d->E <= 0) return RT_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(seg_concat_kernel, dim3(grid_for((size_t)d->B * d->HW * d->ldo)), dim3(256), 0, (hipStream_t)stream, *d);
    RT_CHECK_LAUNCH();
    return RT_OK;
}

extern "C" int rt_mask_loss(const rt_mask_loss_desc* d, rt_stream_t stream) {
    if (!d || !d->pred || !d->target || !d->sums) return RT_ERR_BADARG;
    if (d->B <= 0 || d->h <= 0 || d->w <= 0 || d->Ht <= 0 || d->Wt <= 0 || d->ldp <= 0) return RT_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const int gx = grid_for((size_t)d->Ht * d->Wt, 256);
    if (!d->dpred) {
        if (!d->losses) return RT_ERR_BADARG;
        hipError_t e = rt_zero_f32(d->sums, 4 * (size_t)d->B, s);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(mask_loss_fwd_kernel, dim3(gx, d->B), dim3(256), 0, s, *d);
        RT_CHECK_LAUNCH();
        hipLaunchKernelGGL(mask_loss_final_kernel, dim3(1), dim3(64), 0, s, d->sums, d->losses, d->B, d->Ht * d->Wt, d->inv_norm);
        RT_CHECK_LAUNCH();
    } else {
        if (!d->g_focal || !d->g_dice || d->lddp <= 0 || !d->gbuf) return RT_ERR_BADARG;
        hipLaunchKernelGGL(mask_loss_grad_kernel, dim3(grid_for((size_t)d->Ht * d->Wt, 1024), d->B), dim3(256), 0, s, *d);
        RT_CHECK_LAUNCH();
        hipLaunchKernelGGL(mask_loss_gather_kernel, dim3((d->h * d->w + 255) / 256, d->B), dim3(256), 0, s, *d);
        RT_CHECK_LAUNCH();
    }
    return RT_OK;
}


// ------------------------------------------------------------------------------------------------ CEM block
namespace {

struct CemPix { float a, t, cosv, nrm; };
__device__ __forceinline__ CemPix cem_pixel(const float (&x)[16], const float (&w2)[16], float b2, const float (&r)[16]) {
    float a = b2, dot = 0.f, n2 = 0.f;
#pragma unroll
    for (int c = 0; c < 16; ++c) { a += x[c] * w2[c]; dot += r[c] * x[c]; n2 += x[c] * x[c]; }
    CemPix o;
    o.a = a; o.nrm = fmaxf(sqrtf(n2), 1e-12f); o.cosv = dot / o.nrm;
    o.t = fminf(fmaxf((o.cosv + 1.f) * 0.5f, 1e-6f), 1.f - 1e-6f);
    return o;
}
__device__ __forceinline__ void cem_load(const bf16_t* row, float (&x)[16]) {
    const bf16x8 lo = *reinterpret_cast<const bf16x8*>(row), hi = *reinterpret_cast<const bf16x8*>(row + 8);
#pragma unroll
    for (int c = 0; c < 8; ++c) { x[c] = (float)lo[c]; x[8 + c] = (float)hi[c]; }
}
__device__ __forceinline__ void cem_unit(const float* u, float (&r)[16], float& un) {
    float n2 = 0.f;
#pragma unroll
    for (int c = 0; c < 16; ++c) n2 += u[c] * u[c];
    un = fmaxf(sqrtf(n2), 1e-12f);
#pragma unroll
    for (int c = 0; c < 16; ++c) r[c] = u[c] / un;
}

__global__ __launch_bounds__(1024) void cem_fwd_kernel(const rt_cem_desc p) {
    __shared__ float sm[3][16];
    __shared__ float su[16];
    const int b = blockIdx.x;
    {   // u = c3(hs_b): wave w computes output w
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const bf16_t* h = (const bf16_t*)p.hs + (size_t)b * p.E;
        float a = 0.f;
        for (int k = lane; k < p.E; k += 64) a += (float)h[k] * p.w3[(size_t)wave * p.E + k];
        a = rt_wave_sum(a);
        if (lane == 0) { su[wave] = a + p.b3[wave]; p.u[b * 16 + wave] = su[wave]; }
    }
    __syncthreads();
    float w2[16], r[16], un;
#pragma unroll
    for (int c = 0; c < 16; ++c) w2[c] = p.w2[c];
    cem_unit(su, r, un);
    const float b2 = p.b2[0];
    const bf16_t* res = (const bf16_t*)p.res + (size_t)b * p.HW * p.ld;
    float m = -INFINITY, s = 0.f, e = 0.f;                       // online softmax: s = sum exp(a - m), e = sum t exp(a - m)
    for (int i = threadIdx.x; i < p.HW; i += 1024) {
        float x[16];
        cem_load(res + (size_t)i * p.ld, x);
        const CemPix q = cem_pixel(x, w2, b2, r);
        const float mn = fmaxf(m, q.a), sc = __expf(m - mn), w = __expf(q.a - mn);
        s = s * sc + w; e = e * sc + q.t * w; m = mn;
    }
    // combine (m, s, e) over the block: wave shuffle, then the 16 waves through LDS
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(s, o, 64), e2 = __shfl_xor(e, o, 64);
        const float mn = fmaxf(m, m2), c1 = (m == -INFINITY) ? 0.f : __expf(m - mn), c2 = (m2 == -INFINITY) ? 0.f : __expf(m2 - mn);
        s = s * c1 + s2 * c2; e = e * c1 + e2 * c2; m = mn;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { sm[0][wave] = m; sm[1][wave] = s; sm[2][wave] = e; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float M = -INFINITY;
        for (int w = 0; w < 16; ++w) M = fmaxf(M, sm[0][w]);
        float S = 0.f, E = 0.f;
        for (int w = 0; w < 16; ++w) { const float c = (sm[0][w] == -INFINITY) ? 0.f : __expf(sm[0][w] - M); S += sm[1][w] * c; E += sm[2][w] * c; }
        const float energy = E / S;
        p.energy[b] = energy; p.stats[2 * b] = M; p.stats[2 * b + 1] = S;
        atomicAdd(p.loss, -__logf(energy + 1e-6f) / (float)p.B);
    }
}

__global__ __launch_bounds__(1024) void cem_bwd_kernel(const rt_cem_desc p) {
    __shared__ float sm[16][32];
    const int b = blockIdx.x;
    float w2[16], r[16], un;
#pragma unroll
    for (int c = 0; c < 16; ++c) w2[c] = p.w2[c];
    cem_unit(p.u + b * 16, r, un);
    const float b2 = p.b2[0];
    const float E = p.energy[b], M = p.stats[2 * b], S = p.stats[2 * b + 1];
    const float dE = -p.g[0] / ((float)p.B * (E + 1e-6f));
    const bf16_t* res = (const bf16_t*)p.res + (size_t)b * p.HW * p.ld;
    float* dres = p.dres + (size_t)b * p.HW * p.lddr;
    float dr[16], dw[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) { dr[c] = 0.f; dw[c] = 0.f; }
    for (int i = threadIdx.x; i < p.HW; i += 1024) {
        float x[16];
        cem_load(res + (size_t)i * p.ld, x);
        const CemPix q = cem_pixel(x, w2, b2, r);
        const float ec = __expf(q.a - M) / S;
        const float da = ec * dE * (q.t - E);                    // through the softmax over the pixels
        const float tt = (q.cosv + 1.f) * 0.5f;
        const float dcos = (tt > 1e-6f && tt < 1.f - 1e-6f) ? 0.5f * dE * ec : 0.f;
        const float k1 = dcos / q.nrm, k2 = dcos * q.cosv / (q.nrm * q.nrm);
        float* o = dres + (size_t)i * p.lddr;
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
            f32x4 v = *reinterpret_cast<f32x4*>(o + c4 * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = c4 * 4 + e;
                v[e] += w2[c] * da + k1 * r[c] - k2 * x[c];
                dr[c] += k1 * x[c]; dw[c] += da * x[c];
            }
            *reinterpret_cast<f32x4*>(o + c4 * 4) = v;
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const float a = rt_wave_sum(dr[c]), d2 = rt_wave_sum(dw[c]);
        if (lane == 0) { sm[wave][c] = a; sm[wave][16 + c] = d2; }
    }
    __syncthreads();
    if (threadIdx.x < 32) {
        float t = 0.f;
        for (int w = 0; w < 16; ++w) t += sm[w][threadIdx.x];
        sm[0][threadIdx.x] = t;
    }
    __syncthreads();
    if (threadIdx.x < 16) {
        // r = u / |u|: du = (dr - (dr . r) r) / |u|
        float dot = 0.f;
        for (int c = 0; c < 16; ++c) dot += sm[0][c] * r[c];
        const float du = (sm[0][threadIdx.x] - dot * r[threadIdx.x]) / un;
        sm[1][threadIdx.x] = du;
        atomicAdd(p.db3 + threadIdx.x, du);
        atomicAdd(p.dw2 + threadIdx.x, sm[0][16 + threadIdx.x]);
    }
    __syncthreads();
    // c3 backward: dhs_b = du @ w3, dw3 += du (x) hs_b
    const bf16_t* h = (const bf16_t*)p.hs + (size_t)b * p.E;
    for (int k = threadIdx.x; k < p.E; k += 1024) {
        const float hk = (float)h[k];
        float a = 0.f;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const float du = sm[1][c];
            a += du * p.w3[(size_t)c * p.E + k];
            atomicAdd(p.dw3 + (size_t)c * p.E + k, du * hk);
        }
        p.dhs[(size_t)b * p.E + k] = a;
    }
}

}  // namespace

extern "C" int rt_cem_fwd(const rt_cem_desc* d, rt_stream_t stream) {
    if (!d || !d->hs || !d->w3 || !d->b3 || !d->u || !d->res || !d->w2 || !d->b2 || !d->energy || !d->stats || !d->loss) return RT_ERR_BADARG;
    if (d->B <= 0 || d->HW <= 0 || d->E <= 0 || d->ld < 16 || (d->ld & 7)) return RT_ERR_BADARG;
    if (d->E != 256) return RT_ERR_UNSUPPORTED;      // the kernels hold hidden/16 == 16 channels per image in 16 waves / fixed arrays
    hipStream_t s = (hipStream_t)stream;
    const hipError_t e = rt_zero_f32(d->loss, 1, s);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(cem_fwd_kernel, dim3((unsigned)d->B), dim3(1024), 0, s, *d);
    RT_CHECK_LAUNCH();
    return RT_OK;
}

extern "C" int rt_cem_bwd(const rt_cem_desc* d, rt_stream_t stream) {
    if (!d || !d->hs || !d->w3 || !d->u || !d->res || !d->w2 || !d->b2 || !d->energy || !d->stats || !d->g || !d->dres || !d->dhs ||
        !d->dw3 || !d->db3 || !d->dw2) return RT_ERR_BADARG;
    if (d->B <= 0 || d->HW <= 0 || d->E <= 0 || d->ld < 16 || (d->ld & 7) || d->lddr < 16 || (d->lddr & 3)) return RT_ERR_BADARG;
    if (d->E != 256) return RT_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(cem_bwd_kernel, dim3((unsigned)d->B), dim3(1024), 0, (hipStream_t)stream, *d);
    RT_CHECK_LAUNCH();
    return RT_OK;
}
