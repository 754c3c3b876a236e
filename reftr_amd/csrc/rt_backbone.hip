// Backbone-side kernels that are not the generic implicit GEMM:
//   rt_img_pack      NCHW fp32 image -> zero-haloed NHWC4 bf16 (the stem's input layout)
//   rt_stem_conv     7x7/2 conv + FrozenBN + ReLU on MFMA (K = 7 rows x (8 taps x 4 ch) = 7 x 32)
//   rt_maxpool3x3s2  NHWC bf16 max-pool 3x3 / stride 2 / pad 1
//   rt_stem_pool     the two above in one launch (the stem output never reaches HBM)
//   rt_weight_prep   fp32 master weight -> bf16 GEMM operand(s): [N][T][C] (x FrozenBN scale) and [C][T][N]
//   rt_mask_posenc   pad-mask nearest downsample + DETR sine position encoding (+ level / token-type embeds)
#include "rt_common.h"

namespace {

// ---------------------------------------------------------------- image pack
__global__ __launch_bounds__(256) void img_pack_kernel(const float* __restrict__ img, bf16_t* __restrict__ out,
                                                       int B, int H, int W, int Hp, int Wp) {
    const size_t total = (size_t)B * Hp * Wp;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int xp = (int)(i % Wp);
        const size_t t = i / Wp;
        const int yp = (int)(t % Hp);
        const int b = (int)(t / Hp);
        const int y = yp - 3, x = xp - 3;
        bf16x4 v = {(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
        if (y >= 0 && y < H && x >= 0 && x < W) {
            const size_t base = ((size_t)b * 3 * H + y) * W + x;
            v[0] = (bf16_t)img[base];
            v[1] = (bf16_t)img[base + (size_t)H * W];
            v[2] = (bf16_t)img[base + 2 * (size_t)H * W];
        }
        *reinterpret_cast<bf16x4*>(out + i * 4) = v;
    }
}

// ---------------------------------------------------------------- stem conv
// One wave per output row (b, oh): the 64x224 folded weight lives in registers as 7 x 4 bf16x8 MFMA A
// fragments; for every 16-pixel tile and kernel row kh the B fragment is ONE 16-byte global load per lane:
// lane (i, g) needs taps 2g, 2g+1 of output pixel ow0+i = padded input columns 2*(ow0+i) + 2g, +1 (8 bf16).
__global__ __launch_bounds__(256) void stem_conv_kernel(const bf16_t* __restrict__ xp, const bf16_t* __restrict__ w,
                                                        const float* __restrict__ bias, bf16_t* __restrict__ out,
                                                        int B, int Hp, int Wp, int Ho, int Wo) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, lg = lane >> 4;
    // neighbouring output rows read overlapping input rows (7-row window, stride 2): contiguous runs of blocks per XCD
    const int row = rt_xcd_remap((int)blockIdx.x, (int)gridDim.x, 1) * 4 + wave;          // (b, oh)
    if (row >= B * Ho) return;
    const int b = row / Ho, oh = row % Ho;
    bf16x8 wf[7][4];
#pragma unroll
    for (int kh = 0; kh < 7; ++kh)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
            wf[kh][nt] = *reinterpret_cast<const bf16x8*>(w + ((size_t)(nt * 16 + li) * 7 + kh) * 32 + lg * 8);
    f32x4 bv[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) bv[nt] = *reinterpret_cast<const f32x4*>(bias + nt * 16 + lg * 4);
    const bf16_t* xrow = xp + ((size_t)b * Hp + 2 * oh) * Wp * 4;
    const int tiles = (Wo + 15) >> 4;
    for (int tile = 0; tile < tiles; ++tile) {
        const int ow = tile * 16 + li;
        f32x4 acc[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kh = 0; kh < 7; ++kh) {
            const bf16x8 xf = *reinterpret_cast<const bf16x8*>(xrow + ((size_t)kh * Wp + 2 * ow + 2 * lg) * 4);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kh][nt], xf, acc[nt], 0, 0, 0);
        }
        if (ow < Wo) {
            bf16_t* o = out + (((size_t)b * Ho + oh) * Wo + ow) * 64;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                bf16x4 ov;
#pragma unroll
                for (int r = 0; r < 4; ++r) ov[r] = (bf16_t)fmaxf(acc[nt][r] + bv[nt][r], 0.f);
                *reinterpret_cast<bf16x4*>(o + nt * 16 + lg * 4) = ov;
            }
        }
    }
}

// ---------------------------------------------------------------- max pool
__global__ __launch_bounds__(256) void maxpool_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y,
                                                      int B, int H, int W, int C, int Ho, int Wo) {
    const int cch = C / 8;
    const size_t total = (size_t)B * Ho * Wo * cch;
    // output rows oy and oy + 1 share an input row: contiguous runs of blocks per XCD keep those re-reads in one L2
    for (size_t i = rt_xcd_remap((int)blockIdx.x, (int)gridDim.x, 1) * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int cc = (int)(i % cch);
        size_t t = i / cch;
        const int ox = (int)(t % Wo); t /= Wo;
        const int oy = (int)(t % Ho);
        const int b = (int)(t / Ho);
        float m[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = -INFINITY;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int iy = oy * 2 - 1 + dy;
            if (iy < 0 || iy >= H) continue;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int ix = ox * 2 - 1 + dx;
                if (ix < 0 || ix >= W) continue;
                const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + (((size_t)b * H + iy) * W + ix) * C + cc * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) m[e] = fmaxf(m[e], (float)v[e]);
            }
        }
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16_t)m[e];
        *reinterpret_cast<bf16x8*>(y + i * 8) = o;
    }
}

// ---------------------------------------------------------------- stem conv + max pool in one launch
// A workgroup owns an 8 x 16 tile of POOLED pixels: its four waves compute the 17 x 33 stem pixels under it (36 MFMA pixel tiles
// of 16, nine per wave, weights in registers exactly as in stem_conv_kernel; the next tile's seven 16-byte loads are in flight
// under the 28 MFMAs of the current one), round them to bf16 into LDS (72 KB, 16-byte slots XOR-swizzled by the pixel), and pool
// from there: the 105 MB stem output of a 640 x 640 batch of 8 is neither written nor read back.  Same products, same K order,
// same rounding point as the two launches it replaces, and max commutes with the (monotonic) rounding: bit-identical.
// Values behind the ReLU are non-negative, so the maximum is taken on the bf16 bit patterns as signed 16-bit integers
// (a -0.0, pattern 0x8000, loses against everything else, as it may in fmaxf).
constexpr int SP_PH = 8, SP_PW = 16, SP_SH = 2 * SP_PH + 1, SP_SW = 2 * SP_PW + 1, SP_NPIX = SP_SH * SP_SW;     // 17 x 33 = 561
constexpr int SP_TILES = (SP_NPIX + 15) / 16;                                                                   // 36

__global__ __launch_bounds__(256, 2) void stem_pool_kernel(const bf16_t* __restrict__ xp, const bf16_t* __restrict__ w,
                                                           const float* __restrict__ bias, bf16_t* __restrict__ out,
                                                           int B, int Hp, int Wp, int Ho, int Wo, int Po, int Qo, int tiles_x, int tiles_y) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sp_smem[];
    bf16_t* sm = reinterpret_cast<bf16_t*>(sp_smem);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, lg = lane >> 4;
    // neighbouring tiles share a one-pixel stem halo and six input rows / columns: contiguous runs of blocks per XCD
    int blk = rt_xcd_remap((int)blockIdx.x, (int)gridDim.x, 1);
    const int tx = blk % tiles_x; blk /= tiles_x;
    const int ty = blk % tiles_y;
    const int b = blk / tiles_y;
    const int py0 = ty * SP_PH, px0 = tx * SP_PW;
    const int sy0 = 2 * py0 - 1, sx0 = 2 * px0 - 1;                    // stem pixel of tile-local (0, 0)
    bf16x8 wf[7][4];
#pragma unroll
    for (int kh = 0; kh < 7; ++kh)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
            wf[kh][nt] = *reinterpret_cast<const bf16x8*>(w + ((size_t)(nt * 16 + li) * 7 + kh) * 32 + lg * 8);
    f32x4 bv[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) bv[nt] = *reinterpret_cast<const f32x4*>(bias + nt * 16 + lg * 4);
    const bf16_t* ximg = xp + (size_t)b * Hp * Wp * 4;
    // stem pixels outside the image (the pool's padding, tile overhang) are computed at a clamped position and never read back
    auto src = [&](int t) -> const bf16_t* {
        int p = (wave + 4 * t) * 16 + li; p = p < SP_NPIX ? p : SP_NPIX - 1;
        int sy = sy0 + p / SP_SW, sx = sx0 + p % SP_SW;
        sy = sy < 0 ? 0 : (sy >= Ho ? Ho - 1 : sy);
        sx = sx < 0 ? 0 : (sx >= Wo ? Wo - 1 : sx);
        return ximg + ((size_t)(2 * sy) * Wp + 2 * sx + 2 * lg) * 4;
    };
    auto load = [&](bf16x8 (&x)[7], int t) {
        const bf16_t* s0 = src(t);
#pragma unroll
        for (int kh = 0; kh < 7; ++kh) x[kh] = *reinterpret_cast<const bf16x8*>(s0 + (size_t)kh * Wp * 4);
    };
    auto compute = [&](const bf16x8 (&x)[7], int t) {
        f32x4 acc[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kh = 0; kh < 7; ++kh)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kh][nt], x[kh], acc[nt], 0, 0, 0);
        const int p = (wave + 4 * t) * 16 + li;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            bf16x4 ov;
#pragma unroll
            for (int r = 0; r < 4; ++r) ov[r] = (bf16_t)fmaxf(acc[nt][r] + bv[nt][r], 0.f);
            const int slot = (nt * 2 + (lg >> 1)) ^ (p & 7);
            *reinterpret_cast<bf16x4*>(sm + (size_t)p * 64 + slot * 8 + (lg & 1) * 4) = ov;
        }
    };
    static_assert(SP_TILES % 4 == 0 && (SP_TILES / 4) % 2 == 1, "nine tiles per wave: four pairs and one");
    bf16x8 xa[7], xb[7];
    load(xa, 0);
#pragma unroll 1
    for (int t = 0; t < SP_TILES / 4 - 1; t += 2) {
        load(xb, t + 1);
        compute(xa, t);
        load(xa, t + 2);
        compute(xb, t + 1);
    }
    compute(xa, SP_TILES / 4 - 1);
    __syncthreads();
    typedef short s16x8 __attribute__((ext_vector_type(8)));
#pragma unroll
    for (int it = 0; it < SP_PH * SP_PW * 8 / 256; ++it) {
        const int item = it * 256 + (int)threadIdx.x;
        const int cs = item & 7, pp = item >> 3;
        const int pyl = pp / SP_PW, pxl = pp % SP_PW;
        const int py = py0 + pyl, px = px0 + pxl;
        if (py >= Po || px >= Qo) continue;
        // tile-local stem centre (2 pyl + 1, 2 pxl + 1) = stem pixel (2 py, 2 px): always inside the image
        const int cy = 2 * pyl + 1, cx = 2 * pxl + 1;
        const int pc = cy * SP_SW + cx;
        s16x8 m = *reinterpret_cast<const s16x8*>(sm + (size_t)pc * 64 + ((cs ^ (pc & 7)) * 8));
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy) {
            const int sy = 2 * py + dy;
            if (sy < 0 || sy >= Ho) continue;
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                const int sx = 2 * px + dx;
                if (sx < 0 || sx >= Wo || (dy == 0 && dx == 0)) continue;
                const int q = (cy + dy) * SP_SW + cx + dx;
                const s16x8 v = *reinterpret_cast<const s16x8*>(sm + (size_t)q * 64 + ((cs ^ (q & 7)) * 8));
                m = __builtin_elementwise_max(m, v);
            }
        }
        *reinterpret_cast<s16x8*>(out + (((size_t)b * Po + py) * Qo + px) * 64 + cs * 8) = m;
    }
}

// ---------------------------------------------------------------- weight prep
// src fp32 [N][T][C] -> dst bf16 [N][T][Cpad] (C padded with zeros; used by the stem with T=7x8 taps) and/or
// dst_t bf16 [C][T][N]; both multiplied by scale[n] when given.
__global__ __launch_bounds__(256) void weight_prep_kernel(const float* __restrict__ src, const float* __restrict__ scale,
                                                          bf16_t* __restrict__ dst, bf16_t* __restrict__ dst_t,
                                                          int N, int T, int C) {
    const size_t total = (size_t)N * T * C;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        const size_t r = i / C;
        const int t = (int)(r % T);
        const int n = (int)(r / T);
        float v = src[i];
        if (scale) v *= scale[n];
        const bf16_t bvv = (bf16_t)v;
        if (dst) dst[i] = bvv;
        if (dst_t) dst_t[((size_t)c * T + t) * N + n] = bvv;
    }
}

// One launch for all per-step operand refreshes: job table in device memory, 64(n) x 64(c) tiles per tap.  A thread moves 8
// consecutive elements: two 16-B fp32 loads, one 16-B bf16 store per copy (the element-per-thread version wrote 2 B per lane and
// ran at 3.4 TB/s); the transposed copy goes through a padded LDS tile and is written in 16-B pieces of the [C][T][N] rows.
// Jobs whose C (direct copy) or N (transposed copy) is not a multiple of 8, or whose rows are not 16-B aligned, take the
// element-wise path of the same tile.
__global__ __launch_bounds__(256) void weight_prep_batched_kernel(const int64_t* __restrict__ table, int njobs) {
    __shared__ float tile[64][65];
    int lo = 0, hi = njobs - 1;                       // last job whose first_tile <= blockIdx.x
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid * 8 + 7] <= (int64_t)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const int64_t* j = table + lo * 8;
    const float* src = reinterpret_cast<const float*>(j[0]);
    const float* scale = reinterpret_cast<const float*>(j[1]);
    bf16_t* dst = reinterpret_cast<bf16_t*>(j[2]);
    bf16_t* dst_t = reinterpret_cast<bf16_t*>(j[3]);
    const int N = (int)j[4], T = (int)j[5], C = (int)j[6];
    const int local = blockIdx.x - (int)j[7];
    const int ct = (C + 63) >> 6, nt = (N + 63) >> 6;
    const int tc = local % ct, tn = (local / ct) % nt, tap = local / (ct * nt);
    const bool vec_c = (C & 7) == 0 && ((uintptr_t)src & 15) == 0 && (!dst || ((uintptr_t)dst & 15) == 0);
    const bool vec_n = (N & 7) == 0 && dst_t && ((uintptr_t)dst_t & 15) == 0;
    if (vec_c) {
        const int r0 = threadIdx.x >> 3, c8 = (threadIdx.x & 7) * 8;       // 32 rows x 8 pieces per pass
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int nl = r0 + 32 * rr, n = tn * 64 + nl, c = tc * 64 + c8;
            float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (n < N && c < C) {
                const size_t o = ((size_t)n * T + tap) * C + c;
                const f32x4 a0 = *reinterpret_cast<const f32x4*>(src + o), a1 = *reinterpret_cast<const f32x4*>(src + o + 4);
                const float sc = scale ? scale[n] : 1.f;
                bf16x8 ov;
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = scale ? a0[e] * sc : a0[e]; v[4 + e] = scale ? a1[e] * sc : a1[e]; }
#pragma unroll
                for (int e = 0; e < 8; ++e) ov[e] = (bf16_t)v[e];
                if (dst) *reinterpret_cast<bf16x8*>(dst + o) = ov;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) tile[nl][c8 + e] = v[e];
        }
    } else {
        const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;          // 64 x 4
#pragma unroll 4
        for (int r = 0; r < 16; ++r) {
            const int n = tn * 64 + ty + r * 4, c = tc * 64 + tx;
            float v = 0.f;
            if (n < N && c < C) {
                v = src[((size_t)n * T + tap) * C + c];
                if (scale) v *= scale[n];
                if (dst) dst[((size_t)n * T + tap) * C + c] = (bf16_t)v;
            }
            tile[ty + r * 4][tx] = v;
        }
    }
    if (!dst_t) return;
    __syncthreads();
    if (vec_n) {
        const int r0 = threadIdx.x >> 3, n8 = (threadIdx.x & 7) * 8;
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int cl = r0 + 32 * rr, c = tc * 64 + cl, n = tn * 64 + n8;
            if (c < C && n < N) {
                bf16x8 ov;
#pragma unroll
                for (int e = 0; e < 8; ++e) ov[e] = (bf16_t)tile[n8 + e][cl];
                *reinterpret_cast<bf16x8*>(dst_t + ((size_t)c * T + tap) * N + n) = ov;
            }
        }
    } else {
        const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll 4
        for (int r = 0; r < 16; ++r) {
            const int c = tc * 64 + ty + r * 4, n = tn * 64 + tx;
            if (n < N && c < C) dst_t[((size_t)c * T + tap) * N + n] = (bf16_t)tile[tx][ty + r * 4];
        }
    }
}

// stem: src fp32 [64][7][7][3] (channels_last view of [64,3,7,7]) -> bf16 [64][7][8][4], zero padded
__global__ __launch_bounds__(256) void stem_weight_prep_kernel(const float* __restrict__ src, const float* __restrict__ scale,
                                                               bf16_t* __restrict__ dst) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 64 * 7 * 8 * 4) return;
    const int c = i & 3, kw = (i >> 2) & 7, kh = (i >> 5) % 7, n = i / 224;
    float v = 0.f;
    if (c < 3 && kw < 7) v = src[((n * 7 + kh) * 7 + kw) * 3 + c] * (scale ? scale[n] : 1.f);
    dst[i] = (bf16_t)v;
}

// FrozenBN fold: scale = w * rsqrt(rv + eps), shift = b - rm * scale (backbone.py:70-80)
__global__ void bn_fold_kernel(const float* w, const float* b, const float* rm, const float* rv, float eps,
                               float* scale, float* shift, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) { const float s = w[i] * rsqrtf(rv[i] + eps); scale[i] = s; shift[i] = b[i] - rm[i] * s; }
}

// ---------------------------------------------------------------- mask + position encoding
// One block per image.  mask_out[b][y][x] = mask[b][floor(y*H/h)][floor(x*W/w)] (nearest, backbone.py:107);
// pos = sine encoding of the cumulative not-mask counts (position_encoding.py:36-56) + add_vec[c]
// (level_embed[0] + token_type_emb[1], models/reftr.py:60,70-73), written token-major into the sequence.
__global__ __launch_bounds__(256) void mask_posenc_kernel(const rt_mask_posenc_desc p) {
    extern __shared__ float sm[];
    const int b = blockIdx.x;
    const int hw = p.h * p.w;
    unsigned char* sm_mask = reinterpret_cast<unsigned char*>(sm);             // [hw]
    float* ycum = sm + ((hw + 3) / 4);                                         // [hw]
    float* xcum = ycum + hw;                                                   // [hw]
    for (int i = threadIdx.x; i < hw; i += 256) {
        const int y = i / p.w, x = i % p.w;
        int sy = (int)floorf((float)y * ((float)p.H / (float)p.h));
        int sx = (int)floorf((float)x * ((float)p.W / (float)p.w));
        sy = min(sy, p.H - 1); sx = min(sx, p.W - 1);
        const unsigned char m = p.mask[((size_t)b * p.H + sy) * p.W + sx];
        sm_mask[i] = m;
        if (blockIdx.y == 0) p.kpm_out[(size_t)b * p.kpm_stride + p.kpm_off + i] = m;
    }
    __syncthreads();
    for (int x = threadIdx.x; x < p.w; x += 256) {
        float c = 0.f;
        for (int y = 0; y < p.h; ++y) { c += sm_mask[y * p.w + x] ? 0.f : 1.f; ycum[y * p.w + x] = c; }
    }
    for (int y = threadIdx.x; y < p.h; y += 256) {
        float c = 0.f;
        for (int x = 0; x < p.w; ++x) { c += sm_mask[y * p.w + x] ? 0.f : 1.f; xcum[y * p.w + x] = c; }
    }
    __syncthreads();
    const int npf = p.C / 2;
    const float scale = 6.283185307179586f, eps = 1e-6f;
    // the (pixel, channel) plane is split over gridDim.y workgroups (each recomputes the cheap cumulative sums)
    for (int i = blockIdx.y * 256 + threadIdx.x; i < hw * p.C; i += 256 * gridDim.y) {
        const int c = i % p.C, pix = i / p.C;
        const int y = pix / p.w, x = pix % p.w;
        float e;
        int cc;
        if (c < npf) { cc = c; e = (ycum[pix] - 0.5f) / (ycum[(p.h - 1) * p.w + x] + eps) * scale; }
        else         { cc = c - npf; e = (xcum[pix] - 0.5f) / (xcum[y * p.w + (p.w - 1)] + eps) * scale; }
        const float dim_t = powf(10000.f, (float)(2 * (cc / 2)) / (float)npf);
        const float a = e / dim_t;
        float v = (cc & 1) ? cosf(a) : sinf(a);
        if (p.add_vec) v += p.add_vec[c];
        p.pos_out[((size_t)b * p.pos_rows_per_img + p.pos_row_off + pix) * p.C + c] = v;
    }
}

}  // namespace

extern "C" int rt_img_pack(const float* img, void* out, int B, int H, int W, int Hp, int Wp, rt_stream_t stream) {
    if (!img || !out || Hp < H + 6 || Wp < W + 6) return RT_ERR_BADARG;
    const size_t total = (size_t)B * Hp * Wp;
    int blocks = (int)((total + 255) / 256); if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(img_pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, img, (bf16_t*)out, B, H, W, Hp, Wp);
    RT_CHECK_LAUNCH();
    return RT_OK;
}

extern "C" int rt_stem_conv(const void* xp, const void* w, const float* bias, void* out,
                            int B, int Hp, int Wp, int Ho, int Wo, rt_stream_t stream) {
    if (!xp || !w || !bias || !out) return RT_ERR_BADARG;
    // the last tile of a row reads padded columns up to 2*(16*ceil(Wo/16)-1) + 7
    if (Wp < 2 * (((Wo + 15) / 16) * 16) + 6 || Hp < 2 * (Ho - 1) + 7) return RT_ERR_BADARG;
    hipLaunchKernelGGL(stem_conv_kernel, dim3((B * Ho + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)xp, (const bf16_t*)w, bias, (bf16_t*)out, B, Hp, Wp, Ho, Wo);
    RT_CHECK_LAUNCH();
    return RT_OK;
}

extern "C" int rt_stem_pool(const void* xp, const void* w, const float* bias, void* out,
                            int B, int Hp, int Wp, int Ho, int Wo, rt_stream_t stream) {
    if (!xp || !w || !bias || !out || B <= 0 || Ho <= 0 || Wo <= 0) return RT_ERR_BADARG;
    if (Wp < 2 * Wo + 6 || Hp < 2 * (Ho - 1) + 7) return RT_ERR_BADARG;
    const int Po = (Ho - 1) / 2 + 1, Qo = (Wo - 1) / 2 + 1;
    const int tiles_x = (Qo + SP_PW - 1) / SP_PW, tiles_y = (Po + SP_PH - 1) / SP_PH;
    constexpr int lds = SP_TILES * 16 * 64 * 2;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)stem_pool_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(stem_pool_kernel, dim3(B * tiles_x * tiles_y), dim3(256), lds, (hipStream_t)stream,
                       (const bf16_t*)xp, (const bf16_t*)w, bias, (bf16_t*)out, B, Hp, Wp, Ho, Wo, Po, Qo, tiles_x, tiles_y);
    RT_CHECK_LAUNCH();
    return RT_OK;
}

extern "C" int rt_maxpool3x3s2(const void* x, void* y, int B, int H, int W, int C, int Ho, int Wo, rt_stream_t stream) {
    if (!x || !y || (C & 7)) return RT_ERR_BADARG;
    const size_t total = (size_t)B * Ho * Wo * (C / 8);
    int blocks = (int)((total + 255) / 256); if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(maxpool_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)x, (bf16_t*)y, B, H, W, C, Ho, Wo);
    RT_CHECK_LAUNCH();
    return RT_OK;
}

extern "C" int rt_weight_prep(const float* src, const float* scale, void* dst, void* dst_t, int N, int T, int C,
                              rt_stream_t stream) {
    if (!src || (!dst && !dst_t)) return RT_ERR_BADARG;
    const size_t total = (size_t)N * T * C;
    int blocks = (int)((total + 255) / 256); if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(weight_prep_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       src, scale, (bf16_t*)dst, (bf16_t*)dst_t, N, T, C);
    RT_CHECK_LAUNCH();
    return RT_OK;
}

extern "C" int rt_weight_prep_batched(const int64_t* table, int njobs, int total_tiles, rt_stream_t stream) {
    if (!table || njobs <= 0 || total_tiles <= 0) return RT_ERR_BADARG;
    hipLaunchKernelGGL(weight_prep_batched_kernel, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, table, njobs);
    RT_CHECK_LAUNCH();
    return RT_OK;
}

extern "C" int rt_stem_weight_prep(const float* src, const float* scale, void* dst, rt_stream_t stream) {
    if (!src || !dst) return RT_ERR_BADARG;
    hipLaunchKernelGGL(stem_weight_prep_kernel, dim3((64 * 224 + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       src, scale, (bf16_t*)dst);
    RT_CHECK_LAUNCH();
    return RT_OK;
}

extern "C" int rt_bn_fold(const float* w, const float* b, const float* rm, const float* rv, float eps,
                          float* scale, float* shift, int n, rt_stream_t stream) {
    if (!w || !b || !rm || !rv || !scale || !shift) return RT_ERR_BADARG;
    hipLaunchKernelGGL(bn_fold_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, b, rm, rv, eps, scale, shift, n);
    RT_CHECK_LAUNCH();
    return RT_OK;
}

extern "C" int rt_mask_posenc(const rt_mask_posenc_desc* d, rt_stream_t stream) {
    if (!d || !d->mask || !d->kpm_out || !d->pos_out || (d->C & 3)) return RT_ERR_BADARG;
    const int hw = d->h * d->w;
    const size_t smem = sizeof(float) * ((hw + 3) / 4 + 2 * (size_t)hw);
    if (smem > 64 * 1024) return RT_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(mask_posenc_kernel, dim3(d->B, 16), dim3(256), smem, (hipStream_t)stream, *d);
    RT_CHECK_LAUNCH();
    return RT_OK;
}
