// LayerNorm / GroupNorm forward + backward as wave-reduction kernels (fp32 statistics).
// One 64-lane wave owns one token row (D <= 1024); the epilogue emits everything the next GEMM needs
// (fp32 residual stream, bf16 operand copy, bf16 copy of y + pos) so no separate cast/add kernels run.
#include "rt_common.h"
#include <stdlib.h>

namespace {

constexpr int LN_MAX_PER_LANE = 16;   // D <= 1024
constexpr int GN_BWD_PIX = 32;        // pixels per workgroup of the backward statistics kernel (8 loads in flight per thread)

__device__ __forceinline__ int map_row(int r, int grp_rows, int grp_stride, int grp_off) {
    if (grp_rows > 0) return (r / grp_rows) * grp_stride + grp_off + (r % grp_rows);
    if (grp_rows < 0) return (r / (-grp_rows)) * grp_stride + grp_off;
    return r;
}

__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const rt_layernorm_desc p) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.M) return;
    const int D = p.D;
    const int nper = (D + 63) >> 6;
    const float* xr = p.x + (size_t)row * D;
    float v[LN_MAX_PER_LANE];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAX_PER_LANE; ++i) {
        const int c = lane + (i << 6);
        v[i] = (i < nper && c < D) ? xr[c] : 0.f;
        s += v[i];
    }
    const float mean = rt_wave_sum(s) / D;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAX_PER_LANE; ++i) {
        const int c = lane + (i << 6);
        const float d = (i < nper && c < D) ? v[i] - mean : 0.f;
        ss += d * d;
    }
    const float rstd = rsqrtf(rt_wave_sum(ss) / D + p.eps);
    if (lane == 0) { if (p.mean) p.mean[row] = mean; if (p.rstd) p.rstd[row] = rstd; }
    const int orow = map_row(row, p.grp_rows, p.grp_stride, p.grp_off);
    const bool do_drop = p.drop_p > 0.f;
    const uint32_t thresh = rt_drop_thresh(p.drop_p);
    const float ks = do_drop ? 1.f / (1.f - p.drop_p) : 1.f;
    bf16_t* yb = (bf16_t*)p.y_bf16;
    bf16_t* ypb = (bf16_t*)p.ypos_bf16;
#pragma unroll
    for (int i = 0; i < LN_MAX_PER_LANE; ++i) {
        const int c = lane + (i << 6);
        if (i < nper && c < D) {
            float y = (v[i] - mean) * rstd * p.gamma[c] + p.beta[c];
            if (p.act == RT_ACT_RELU) y = fmaxf(y, 0.f);
            if (do_drop) y = (rt_hash32(rt_site_seed(p.seed_dev, p.drop_seed), (uint32_t)(row * D + c)) >= thresh) ? y * ks : 0.f;
            const size_t o = (size_t)orow * D + c;
            if (p.y_f32) p.y_f32[o] = y;
            if (yb) yb[o] = (bf16_t)y;
            if (ypb) ypb[o] = (bf16_t)(y + p.pos[o]);
        }
    }
}

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma  (dy already through act/dropout)
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const rt_layernorm_bwd_desc p) {
    __shared__ float sm_g[4][1024];
    __shared__ float sm_b[4][1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int D = p.D;
    const int nper = (D + 63) >> 6;
    float dg[LN_MAX_PER_LANE], db[LN_MAX_PER_LANE];
#pragma unroll
    for (int i = 0; i < LN_MAX_PER_LANE; ++i) { dg[i] = 0.f; db[i] = 0.f; }
    const bool do_drop = p.drop_p > 0.f;
    const uint32_t thresh = rt_drop_thresh(p.drop_p);
    const float ks = do_drop ? 1.f / (1.f - p.drop_p) : 1.f;
    const bool do_drop2 = p.drop2_p > 0.f;
    const uint32_t thresh2 = rt_drop_thresh(p.drop2_p);
    const float ks2 = do_drop2 ? 1.f / (1.f - p.drop2_p) : 1.f;
    bf16_t* dxb = (bf16_t*)p.dx_bf16;

    for (int row = blockIdx.x * 4 + wave; row < p.M; row += gridDim.x * 4) {
        const int orow = map_row(row, p.grp_rows, p.grp_stride, p.grp_off);
        const float mean = p.mean[row], rstd = p.rstd[row];
        const float* xr = p.x + (size_t)row * D;
        float xh[LN_MAX_PER_LANE], g[LN_MAX_PER_LANE];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAX_PER_LANE; ++i) {
            const int c = lane + (i << 6);
            xh[i] = 0.f; g[i] = 0.f;
            if (i < nper && c < D) {
                const size_t o = (size_t)orow * D + c;
                float d = p.dy[o];
                if (p.dy2) d += p.dy2[o];
                xh[i] = (xr[c] - mean) * rstd;
                const float gam = p.gamma[c];
                if (do_drop) d = (rt_hash32(rt_site_seed(p.seed_dev, p.drop_seed), (uint32_t)(row * D + c)) >= thresh) ? d * ks : 0.f;
                if (p.act == RT_ACT_RELU) { if (xh[i] * gam + p.beta[c] <= 0.f) d = 0.f; }
                dg[i] += d * xh[i]; db[i] += d;
                g[i] = d * gam;
                s1 += g[i]; s2 += g[i] * xh[i];
            }
        }
        s1 = rt_wave_sum(s1) / D; s2 = rt_wave_sum(s2) / D;
#pragma unroll
        for (int i = 0; i < LN_MAX_PER_LANE; ++i) {
            const int c = lane + (i << 6);
            if (i < nper && c < D) {
                const float dx = rstd * (g[i] - s1 - xh[i] * s2);
                const size_t o = (size_t)row * D + c;
                if (p.dx_f32) p.dx_f32[o] = dx;
                if (dxb) {
                    float d2 = dx;
                    if (do_drop2) d2 = (rt_hash32(rt_site_seed(p.seed_dev, p.drop2_seed), (uint32_t)o) >= thresh2) ? dx * ks2 : 0.f;
                    dxb[o] = (bf16_t)d2;
                }
            }
        }
    }
    if (!p.dgamma && !p.dbeta && !p.partials) return;
#pragma unroll
    for (int i = 0; i < LN_MAX_PER_LANE; ++i) {
        const int c = lane + (i << 6);
        if (i < nper && c < D) { sm_g[wave][c] = dg[i]; sm_b[wave][c] = db[i]; }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += 256) {
        const float a = sm_g[0][c] + sm_g[1][c] + sm_g[2][c] + sm_g[3][c];
        const float b = sm_b[0][c] + sm_b[1][c] + sm_b[2][c] + sm_b[3][c];
        if (p.partials) {
            p.partials[((size_t)blockIdx.x * 2) * D + c] = a; p.partials[((size_t)blockIdx.x * 2 + 1) * D + c] = b;
        } else {
            if (p.dgamma) atomicAdd(p.dgamma + c, a);
            if (p.dbeta) atomicAdd(p.dbeta + c, b);
        }
    }
}

// ---------------- vectorised variants for D = 256 * V (V = 1: transformer width, V = 3: BERT width) ----------------
// Same arithmetic as the kernels above; lane l owns channels 4*(64*i + l) .. +3 of float4 group i, so every tensor row
// is moved with 16-byte accesses (8-byte for bf16) instead of four strided dword accesses per group.
template <int V>
__global__ __launch_bounds__(256) void layernorm_fwd_vec_kernel(const rt_layernorm_desc p) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.M) return;
    constexpr int D = 256 * V;
    const f32x4* xr = reinterpret_cast<const f32x4*>(p.x + (size_t)row * D);
    f32x4 v[V];
    // every load of the kernel is requested up front (one round trip): the affine parameters and the positional rows do not depend
    // on the statistics, but sit behind the mean / rstd stores in program order (possible aliasing), where the compiler leaves them
    const int orow = map_row(row, p.grp_rows, p.grp_stride, p.grp_off);
    f32x4 gam_[V], bet_[V], pos_[V];
#pragma unroll
    for (int i = 0; i < V; ++i) {
        const int c = (i * 64 + lane) * 4;
        v[i] = xr[i * 64 + lane];
        gam_[i] = *reinterpret_cast<const f32x4*>(p.gamma + c); bet_[i] = *reinterpret_cast<const f32x4*>(p.beta + c);
        pos_[i] = p.ypos_bf16 ? *reinterpret_cast<const f32x4*>(p.pos + (size_t)orow * D + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    const float mean = rt_wave_sum(s) * (1.f / D);
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mean; ss += d * d; }
    const float rstd = rsqrtf(rt_wave_sum(ss) * (1.f / D) + p.eps);
    if (lane == 0) { if (p.mean) p.mean[row] = mean; if (p.rstd) p.rstd[row] = rstd; }
    const bool do_drop = p.drop_p > 0.f;
    const uint32_t thresh = rt_drop_thresh(p.drop_p);
    const float ks = do_drop ? 1.f / (1.f - p.drop_p) : 1.f;
    const uint32_t seed = do_drop ? rt_site_seed(p.seed_dev, p.drop_seed) : 0u;
    bf16_t* yb = (bf16_t*)p.y_bf16;
    bf16_t* ypb = (bf16_t*)p.ypos_bf16;
#pragma unroll
    for (int i = 0; i < V; ++i) {
        const int c = (i * 64 + lane) * 4;
        const f32x4 gam = gam_[i], bet = bet_[i];
        f32x4 y;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            y[e] = (v[i][e] - mean) * rstd * gam[e] + bet[e];
            if (p.act == RT_ACT_RELU) y[e] = fmaxf(y[e], 0.f);
            if (do_drop) y[e] = (rt_hash32(seed, (uint32_t)(row * D + c + e)) >= thresh) ? y[e] * ks : 0.f;
        }
        const size_t o = (size_t)orow * D + c;
        if (p.y_f32) *reinterpret_cast<f32x4*>(p.y_f32 + o) = y;
        if (yb) {
            bf16x4 b;
#pragma unroll
            for (int e = 0; e < 4; ++e) b[e] = (bf16_t)y[e];
            *reinterpret_cast<bf16x4*>(yb + o) = b;
        }
        if (ypb) {
            const f32x4 ps = pos_[i];
            bf16x4 b;
#pragma unroll
            for (int e = 0; e < 4; ++e) b[e] = (bf16_t)(y[e] + ps[e]);
            *reinterpret_cast<bf16x4*>(ypb + o) = b;
        }
    }
}

template <int V>
__global__ __launch_bounds__(256) void layernorm_bwd_vec_kernel(const rt_layernorm_bwd_desc p) {
    constexpr int D = 256 * V;
    __shared__ float sm_g[4][D];
    __shared__ float sm_b[4][D];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 dg[V], db[V], gam[V], bet[V];
#pragma unroll
    for (int i = 0; i < V; ++i) {
        dg[i] = f32x4{0.f, 0.f, 0.f, 0.f}; db[i] = dg[i];
        gam[i] = *reinterpret_cast<const f32x4*>(p.gamma + (i * 64 + lane) * 4);
        bet[i] = p.beta ? *reinterpret_cast<const f32x4*>(p.beta + (i * 64 + lane) * 4) : dg[i];
    }
    const bool do_drop = p.drop_p > 0.f;
    const uint32_t thresh = rt_drop_thresh(p.drop_p);
    const float ks = do_drop ? 1.f / (1.f - p.drop_p) : 1.f;
    const bool do_drop2 = p.drop2_p > 0.f;
    const uint32_t thresh2 = rt_drop_thresh(p.drop2_p);
    const float ks2 = do_drop2 ? 1.f / (1.f - p.drop2_p) : 1.f;
    const uint32_t seed = do_drop ? rt_site_seed(p.seed_dev, p.drop_seed) : 0u;
    const uint32_t seed2 = do_drop2 ? rt_site_seed(p.seed_dev, p.drop2_seed) : 0u;
    bf16_t* dxb = (bf16_t*)p.dx_bf16;

    for (int row = blockIdx.x * 4 + wave; row < p.M; row += gridDim.x * 4) {
#pragma clang fp contract(off)      // rt_decoder_bwd repeats this arithmetic and must round the same way: no fused multiply-adds
        const int orow = map_row(row, p.grp_rows, p.grp_stride, p.grp_off);
        const float mean = p.mean[row], rstd = p.rstd[row];
        const f32x4* xr = reinterpret_cast<const f32x4*>(p.x + (size_t)row * D);
        const f32x4* dyr = reinterpret_cast<const f32x4*>(p.dy + (size_t)orow * D);
        const f32x4* dy2r = p.dy2 ? reinterpret_cast<const f32x4*>(p.dy2 + (size_t)orow * D) : nullptr;
        f32x4 xh[V], g[V];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < V; ++i) {
            f32x4 d = dyr[i * 64 + lane];
            if (dy2r) d += dy2r[i * 64 + lane];
            const f32x4 xv = xr[i * 64 + lane];
            const int c = (i * 64 + lane) * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                xh[i][e] = (xv[e] - mean) * rstd;
                float de = d[e];
                if (do_drop) de = (rt_hash32(seed, (uint32_t)(row * D + c + e)) >= thresh) ? de * ks : 0.f;
                if (p.act == RT_ACT_RELU) { if (xh[i][e] * gam[i][e] + bet[i][e] <= 0.f) de = 0.f; }
                dg[i][e] += de * xh[i][e]; db[i][e] += de;
                g[i][e] = de * gam[i][e];
                s1 += g[i][e]; s2 += g[i][e] * xh[i][e];
            }
        }
        s1 = rt_wave_sum(s1) * (1.f / D); s2 = rt_wave_sum(s2) * (1.f / D);
#pragma unroll
        for (int i = 0; i < V; ++i) {
            const int c = (i * 64 + lane) * 4;
            const size_t o = (size_t)row * D + c;
            f32x4 dx;
#pragma unroll
            for (int e = 0; e < 4; ++e) dx[e] = rstd * (g[i][e] - s1 - xh[i][e] * s2);
            if (p.dx_f32) *reinterpret_cast<f32x4*>(p.dx_f32 + o) = dx;
            if (dxb) {
                bf16x4 b;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float d2 = dx[e];
                    if (do_drop2) d2 = (rt_hash32(seed2, (uint32_t)(o + e)) >= thresh2) ? d2 * ks2 : 0.f;
                    b[e] = (bf16_t)d2;
                }
                *reinterpret_cast<bf16x4*>(dxb + o) = b;
            }
        }
    }
    if (!p.dgamma && !p.dbeta && !p.partials) return;
#pragma unroll
    for (int i = 0; i < V; ++i) {
        *reinterpret_cast<f32x4*>(&sm_g[wave][(i * 64 + lane) * 4]) = dg[i];
        *reinterpret_cast<f32x4*>(&sm_b[wave][(i * 64 + lane) * 4]) = db[i];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += 256) {
        const float a = sm_g[0][c] + sm_g[1][c] + sm_g[2][c] + sm_g[3][c];
        const float b = sm_b[0][c] + sm_b[1][c] + sm_b[2][c] + sm_b[3][c];
        if (p.partials) {
            p.partials[((size_t)blockIdx.x * 2) * D + c] = a; p.partials[((size_t)blockIdx.x * 2 + 1) * D + c] = b;
        } else {
            if (p.dgamma) atomicAdd(p.dgamma + c, a);
            if (p.dbeta) atomicAdd(p.dbeta + c, b);
        }
    }
}

// ---------------- GroupNorm over token-major images x[b][p][c], groups of C/G channels ----------------
// part[b][chunk][g] = {sum, sumsq} of one pixel chunk, one thread per channel; the apply kernel adds the chunks in a fixed
// order (no atomics: the forward is bit-reproducible run to run).
__global__ __launch_bounds__(256) void gn_stats_kernel(const float* __restrict__ x, float* __restrict__ part,
                                                       int HW, int C, int G, int pix_per_chunk) {
    const int b = blockIdx.y, c = threadIdx.x;
    const int p0 = blockIdx.x * pix_per_chunk;
    const int p1 = min(p0 + pix_per_chunk, HW);
    float s = 0.f, ss = 0.f;
    if (c < C)
        for (int pix = p0; pix < p1; ++pix) { const float v = x[((size_t)b * HW + pix) * C + c]; s += v; ss += v * v; }
    const int cpg = C / G;             // channels per group (power of two <= 64 assumed)
    for (int o = cpg >> 1; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); ss += __shfl_xor(ss, o, 64); }
    if (c < C && (c % cpg) == 0) {
        float* dst = part + (((size_t)b * gridDim.x + blockIdx.x) * G + c / cpg) * 2;
        dst[0] = s; dst[1] = ss;
    }
}

// grid (row blocks, B): every block first reduces its image's chunk partials (wave per group, xor butterfly: the same
// value in every block), block x = 0 keeps the sums for backward, then rows blockIdx.x, + gridDim.x, ... are normalised.
__global__ __launch_bounds__(256) void gn_apply_kernel(const rt_groupnorm_desc p) {
    __shared__ float sm[256][2];
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cpg = p.C / p.G;
    // 8 threads per group (G <= 32) take the chunks round-robin, then a fixed 3-step butterfly inside each 8-lane group:
    // one round of loads instead of a serial walk over the groups (this prologue runs in every workgroup)
    if (p.G <= 32) {
        const int g = tid >> 3, sub = tid & 7;
        float s = 0.f, ss = 0.f;
        if (g < p.G)
            for (int ch = sub; ch < p.chunks; ch += 8) {
                const float* q = p.partials + (((size_t)b * p.chunks + ch) * p.G + g) * 2;
                s += q[0]; ss += q[1];
            }
        for (int o = 4; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); ss += __shfl_xor(ss, o, 64); }
        if (g < p.G && sub == 0) {
            sm[g][0] = s; sm[g][1] = ss;
            if (blockIdx.x == 0) { p.stats[((size_t)b * p.G + g) * 2] = s; p.stats[((size_t)b * p.G + g) * 2 + 1] = ss; }
        }
    } else
    for (int g = wave; g < p.G; g += 4) {
        float s = 0.f, ss = 0.f;
        for (int ch = lane; ch < p.chunks; ch += 64) {
            const float* q = p.partials + (((size_t)b * p.chunks + ch) * p.G + g) * 2;
            s += q[0]; ss += q[1];
        }
        for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); ss += __shfl_xor(ss, o, 64); }
        if (lane == 0) {
            sm[g][0] = s; sm[g][1] = ss;
            if (blockIdx.x == 0) { p.stats[((size_t)b * p.G + g) * 2] = s; p.stats[((size_t)b * p.G + g) * 2 + 1] = ss; }
        }
    }
    __syncthreads();
    const int c = tid;
    if (c >= p.C) return;
    const float inv_n = 1.f / (float)(cpg * p.HW);
    const float mean = sm[c / cpg][0] * inv_n;
    const float var = fmaxf(sm[c / cpg][1] * inv_n - mean * mean, 0.f);
    const float rstd = rsqrtf(var + p.eps);
    bf16_t* yb = (bf16_t*)p.y_bf16; bf16_t* ypb = (bf16_t*)p.ypos_bf16;
    for (int pix = blockIdx.x; pix < p.HW; pix += gridDim.x) {
        const size_t i = ((size_t)b * p.HW + pix) * p.C + c;
        const float y = (p.x[i] - mean) * rstd * p.gamma[c] + p.beta[c];
        const size_t o = ((size_t)b * p.out_rows_per_img + p.out_row_off + pix) * p.C + c;
        if (p.y_f32) p.y_f32[o] = y;
        if (yb) yb[o] = (bf16_t)y;
        if (ypb) ypb[o] = (bf16_t)(y + p.pos[o]);
    }
}

// backward pass 1: per (b, g): sums of g=dy*gamma and g*xhat; per channel dgamma/dbeta
__global__ __launch_bounds__(256) void gn_bwd_stats_kernel(const rt_groupnorm_bwd_desc p) {
    const int b = blockIdx.y, c = threadIdx.x;
    const int p0 = blockIdx.x * GN_BWD_PIX, p1 = min(p0 + GN_BWD_PIX, p.HW);
    const int cpg = p.C / p.G;
    const float inv_n = 1.f / (float)(cpg * p.HW);
    float s1 = 0.f, s2 = 0.f, dg = 0.f, db = 0.f;
    if (c < p.C) {
        const float* st = p.stats + ((size_t)b * p.G + c / cpg) * 2;
        const float mean = st[0] * inv_n;
        const float rstd = rsqrtf(fmaxf(st[1] * inv_n - mean * mean, 0.f) + p.eps);
        const float gam = p.gamma[c];
        // 8 pixels' loads in flight at a time (the plain loop is a chain of load -> accumulate round trips), 32 pixels per workgroup:
        // a quarter of the per-channel atomics of the 8-pixel blocking (round 3: 22.7 -> see profiles/r03 kernel stats)
        for (int q0 = p0; q0 < p1; q0 += 8) {
            float dv[8], xv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int pix = q0 + j < p1 ? q0 + j : p1 - 1;
                const size_t o = ((size_t)b * p.out_rows_per_img + p.out_row_off + pix) * p.C + c;
                dv[j] = p.dy[o];
                if (p.dy2) dv[j] += p.dy2[o];
                xv[j] = p.x[((size_t)b * p.HW + pix) * p.C + c];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (q0 + j >= p1) continue;
                const float d = dv[j];
                const float xh = (xv[j] - mean) * rstd;
                dg += d * xh; db += d;
                s1 += d * gam; s2 += d * gam * xh;
            }
        }
    }
    for (int o = cpg >> 1; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
    if (c < p.C) {
        if ((c % cpg) == 0) {
            atomicAdd(p.bstats + ((size_t)b * p.G + c / cpg) * 2, s1);
            atomicAdd(p.bstats + ((size_t)b * p.G + c / cpg) * 2 + 1, s2);
        }
        if (p.dgamma) atomicAdd(p.dgamma + c, dg);
        if (p.dbeta) atomicAdd(p.dbeta + c, db);
    }
}

__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const rt_groupnorm_bwd_desc p) {
    const size_t total = (size_t)p.B * p.HW * p.C;
    const int cpg = p.C / p.G;
    const float inv_n = 1.f / (float)(cpg * p.HW);
    bf16_t* dxb = (bf16_t*)p.dx_bf16;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % p.C);
        const size_t pixg = i / p.C;
        const int b = (int)(pixg / p.HW);
        const int pix = (int)(pixg % p.HW);
        const int g = c / cpg;
        const float* st = p.stats + ((size_t)b * p.G + g) * 2;
        const float mean = st[0] * inv_n;
        const float rstd = rsqrtf(fmaxf(st[1] * inv_n - mean * mean, 0.f) + p.eps);
        const float xh = (p.x[i] - mean) * rstd;
        const size_t o = ((size_t)b * p.out_rows_per_img + p.out_row_off + pix) * p.C + c;
        float d = p.dy[o];
        if (p.dy2) d += p.dy2[o];
        const float m1 = p.bstats[((size_t)b * p.G + g) * 2] * inv_n;
        const float m2 = p.bstats[((size_t)b * p.G + g) * 2 + 1] * inv_n;
        const float dx = rstd * (d * p.gamma[c] - m1 - xh * m2);
        if (p.dx_f32) p.dx_f32[i] = dx;
        if (dxb) dxb[i] = (bf16_t)dx;
    }
}

struct LnPgJobs { rt_ln_pg_job j[64]; int first[65]; int n; };
// one workgroup = 64 channels of one job x 4 row lanes (4 loads in flight each), combined through LDS
__global__ __launch_bounds__(256) void ln_param_grad_grouped_kernel(const LnPgJobs p) {
    __shared__ float red[2][3][64];
    int lo = 0, hi = p.n - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (p.first[mid] <= (int)blockIdx.x) lo = mid; else hi = mid - 1; }
    const rt_ln_pg_job& q = p.j[lo];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = ((int)blockIdx.x - p.first[lo]) * 64 + tx;
    const bool ok = c < q.D;
    float a = 0.f, b = 0.f;
    if (ok) {
        int r = ty;
        for (; r + 12 < q.n_blocks; r += 16) {
            const float* p0 = q.partials + ((size_t)r * 2) * q.D + c;
            const size_t st = (size_t)8 * q.D;               // 4 partial rows (x2 vectors) ahead
            const float a0 = p0[0], b0 = p0[q.D], a1 = p0[st], b1 = p0[st + q.D], a2 = p0[2 * st], b2 = p0[2 * st + q.D],
                        a3 = p0[3 * st], b3 = p0[3 * st + q.D];
            a += (a0 + a1) + (a2 + a3); b += (b0 + b1) + (b2 + b3);
        }
        for (; r < q.n_blocks; r += 4) { a += q.partials[((size_t)r * 2) * q.D + c]; b += q.partials[((size_t)r * 2 + 1) * q.D + c]; }
    }
    if (ty) { red[0][ty - 1][tx] = a; red[1][ty - 1][tx] = b; }
    __syncthreads();
    if (ty || !ok) return;
    a += red[0][0][tx] + red[0][1][tx] + red[0][2][tx]; b += red[1][0][tx] + red[1][1][tx] + red[1][2][tx];
    if (q.dgamma) atomicAdd(q.dgamma + c, a);        // atomics: one LayerNorm (decoder.norm) can appear several times in a group
    if (q.dbeta) atomicAdd(q.dbeta + c, b);
}

}  // namespace

extern "C" int rt_layernorm_fwd(const rt_layernorm_desc* d, rt_stream_t stream) {
    if (!d || !d->x || !d->gamma || !d->beta) return RT_ERR_BADARG;
    if (d->D <= 0 || d->D > 64 * LN_MAX_PER_LANE || d->M <= 0) return RT_ERR_UNSUPPORTED;
    if (d->ypos_bf16 && !d->pos) return RT_ERR_BADARG;
    static const int vec = RT_TUNE("REFTR_LNVEC", 1);
    const dim3 grid((d->M + 3) / 4);
    if (vec && d->D == 256)      hipLaunchKernelGGL(layernorm_fwd_vec_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, *d);
    else if (vec && d->D == 768) hipLaunchKernelGGL(layernorm_fwd_vec_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, *d);
    else                         hipLaunchKernelGGL(layernorm_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, *d);
    RT_CHECK_LAUNCH();
    return RT_OK;
}

extern "C" int rt_layernorm_bwd(const rt_layernorm_bwd_desc* d, rt_stream_t stream) {
    if (!d || !d->dy || !d->x || !d->gamma || !d->mean || !d->rstd) return RT_ERR_BADARG;
    if (d->D <= 0 || d->D > 64 * LN_MAX_PER_LANE || d->M <= 0) return RT_ERR_UNSUPPORTED;
    if (d->act == RT_ACT_RELU && !d->beta) return RT_ERR_BADARG;
    // few, fat workgroups: every block ends with D atomics per parameter vector, so contention (not bandwidth)
    // is what scales with the block count
    static const int lnb = RT_TUNE("REFTR_LNB", 256);   // A/B on the step: 64..1024, the per-block dgamma/dbeta atomics dominate
    int blocks = (d->M + 3) / 4;
    if (blocks > lnb) blocks = lnb;
    if (!d->dgamma && !d->dbeta && !d->partials) { blocks = (d->M + 3) / 4; if (blocks > 1024) blocks = 1024; }
    static const int lnpb = RT_TUNE("REFTR_LNPB", 880);
    if (d->partials) { blocks = (d->M + 3) / 4; if (blocks > lnpb) blocks = lnpb; }
    if (d->partials && d->n_blocks_out) *d->n_blocks_out = blocks;
    static const int vec = RT_TUNE("REFTR_LNVEC", 1);
    if (vec && d->D == 256)      hipLaunchKernelGGL(layernorm_bwd_vec_kernel<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, *d);
    else if (vec && d->D == 768) hipLaunchKernelGGL(layernorm_bwd_vec_kernel<3>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, *d);
    else                         hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, *d);
    RT_CHECK_LAUNCH();
    return RT_OK;
}

extern "C" int rt_groupnorm_fwd(const rt_groupnorm_desc* d, rt_stream_t stream) {
    if (!d || !d->x || !d->gamma || !d->beta || !d->stats) return RT_ERR_BADARG;
    if (d->C <= 0 || d->C > 256 || d->G <= 0 || (d->C % d->G) || (d->C / d->G) > 64 || ((d->C / d->G) & (d->C / d->G - 1)))
        return RT_ERR_UNSUPPORTED;
    if (!d->partials || d->chunks < 1 || d->chunks > 64) return RT_ERR_BADARG;
    hipStream_t s = (hipStream_t)stream;
    const int ppc = (d->HW + d->chunks - 1) / d->chunks;          // chunks past the last pixel write zeros
    hipLaunchKernelGGL(gn_stats_kernel, dim3(d->chunks, d->B), dim3(256), 0, s, d->x, d->partials, d->HW, d->C, d->G, ppc);
    RT_CHECK_LAUNCH();
    int rows = d->HW < 64 ? d->HW : 64;
    hipLaunchKernelGGL(gn_apply_kernel, dim3(rows, d->B), dim3(256), 0, s, *d);
    RT_CHECK_LAUNCH();
    return RT_OK;
}

extern "C" int rt_groupnorm_bwd(const rt_groupnorm_bwd_desc* d, rt_stream_t stream) {
    if (!d || !d->x || !d->dy || !d->gamma || !d->stats || !d->bstats) return RT_ERR_BADARG;
    if (d->C <= 0 || d->C > 256 || d->G <= 0 || (d->C % d->G) || (d->C / d->G) > 64) return RT_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = rt_zero_f32(d->bstats, 2 * (size_t)d->B * d->G, s);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(gn_bwd_stats_kernel, dim3((d->HW + GN_BWD_PIX - 1) / GN_BWD_PIX, d->B), dim3(256), 0, s, *d);
    RT_CHECK_LAUNCH();
    const size_t total = (size_t)d->B * d->HW * d->C;
    int blocks = (int)((total + 255) / 256); if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(blocks), dim3(256), 0, s, *d);
    RT_CHECK_LAUNCH();
    return RT_OK;
}

extern "C" int rt_ln_param_grad_grouped(const rt_ln_pg_job* jobs, int n, rt_stream_t stream) {
    if (!jobs || n <= 0) return RT_ERR_BADARG;
    for (int base = 0; base < n; base += 64) {
        LnPgJobs p;
        p.n = n - base < 64 ? n - base : 64;
        int blocks = 0;
        for (int i = 0; i < p.n; ++i) {
            const rt_ln_pg_job& q = jobs[base + i];
            if (!q.partials || q.n_blocks <= 0 || q.D <= 0) return RT_ERR_BADARG;
            p.j[i] = q; p.first[i] = blocks; blocks += (q.D + 63) / 64;
        }
        p.first[p.n] = blocks;
        hipLaunchKernelGGL(ln_param_grad_grouped_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
        RT_CHECK_LAUNCH();
    }
    return RT_OK;
}
