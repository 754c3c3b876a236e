// The "few rows" region between the encoder's forward and the encoder's backward -- QueryEncoder, box head, box loss and their
// backward (models/reftr_transformer.py:41-66,287; models/modeling/backbone.py:26-38; models/criterion.py:113-153) -- as THREE
// launches instead of ~75.  Every operation here works on B * n_phrase (8 ... 128) rows of 256 features: as separate launches each
// costs a ~4.6 us graph node for < 1 us of work and the chip idles (profiles/r04i_concurrent_timeline.txt: 0.69 ms of the step).
//
//   rt_qenc_fwd    one workgroup per image: linear1/2/3 -> CLS-key attention -> context_out (Linear + LN) + residual -> concat with
//                  the mapped phrase feature -> fuse_encoder_query (Linear-LN-ReLU-Dropout-Linear-LN-ReLU) -> + query_embed
//   rt_head_loss   one workgroup per decoder layer: decoder.norm -> bbox MLP -> box loss + d total / d logits -> the MLP's and the
//                  norm's backward-data (everything between rt_decoder_fwd and rt_decoder_bwd)
//   rt_qenc_bwd    one workgroup per image: the backward-data chain of rt_qenc_fwd down to d memory
//
// A workgroup never waits for another one: stages are separated by __syncthreads() only, intermediate tensors go through global
// memory (they are the tensors the backward / the weight-gradient launches need anyway), the products run on MFMA
// (v_mfma_f32_16x16x32_bf16) with both operands fetched straight from global memory in fragment order -- at <= 48 rows there is
// nothing to reuse through LDS.  Rounding points (which tensors are bf16, where fp32) are those of the launches these kernels
// replace: rt_conv_gemm (bf16 operands, fp32 accumulate, fp32 bias), rt_layernorm_*, rt_qenc_attn_*, rt_small_dgrad, rt_box_loss;
// only the accumulation ORDER of the products differs (tests/test_qregion_gpu.py: fused vs launched, <= 2e-6 relative).
// Weight gradients stay with the grouped launches (rt_small_wgrad_grouped / rt_conv_wgrad_grouped): the kernels emit the bf16 dy
// operands those read.
#include "rt_common.h"
#include "rt_loss_row.h"

namespace {

constexpr int QE = 256;          // hidden size these kernels are built for (every reference config: --hidden_dim 256)
constexpr int QT = 1024;         // threads per workgroup (16 waves: one 16-feature tile of a 256-wide product per wave)
constexpr int QH = QT / 256;     // 256-thread groups
constexpr int QW = QT / 64;

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, const f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ void st4b(bf16_t* p, const f32x4 v) {
    bf16x4 b;
#pragma unroll
    for (int e = 0; e < 4; ++e) b[e] = (bf16_t)v[e];
    *reinterpret_cast<bf16x4*>(p) = b;
}

// One 16-feature tile n0 .. n0+15 of Y[m][n] = sum_k X[m][k] W[n][k] for up to MT * 16 rows; run by ONE wave.
// W: bf16 [nmax][ldw] row-major; xrow(m) -> the bf16 row m of X (K contiguous values).  The weight fragment is the A operand, so a
// lane ends up with 4 CONSECUTIVE features (n0 + (lane >> 4) * 4 ...) of row m = mt * 16 + (lane & 15): epi(m, n, acc) gets them.
template <int MT, int K, typename XRow, typename Epi>
__device__ __forceinline__ void q_tile(const bf16_t* __restrict__ W, const int ldw, const int n0, const int nmax,
                                       XRow xrow, const int M, Epi epi) {
    const int lane = threadIdx.x & 63, li = lane & 15, lg = lane >> 4;
    int n = n0 + li; if (n >= nmax) n = nmax - 1;
    const bf16_t* wp = W + (size_t)n * ldw + lg * 8;
    const bf16_t* xp[MT];
    f32x4 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        int m = t * 16 + li; if (m >= M) m = M - 1;
        xp[t] = xrow(m) + lg * 8;
        acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // The launch is latency-bound (8 ... 16 workgroups on the chip, cold weights): every fragment of a 256-deep K chunk is requested
    // before the first MFMA -- one memory round trip per chunk instead of one per K step (the scheduler would otherwise sink each
    // load next to its use: 2 loads in flight, measured 86 us for the forward launch).
    constexpr int KC = MT >= 3 ? 128 : (K < 256 ? K : 256), NK = KC / 32;      // <= 128 registers per lane at 16 waves per workgroup
#pragma unroll
    for (int k0 = 0; k0 < K; k0 += KC) {
        bf16x8 a[NK], x[MT][NK];
#pragma unroll
        for (int i = 0; i < NK; ++i) a[i] = *reinterpret_cast<const bf16x8*>(wp + k0 + i * 32);
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int i = 0; i < NK; ++i) x[t][i] = *reinterpret_cast<const bf16x8*>(xp[t] + k0 + i * 32);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NK; ++i)
#pragma unroll
            for (int t = 0; t < MT; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], x[t][i], acc[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int m = t * 16 + li;
        if (m < M && n0 + lg * 4 < nmax) epi(m, n0 + lg * 4, acc[t]);
    }
}
constexpr int QLMAX = 96;                 // language tokens per image the LDS operand images are sized for
constexpr int QLD = QE + 8;               // LDS row stride of a bf16 operand (16 bytes of padding: rows 528 B apart spread the banks)
// Stage stamps of workgroup 0 (100 MHz wall clock), one row per kernel: benchmarks/qregion_trace.py reads them through rt_qregion_trace.
__device__ unsigned long long g_qtrace[3][24];
__device__ __forceinline__ void q_stamp(const int kernel, int& slot) {
    if (blockIdx.x == 0 && threadIdx.x == 0 && slot < 24) g_qtrace[kernel][slot] = __builtin_amdgcn_s_memrealtime();
    ++slot;
}
// The same tile with the X operand in LDS (rows of ldx bf16 values, 16-byte aligned): the weight fragments of the whole K extent and
// the bias piece are requested once, then every 16-row group of X is read from LDS and multiplied -- a compute unit's fill path
// delivers ~20-40 KB/us (stage stamps, benchmarks/qregion_trace.py), so an operand that all 16 waves re-read must not come through it.
template <int K, typename Epi>
__device__ __forceinline__ void q_tile_lds(const bf16_t* __restrict__ W, const int ldw, const int n0, const int nmax,
                                           const bf16_t* Xs, const int ldx, const int M, const float* __restrict__ bias, Epi epi) {
    const int lane = threadIdx.x & 63, li = lane & 15, lg = lane >> 4;
    int n = n0 + li; if (n >= nmax) n = nmax - 1;
    const bf16_t* wp = W + (size_t)n * ldw + lg * 8;
    constexpr int NK = K / 32;
    bf16x8 a[NK];
#pragma unroll
    for (int i = 0; i < NK; ++i) a[i] = *reinterpret_cast<const bf16x8*>(wp + i * 32);
    const bool nok = n0 + lg * 4 < nmax;
    const f32x4 b4 = (bias && nok) ? ld4(bias + n0 + lg * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    for (int m0 = 0; m0 < M; m0 += 16) {
        int m = m0 + li; if (m >= M) m = M - 1;
        const bf16_t* xp = Xs + (size_t)m * ldx + lg * 8;
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < NK; ++i)
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], *reinterpret_cast<const bf16x8*>(xp + i * 32), acc, 0, 0, 0);
        if (m0 + li < M && nok) epi(m0 + li, n0 + lg * 4, acc + b4);
    }
}
// All N features of a product with M rows, the 16-feature tiles dealt round-robin to the workgroup's waves; rows in groups of 48.
template <int K, typename XRow, typename Epi>
__device__ __forceinline__ void q_gemm(const bf16_t* __restrict__ W, const int ldw, const int N, XRow xrow, const int M, Epi epi,
                                       const int tile0 = 0, const int tile_stride = QW) {
    const int wave = threadIdx.x >> 6;
    const int ntiles = (N + 15) >> 4;
    for (int t = tile0 + wave; t < ntiles; t += tile_stride) {
        for (int m0 = 0; m0 < M; m0 += 48) {
            const int rows = min(48, M - m0);
            auto xr = [&](int m) { return xrow(m0 + m); };
            auto ep = [&](int m, int n, const f32x4& v) { epi(m0 + m, n, v); };
            if (rows <= 16)      q_tile<1, K>(W, ldw, t * 16, N, xr, rows, ep);
            else if (rows <= 32) q_tile<2, K>(W, ldw, t * 16, N, xr, rows, ep);
            else                 q_tile<3, K>(W, ldw, t * 16, N, xr, rows, ep);
        }
    }
}

// LayerNorm of one 256-feature row by one wave (lane: features 4 lane .. 4 lane + 3): rt_layernorm_fwd's arithmetic
__device__ __forceinline__ f32x4 ln_row(const f32x4 v, const f32x4 gam, const f32x4 bet, const float eps, const bool relu,
                                        float& mean, float& rstd) {
    const float s = (v[0] + v[1]) + (v[2] + v[3]);
    mean = rt_wave_sum(s) * (1.f / QE);
    float ss = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) { const float d = v[e] - mean; ss += d * d; }
    rstd = rsqrtf(rt_wave_sum(ss) * (1.f / QE) + eps);
    f32x4 y;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        y[e] = (v[e] - mean) * rstd * gam[e] + bet[e];
        if (relu) y[e] = fmaxf(y[e], 0.f);
    }
    return y;
}
// its backward (rt_layernorm_bwd's arithmetic, un-contracted like there): d = dy of the row, returns dx; dg / db accumulate the
// row's contribution to d gamma / d beta.  drop: the forward's dropout mask (behind the ReLU) regenerated from (seed, row * 256 + c).
__device__ __forceinline__ f32x4 ln_row_bwd(const f32x4 d, const f32x4 xv, const float mean, const float rstd, const f32x4 gam,
                                            const f32x4 bet, const bool relu, const bool do_drop, const uint32_t seed,
                                            const uint32_t thresh, const float ks, const uint32_t idx0, f32x4& dg, f32x4& db) {
#pragma clang fp contract(off)
    f32x4 xh, g;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        xh[e] = (xv[e] - mean) * rstd;
        float de = d[e];
        if (do_drop) de = (rt_hash32(seed, idx0 + (uint32_t)e) >= thresh) ? de * ks : 0.f;
        if (relu) { if (xh[e] * gam[e] + bet[e] <= 0.f) de = 0.f; }
        dg[e] += de * xh[e]; db[e] += de;
        g[e] = de * gam[e];
        s1 += g[e]; s2 += g[e] * xh[e];
    }
    s1 = rt_wave_sum(s1) * (1.f / QE); s2 = rt_wave_sum(s2) * (1.f / QE);
    f32x4 dx;
#pragma unroll
    for (int e = 0; e < 4; ++e) dx[e] = rstd * (g[e] - s1 - xh[e] * s2);
    return dx;
}
// the waves' d gamma / d beta sums of one LayerNorm -> this workgroup's partial-sum row pair (rt_ln_param_grad_grouped's input)
__device__ __forceinline__ void ln_partials(float (*sm)[QE], const f32x4 dg, const f32x4 db, float* __restrict__ part) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    st4(&sm[wave][lane * 4], dg); st4(&sm[QW + wave][lane * 4], db);
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * QE; c += QT) {
        const int which = c / QE, col = c - which * QE;
        float a = 0.f;
#pragma unroll
        for (int w = 0; w < QW; ++w) a += sm[which * QW + w][col];
        part[((size_t)blockIdx.x * 2 + which) * QE + col] = a;
    }
}

// ================================================================================================ rt_qenc_fwd
__global__ __launch_bounds__(QT) void qenc_fwd_kernel(const rt_qenc_fwd_desc p) {
    int q_slot = 0; q_stamp(0, q_slot);
    __shared__ float sw[128];                 // raw scores k . q_l of the image
    __shared__ float pw[128];                 // one phrase's softmax weights
    __shared__ __attribute__((aligned(16))) bf16_t xs[QLMAX * QLD];      // the image's language rows (X operand of linear2 / linear3)
    __shared__ __attribute__((aligned(16))) float part[4][QE];
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int S = p.S, L = p.L, P = p.P;
    const bf16_t* mem16 = (const bf16_t*)p.mem16 + (size_t)b * S * QE;
    bf16_t* cls16 = (bf16_t*)p.cls16 + (size_t)b * QE;
    bf16_t* lang16 = (bf16_t*)p.lang16 + (size_t)b * L * QE;
    float* kq = p.kq + (size_t)b * QE;
    float* qs = p.qs + (size_t)b * L * QE;
    float* vs = p.vs + (size_t)b * L * QE;

    // ---- stage 1: the image's language rows -> LDS (and the bf16 copies backward's weight gradients read), then linear1 (CLS row) /
    //      linear2 / linear3 (all rows) with the X operand read from LDS
    for (int i = threadIdx.x; i < L * (QE / 8); i += QT) {
        const int r = i / (QE / 8), c = (i % (QE / 8)) * 8;
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(mem16 + (size_t)r * QE + c);
        *reinterpret_cast<bf16x8*>(xs + (size_t)r * QLD + c) = v;
        *reinterpret_cast<bf16x8*>(lang16 + (size_t)r * QE + c) = v;
        if (r == 0) *reinterpret_cast<bf16x8*>(cls16 + c) = v;
    }
    __syncthreads(); q_stamp(0, q_slot);
    for (int job = wave; job < 48; job += QW) {          // 48 tile jobs (16 per product): job = product * 16 + tile
        const int prod = job >> 4, t = job & 15;
        if (prod == 0)
            q_tile_lds<QE>((const bf16_t*)p.W1, QE, t * 16, QE, xs, QLD, 1, p.b1, [&](int, int n, const f32x4& v) { st4(kq + n, v); });
        else {
            float* out = prod == 1 ? qs : vs;
            q_tile_lds<QE>((const bf16_t*)(prod == 1 ? p.W2 : p.W3), QE, t * 16, QE, xs, QLD, L, prod == 1 ? p.b2 : p.b3,
                           [&](int m, int n, const f32x4& v) { st4(out + (size_t)m * QE + n, v); });
        }
    }
    __syncthreads(); q_stamp(0, q_slot);

    // ---- stage 2: w[j, :] = softmax_l(k . q_l masked by ctx[b, j, l]) (no 1 / sqrt(d): reftr_transformer.py:48-55), c[j] = sum_l w v_l
    for (int l = wave; l < L; l += QW) {
        float s = 0.f;
        for (int d = lane; d < QE; d += 64) s += kq[d] * qs[(size_t)l * QE + d];
        s = rt_wave_sum(s);
        if (lane == 0) sw[l] = s;
    }
    __syncthreads(); q_stamp(0, q_slot);
    for (int j = 0; j < P; ++j) {
        const int r = b * P + j;
        const uint8_t* cm = p.ctx + (size_t)r * L;
        if (wave == 0) {                                        // L <= 128: one wave holds the phrase's scores (lane, lane + 64)
            const float s0 = lane < L ? (cm[lane] ? -INFINITY : sw[lane]) : -INFINITY;
            const float s1 = lane + 64 < L ? (cm[lane + 64] ? -INFINITY : sw[lane + 64]) : -INFINITY;
            const float m = rt_wave_max(fmaxf(s0, s1));
            const float e0 = lane < L ? __expf(s0 - m) : 0.f, e1 = lane + 64 < L ? __expf(s1 - m) : 0.f;
            const float tot = rt_wave_sum(e0) + rt_wave_sum(e1);
            const float inv = 1.f / tot;
            if (lane < L) { pw[lane] = e0 * inv; p.qw[(size_t)r * L + lane] = e0 * inv; }
            if (lane + 64 < L) { pw[lane + 64] = e1 * inv; p.qw[(size_t)r * L + lane + 64] = e1 * inv; }
        }
        __syncthreads();
        {   // thread (quarter, d): a quarter of the tokens each, the quarters meet in a fixed order (as rt_qenc_attn_fwd)
            const int qt = threadIdx.x >> 8, d = threadIdx.x & 255;
            const int per = (L + 3) >> 2, l0 = qt * per, l1 = min(L, l0 + per);
            float a = 0.f;
            for (int l = l0; l < l1; ++l) a += pw[l] * vs[(size_t)l * QE + d];
            part[qt][d] = a;
        }
        __syncthreads();
        if (threadIdx.x < QE) {
            const int d = threadIdx.x;
            ((bf16_t*)p.c16)[(size_t)r * QE + d] = (bf16_t)((part[0][d] + part[1][d]) + (part[2][d] + part[3][d]));
        }
        __syncthreads();
    }

    // ---- stage 3: context_out.0 (Linear) on the image's P rows
    {
        const bf16_t* c16 = (const bf16_t*)p.c16 + (size_t)b * P * QE;
        float* co = p.co + (size_t)b * P * QE;
        const float* bc = p.bc;
        q_gemm<QE>((const bf16_t*)p.Wc, QE, QE, [&](int m) { return c16 + (size_t)m * QE; }, P,
                   [&](int m, int n, const f32x4& v) { st4(co + (size_t)m * QE + n, v + ld4(bc + n)); });
    }
    __syncthreads(); q_stamp(0, q_slot);
    // ---- stage 4: context_out.1 (LayerNorm) + the CLS row (residual) -> even half of the concatenated row (bf16)
    bf16_t* cat16 = (bf16_t*)p.cat16;
    for (int j = wave; j < P; j += QW) {
        const int r = b * P + j;
        float mean, rstd;
        const f32x4 y = ln_row(ld4(p.co + (size_t)r * QE + lane * 4), ld4(p.gc + lane * 4), ld4(p.betc + lane * 4), p.eps, false, mean, rstd);
        if (lane == 0) { p.cmean[r] = mean; p.crstd[r] = rstd; }
        st4b(cat16 + (size_t)r * 2 * QE + lane * 4, y + ld4(p.mem32 + (size_t)b * S * QE + lane * 4));
    }
    __syncthreads(); q_stamp(0, q_slot);
    // ---- stage 5: fuse_encoder_query.0 (Linear 512 -> 256)
    {
        const float* bf0 = p.bf0;
        q_gemm<2 * QE>((const bf16_t*)p.Wf0, 2 * QE, QE, [&](int m) { return cat16 + (size_t)(b * P + m) * 2 * QE; }, P,
                       [&](int m, int n, const f32x4& v) { st4(p.t1 + (size_t)(b * P + m) * QE + n, v + ld4(bf0 + n)); });
    }
    __syncthreads(); q_stamp(0, q_slot);
    // ---- stage 6: LayerNorm + ReLU + Dropout(0.1) -> a16
    {
        const bool do_drop = p.drop_p > 0.f;
        const uint32_t thresh = rt_drop_thresh(p.drop_p);
        const float ks = do_drop ? 1.f / (1.f - p.drop_p) : 1.f;
        const uint32_t seed = do_drop ? rt_site_seed(p.seed_dev, p.drop_seed) : 0u;
        for (int j = wave; j < P; j += QW) {
            const int r = b * P + j;
            float mean, rstd;
            f32x4 y = ln_row(ld4(p.t1 + (size_t)r * QE + lane * 4), ld4(p.g1 + lane * 4), ld4(p.bet1 + lane * 4), p.eps, true, mean, rstd);
            if (lane == 0) { p.m1[r] = mean; p.r1[r] = rstd; }
            if (do_drop) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    y[e] = (rt_hash32(seed, (uint32_t)(r * QE + lane * 4 + e)) >= thresh) ? y[e] * ks : 0.f;
            }
            st4b((bf16_t*)p.a16 + (size_t)r * QE + lane * 4, y);
        }
    }
    __syncthreads(); q_stamp(0, q_slot);
    // ---- stage 7: fuse_encoder_query.4 (Linear)
    {
        const bf16_t* a16 = (const bf16_t*)p.a16;
        const float* bf4 = p.bf4;
        q_gemm<QE>((const bf16_t*)p.Wf4, QE, QE, [&](int m) { return a16 + (size_t)(b * P + m) * QE; }, P,
                   [&](int m, int n, const f32x4& v) { st4(p.t2 + (size_t)(b * P + m) * QE + n, v + ld4(bf4 + n)); });
    }
    __syncthreads(); q_stamp(0, q_slot);
    // ---- stage 8: LayerNorm + ReLU -> fused phrase feature; tgt = f + query_embed[:, :E], query_pos = f + query_embed[:, E:]
    for (int j = wave; j < P; j += QW) {
        const int r = b * P + j;
        float mean, rstd;
        const f32x4 f = ln_row(ld4(p.t2 + (size_t)r * QE + lane * 4), ld4(p.g5 + lane * 4), ld4(p.bet5 + lane * 4), p.eps, true, mean, rstd);
        if (lane == 0) { p.m2[r] = mean; p.r2[r] = rstd; }
        for (int q = 0; q < p.nq; ++q) {
            const size_t o = ((size_t)r * p.nq + q) * QE + lane * 4;
            const f32x4 tg = f + ld4(p.qembed + (size_t)q * 2 * QE + lane * 4);
            const f32x4 qp = f + ld4(p.qembed + (size_t)q * 2 * QE + QE + lane * 4);
            st4(p.tgt32 + o, tg); st4(p.qpos + o, qp);
            st4b((bf16_t*)p.tgt16 + o, tg); st4b((bf16_t*)p.tgtq16 + o, tg + qp);
        }
    }
    q_stamp(0, q_slot);
}

// ================================================================================================ rt_head_loss
__global__ __launch_bounds__(QT) void head_loss_kernel(const rt_head_loss_desc p) {
    int q_slot = 0; q_stamp(1, q_slot);
    __shared__ __attribute__((aligned(16))) float sm[2 * QW][QE];
    __shared__ float red[QW][6];
    const int l = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int N = p.B * p.P * p.K;                   // rows per layer
    const size_t r0 = (size_t)l * N;
    const float* t3 = p.t3 + r0 * QE;
    bf16_t* hs16 = (bf16_t*)p.hs16 + r0 * QE;
    bf16_t* y1 = (bf16_t*)p.y1 + r0 * QE;
    bf16_t* y2 = (bf16_t*)p.y2 + r0 * QE;
    bf16_t* dy2 = (bf16_t*)p.dy2 + r0 * QE;
    bf16_t* dy1 = (bf16_t*)p.dy1 + r0 * QE;
    float* dhs = p.dhs + r0 * QE;

    // ---- decoder.norm on the layer's rows (transformer.py:131-141)
    for (int m = wave; m < N; m += QW) {
        float mean, rstd;
        const f32x4 y = ln_row(ld4(t3 + (size_t)m * QE + lane * 4), ld4(p.gn + lane * 4), ld4(p.betn + lane * 4), p.eps, false, mean, rstd);
        if (lane == 0) { p.hmean[r0 + m] = mean; p.hrstd[r0 + m] = rstd; }
        st4b(hs16 + (size_t)m * QE + lane * 4, y);
    }
    __syncthreads(); q_stamp(1, q_slot);
    // ---- bbox MLP (backbone.py:26-38): two Linear + ReLU (bf16 out), one Linear 256 -> 4 (fp32 logits)
    {
        const float* b0 = p.b0;
        q_gemm<QE>((const bf16_t*)p.W0, QE, QE, [&](int m) { return hs16 + (size_t)m * QE; }, N, [&](int m, int n, const f32x4& v) {
            f32x4 y = v + ld4(b0 + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = fmaxf(y[e], 0.f);
            st4b(y1 + (size_t)m * QE + n, y);
        });
    }
    __syncthreads(); q_stamp(1, q_slot);
    {
        const float* b1 = p.b1;
        q_gemm<QE>((const bf16_t*)p.W1, QE, QE, [&](int m) { return y1 + (size_t)m * QE; }, N, [&](int m, int n, const f32x4& v) {
            f32x4 y = v + ld4(b1 + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = fmaxf(y[e], 0.f);
            st4b(y2 + (size_t)m * QE + n, y);
        });
    }
    __syncthreads(); q_stamp(1, q_slot);
    // ---- logits + box loss + d total / d logits: the lane that holds a row's four logits evaluates the row
    float l1_sum = 0.f, gi_sum = 0.f, dbias[4] = {0.f, 0.f, 0.f, 0.f};
    {
        const float nb = fmaxf(p.num_boxes[0], 1.f) * (float)p.K;
        const float wb = p.weights[l * 2], wg = p.weights[l * 2 + 1];
        const f32x4 b2 = ld4(p.b2);
        // one 16-feature tile (4 real features): rows in groups of 48, the groups dealt to the waves
        for (int m0 = wave * 48; m0 < N; m0 += QW * 48) {
            const int rows = min(48, N - m0);
            auto xr = [&](int m) { return y2 + (size_t)(m0 + m) * QE; };
            auto ep = [&](int mm, int n, const f32x4& v) {
                if (n != 0) return;
                const int m = m0 + mm;
                const f32x4 lg4 = v + b2;
                st4(p.logits + (r0 + m) * 4, lg4);
                const int k = m % p.K, ph = (m / p.K) % p.P, bi = m / (p.K * p.P);
                f32x4 g4 = f32x4{0.f, 0.f, 0.f, 0.f};
                if ((p.valid[(size_t)bi * p.P * p.K + ph * p.K + k] != 0) != (p.invert_valid != 0)) {
                    int rank = 0, nvalid = 0;              // masked_select keeps phrase order (criterion.py:126)
                    for (int q = 0; q < p.P; ++q) {
                        const int vv = ((p.valid[(size_t)bi * p.P * p.K + q * p.K] != 0) != (p.invert_valid != 0)) ? 1 : 0;
                        nvalid += vv;
                        if (q < ph) rank += vv;
                    }
                    const int count = p.tgt_off[bi + 1] - p.tgt_off[bi];
                    const bool mismatch = nvalid != count;   // criterion.py:127 asserts; a kernel poisons the loss instead (rt_box_loss)
                    const float lg[4] = {lg4[0], lg4[1], lg4[2], lg4[3]};
                    const float* tg = mismatch ? lg : p.targets + ((size_t)p.tgt_off[bi] + rank) * 4;
                    if (mismatch) l1_sum = __builtin_nanf("");
                    float g[4];
                    rt_box_loss_row(lg, tg, wb, wg, nb, l1_sum, gi_sum, g);
                    g4 = f32x4{g[0], g[1], g[2], g[3]};
                }
                st4(p.dlogits + (r0 + m) * 4, g4);
                st4b((bf16_t*)p.dl16 + (r0 + m) * 4, g4);
#pragma unroll
                for (int e = 0; e < 4; ++e) dbias[e] += g4[e];
            };
            if (rows <= 16)      q_tile<1, QE>((const bf16_t*)p.W2, QE, 0, 4, xr, rows, ep);
            else if (rows <= 32) q_tile<2, QE>((const bf16_t*)p.W2, QE, 0, 4, xr, rows, ep);
            else                 q_tile<3, QE>((const bf16_t*)p.W2, QE, 0, 4, xr, rows, ep);
        }
        // the layer's two loss terms: lanes -> wave -> workgroup in a fixed order (no atomics, no clear: this workgroup owns them)
        l1_sum = rt_wave_sum(l1_sum); gi_sum = rt_wave_sum(gi_sum);
#pragma unroll
        for (int e = 0; e < 4; ++e) dbias[e] = rt_wave_sum(dbias[e]);
        if (lane == 0) {
            red[wave][0] = l1_sum; red[wave][1] = gi_sum;
#pragma unroll
            for (int e = 0; e < 4; ++e) red[wave][2 + e] = dbias[e];
        }
        __syncthreads();
        if (threadIdx.x < 6) {
            float a = 0.f;
#pragma unroll
            for (int w = 0; w < QW; ++w) a += red[w][threadIdx.x];
            if (threadIdx.x < 2) {
                // write-through (agent-scope) store: the workgroup that draws the last ticket below adds the layers up
                __hip_atomic_store(p.losses + l * 2 + threadIdx.x, a / nb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else if (p.db2_part) {                      // this layer's share of the last Linear's bias gradient: partial-sum row pair
                p.db2_part[l * 8 + (threadIdx.x - 2)] = a; p.db2_part[l * 8 + 4 + (threadIdx.x - 2)] = 0.f;
            }
        }
    }
    if (p.total && p.ticket && threadIdx.x == 0) {
        // the weighted total (engine_vg.py:43) by the LAST workgroup to get here, in layer order (reproducible): the loss stores above
        // are write-through, the reads bypass this compute unit's L1
        // ordering at the language level: the ticket is an acquire-release agent-scope RMW -- this workgroup's loss
        // stores happen-before the increment, and the workgroup that draws the last ticket sees every earlier one's stores
        const int old = __hip_atomic_fetch_add(p.ticket, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (old == p.NL - 1) {
            float tot = 0.f;
            for (int i = 0; i < p.NL; ++i) {
                const float lb = __hip_atomic_load(p.losses + i * 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const float lg2 = __hip_atomic_load(p.losses + i * 2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                tot += p.weights[i * 2] * lb + p.weights[i * 2 + 1] * lg2;
            }
            p.total[0] = tot;
            __hip_atomic_store(p.ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch
        }
    }
    __syncthreads(); q_stamp(1, q_slot);
    // ---- backward-data of the last Linear in fp32 (rt_small_dgrad: 4 products per element) with y2's ReLU mask -> bf16
    {
        const float* dl = p.dlogits + r0 * 4;
        for (int i = threadIdx.x; i < N * (QE / 4); i += QT) {
            const int m = i / (QE / 4), k = (i % (QE / 4)) * 4;
            const f32x4 d = ld4(dl + (size_t)m * 4);
            const bf16x4 gt = *reinterpret_cast<const bf16x4*>(y2 + (size_t)m * QE + k);
            f32x4 a;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float s = 0.f;
#pragma unroll
                for (int n = 0; n < 4; ++n) s += d[n] * p.w2_f32[(size_t)n * QE + k + e];
                a[e] = ((float)gt[e] > 0.f) ? s : 0.f;
            }
            st4b(dy2 + (size_t)m * QE + k, a);
        }
    }
    __syncthreads(); q_stamp(1, q_slot);
    {   // d y1 = (d y2 @ W1) * (y1 > 0)
        q_gemm<QE>((const bf16_t*)p.W1T, QE, QE, [&](int m) { return dy2 + (size_t)m * QE; }, N, [&](int m, int n, const f32x4& v) {
            const bf16x4 gt = *reinterpret_cast<const bf16x4*>(y1 + (size_t)m * QE + n);
            f32x4 y;
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = ((float)gt[e] > 0.f) ? v[e] : 0.f;
            st4b(dy1 + (size_t)m * QE + n, y);
        });
    }
    __syncthreads(); q_stamp(1, q_slot);
    q_gemm<QE>((const bf16_t*)p.W0T, QE, QE, [&](int m) { return dy1 + (size_t)m * QE; }, N,
               [&](int m, int n, const f32x4& v) { st4(dhs + (size_t)m * QE + n, v); });
    __syncthreads(); q_stamp(1, q_slot);
    // ---- decoder.norm backward: d t3 per row, d gamma / d beta as this workgroup's partial sums
    f32x4 dg = f32x4{0.f, 0.f, 0.f, 0.f}, db = dg;
    {
        const f32x4 gam = ld4(p.gn + lane * 4), bet = ld4(p.betn + lane * 4);
        for (int m = wave; m < N; m += QW) {
            const f32x4 dx = ln_row_bwd(ld4(dhs + (size_t)m * QE + lane * 4), ld4(t3 + (size_t)m * QE + lane * 4), p.hmean[r0 + m],
                                        p.hrstd[r0 + m], gam, bet, false, false, 0u, 0u, 1.f, 0u, dg, db);
            st4(p.dnorm + (r0 + m) * QE + lane * 4, dx);
        }
    }
    ln_partials(sm, dg, db, p.part_n);
    q_stamp(1, q_slot);
}

// ================================================================================================ rt_qenc_bwd
__global__ __launch_bounds__(QT) void qenc_bwd_kernel(const rt_qenc_bwd_desc p) {
    int q_slot = 0; q_stamp(2, q_slot);
    __shared__ __attribute__((aligned(16))) float sm[2 * QW][QE];
    __shared__ float sds[16][128];            // d s[j][l] of the image's phrases
    __shared__ __attribute__((aligned(16))) bf16_t xdq[QLMAX * QLD], xdv[QLMAX * QLD], xk[QLD];     // LDS images of d q, d v, d k
    __shared__ __attribute__((aligned(16))) float srow[QE];
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int S = p.S, L = p.L, P = p.P;
    const int rb = b * P;                     // first phrase row of the image
    const bool do_drop = p.drop_p > 0.f;
    const uint32_t thresh = rt_drop_thresh(p.drop_p);
    const float ks = do_drop ? 1.f / (1.f - p.drop_p) : 1.f;
    const uint32_t seed = do_drop ? rt_site_seed(p.seed_dev, p.drop_seed) : 0u;

    // ---- d fused = ga (+ gb) + d query_pos; d query_embed; fuse_encoder_query.5 (LayerNorm + ReLU) backward -> dt2b
    {
        f32x4 dg = f32x4{0.f, 0.f, 0.f, 0.f}, db = dg, sa = dg, sq = dg;
        const f32x4 gam = ld4(p.g5 + lane * 4), bet = ld4(p.bet5 + lane * 4);
        for (int j = wave; j < P; j += QW) {
            const int r = rb + j;
            const f32x4 a = ld4(p.ga + (size_t)r * QE + lane * 4), q = ld4(p.dqpos + (size_t)r * QE + lane * 4);
            f32x4 d = a;
            if (p.gb) { const f32x4 g2 = ld4(p.gb + (size_t)r * QE + lane * 4); d = d + g2; sa = sa + g2; }
            d = d + q;
            sa = sa + a; sq = sq + q;
            const f32x4 dx = ln_row_bwd(d, ld4(p.t2 + (size_t)r * QE + lane * 4), p.m2[r], p.r2[r], gam, bet, true, false, 0u, 0u, 1.f, 0u, dg, db);
            st4b((bf16_t*)p.dt2b + (size_t)r * QE + lane * 4, dx);
        }
        if (p.dqembed) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (sa[e] != 0.f) atomicAdd(p.dqembed + lane * 4 + e, sa[e]);
                if (sq[e] != 0.f) atomicAdd(p.dqembed + QE + lane * 4 + e, sq[e]);
            }
        }
        ln_partials(sm, dg, db, p.part5);
    }
    __syncthreads(); q_stamp(2, q_slot);
    // ---- fuse_encoder_query.4 backward-data -> da (fp32)
    q_gemm<QE>((const bf16_t*)p.Wf4T, QE, QE, [&](int m) { return (const bf16_t*)p.dt2b + (size_t)(rb + m) * QE; }, P,
               [&](int m, int n, const f32x4& v) { st4(p.da + (size_t)(rb + m) * QE + n, v); });
    __syncthreads(); q_stamp(2, q_slot);
    // ---- fuse_encoder_query.1 (LayerNorm + ReLU + Dropout) backward -> dt1b
    {
        f32x4 dg = f32x4{0.f, 0.f, 0.f, 0.f}, db = dg;
        const f32x4 gam = ld4(p.g1 + lane * 4), bet = ld4(p.bet1 + lane * 4);
        for (int j = wave; j < P; j += QW) {
            const int r = rb + j;
            const f32x4 dx = ln_row_bwd(ld4(p.da + (size_t)r * QE + lane * 4), ld4(p.t1 + (size_t)r * QE + lane * 4), p.m1[r], p.r1[r], gam, bet,
                                        true, do_drop, seed, thresh, ks, (uint32_t)(r * QE + lane * 4), dg, db);
            st4b((bf16_t*)p.dt1b + (size_t)r * QE + lane * 4, dx);
        }
        ln_partials(sm, dg, db, p.part1);
    }
    __syncthreads(); q_stamp(2, q_slot);
    // ---- fuse_encoder_query.0 backward-data -> d cat (fp32 [N, 2E]: even half = context branch, odd half = map_phrase's output)
    q_gemm<QE>((const bf16_t*)p.Wf0T, QE, 2 * QE, [&](int m) { return (const bf16_t*)p.dt1b + (size_t)(rb + m) * QE; }, P,
               [&](int m, int n, const f32x4& v) { st4(p.dcat + (size_t)(rb + m) * 2 * QE + n, v); });
    __syncthreads(); q_stamp(2, q_slot);
    // ---- context_out.1 (LayerNorm) backward -> dcob; the residual's gradient (sum over the image's phrases) -> d memory[CLS row]
    {
        f32x4 dg = f32x4{0.f, 0.f, 0.f, 0.f}, db = dg;
        const f32x4 gam = ld4(p.gc + lane * 4), bet = ld4(p.betc + lane * 4);
        for (int j = wave; j < P; j += QW) {
            const int r = rb + j;
            const f32x4 dx = ln_row_bwd(ld4(p.dcat + (size_t)r * 2 * QE + lane * 4), ld4(p.co + (size_t)r * QE + lane * 4), p.cmean[r], p.crstd[r],
                                        gam, bet, false, false, 0u, 0u, 1.f, 0u, dg, db);
            st4b((bf16_t*)p.dcob + (size_t)r * QE + lane * 4, dx);
        }
        ln_partials(sm, dg, db, p.partc);
        if (threadIdx.x < QE) {
            float a = 0.f;
            for (int j = 0; j < P; ++j) a += p.dcat[(size_t)(rb + j) * 2 * QE + threadIdx.x];
            srow[threadIdx.x] = a;                     // added to d memory's CLS row at the end (one read-modify-write per element)
        }
    }
    __syncthreads(); q_stamp(2, q_slot);
    // ---- context_out.0 backward-data -> dc (fp32)
    q_gemm<QE>((const bf16_t*)p.WcT, QE, QE, [&](int m) { return (const bf16_t*)p.dcob + (size_t)(rb + m) * QE; }, P,
               [&](int m, int n, const f32x4& v) { st4(p.dc + (size_t)(rb + m) * QE + n, v); });
    __syncthreads(); q_stamp(2, q_slot);
    // ---- attention backward (rt_qenc_attn_bwd): dw[j][l] = dc[j] . v_l; ds = w (dw - sum_l w dw); dv_l = sum_j w[j][l] dc[j];
    //      dq_l = sum_j ds[j][l] k;  dk = sum_j sum_l ds[j][l] q_l      (sums over the phrases in phrase order: reproducible)
    {
        const float* qs = p.qs + (size_t)b * L * QE;
        const float* vs = p.vs + (size_t)b * L * QE;
        const float* kq = p.kq + (size_t)b * QE;
        for (int i = wave; i < P * L; i += QW) {
            const int j = i / L, l = i - j * L;
            float s = 0.f;
            for (int d = lane; d < QE; d += 64) s += p.dc[(size_t)(rb + j) * QE + d] * vs[(size_t)l * QE + d];
            s = rt_wave_sum(s);
            if (lane == 0) sds[j][l] = s;
        }
        __syncthreads();
        for (int j = wave; j < P; j += QW) {
            const float* wr = p.qw + (size_t)(rb + j) * L;
            const float w0 = lane < L ? wr[lane] : 0.f, w1 = lane + 64 < L ? wr[lane + 64] : 0.f;
            const float d0 = lane < L ? sds[j][lane] : 0.f, d1 = lane + 64 < L ? sds[j][lane + 64] : 0.f;
            const float dot = rt_wave_sum(w0 * d0) + rt_wave_sum(w1 * d1);
            if (lane < L) sds[j][lane] = w0 * (d0 - dot);
            if (lane + 64 < L) sds[j][lane + 64] = w1 * (d1 - dot);
        }
        __syncthreads();
        // thread (group, d): tokens l = group, group + 4, ...; bf16 images of d q_l, d v_l (the operands of linear2 / linear3's backward)
        const int d = threadIdx.x & 255, hf = threadIdx.x >> 8;
        const float kd = kq[d];
        float gk = 0.f;
        for (int l = hf; l < L; l += QH) {
            float dq = 0.f, dv = 0.f;
            for (int j = 0; j < P; ++j) {
                const float ds = sds[j][l];
                dq += ds * kd;
                dv += p.qw[(size_t)(rb + j) * L + l] * p.dc[(size_t)(rb + j) * QE + d];
                gk += ds * qs[(size_t)l * QE + d];
            }
            ((bf16_t*)p.dqs16)[((size_t)b * L + l) * QE + d] = (bf16_t)dq; xdq[(size_t)l * QLD + d] = (bf16_t)dq;
            ((bf16_t*)p.dvs16)[((size_t)b * L + l) * QE + d] = (bf16_t)dv; xdv[(size_t)l * QLD + d] = (bf16_t)dv;
        }
        sm[hf][d] = gk;
        __syncthreads();
        if (threadIdx.x < QE) {
            const bf16_t v = (bf16_t)((sm[0][d] + sm[1][d]) + (sm[2][d] + sm[3][d]));
            ((bf16_t*)p.dk16)[(size_t)b * QE + d] = v; xk[d] = v;
        }
    }
    __syncthreads(); q_stamp(2, q_slot);
    // ---- linear1 / linear2 / linear3 backward-data, accumulated into d memory: language rows += d q W2 + d v W3, CLS row += d k W1
    //      (+ the context residual's gradient).  This launch is the only writer of these rows while it runs.  The X operands (d q,
    //      d v, d k: just written by this workgroup) are read from their LDS images.
    {
        float* dmem = p.dmem + (size_t)b * S * QE;
        for (int t = wave; t < QE / 16; t += QW) {
            const int lanei = lane & 15;
            // d cls (one row) first: it is added to row 0 together with the residual's gradient
            f32x4 dcls = f32x4{0.f, 0.f, 0.f, 0.f};
            q_tile_lds<QE>((const bf16_t*)p.W1T, QE, t * 16, QE, xk, QLD, 1, nullptr, [&](int, int, const f32x4& v) { dcls = v; });
            // two products per tile and 16-row group: acc(d v W3) + acc(d q W2), in rt_conv_gemm's order (product, then the fp32 residual)
            const int lg = lane >> 4;
            const bf16_t* w2p = (const bf16_t*)p.W2T + (size_t)(t * 16 + lanei) * QE + lg * 8;
            const bf16_t* w3p = (const bf16_t*)p.W3T + (size_t)(t * 16 + lanei) * QE + lg * 8;
            bf16x8 a2[QE / 32], a3[QE / 32];
#pragma unroll
            for (int i = 0; i < QE / 32; ++i) { a2[i] = *reinterpret_cast<const bf16x8*>(w2p + i * 32); a3[i] = *reinterpret_cast<const bf16x8*>(w3p + i * 32); }
            for (int m0 = 0; m0 < L; m0 += 16) {
                int m = m0 + lanei; if (m >= L) m = L - 1;
                const bf16_t* xq = xdq + (size_t)m * QLD + lg * 8;
                const bf16_t* xv = xdv + (size_t)m * QLD + lg * 8;
                f32x4 aq = f32x4{0.f, 0.f, 0.f, 0.f}, av = aq;
#pragma unroll
                for (int i = 0; i < QE / 32; ++i) {
                    aq = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2[i], *reinterpret_cast<const bf16x8*>(xq + i * 32), aq, 0, 0, 0);
                    av = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3[i], *reinterpret_cast<const bf16x8*>(xv + i * 32), av, 0, 0, 0);
                }
                if (m0 + lanei < L) {
                    float* o = dmem + (size_t)(m0 + lanei) * QE + t * 16 + lg * 4;
                    f32x4 cur = ld4(o);
                    if (m0 + lanei == 0) {
                        cur = cur + ld4(srow + t * 16 + lg * 4);
                        cur = cur + dcls;
                    }
                    st4(o, cur + (av + aq));
                }
            }
        }
    }
    q_stamp(2, q_slot);
}

}  // namespace

extern "C" int rt_qenc_fwd(const rt_qenc_fwd_desc* d, rt_stream_t stream) {
    if (!d || !d->mem16 || !d->mem32 || !d->ctx || !d->cat16 || !d->tgt32) return RT_ERR_BADARG;
    if (d->E != QE || d->L <= 0 || d->L > QLMAX || d->P <= 0 || d->P > 16 || d->nq <= 0 || d->B <= 0) return RT_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(qenc_fwd_kernel, dim3(d->B), dim3(QT), 0, (hipStream_t)stream, *d);
    RT_CHECK_LAUNCH();
    return RT_OK;
}

extern "C" int rt_head_loss(const rt_head_loss_desc* d, rt_stream_t stream) {
    if (!d || !d->t3 || !d->logits || !d->losses || !d->weights || !d->valid || !d->targets || !d->tgt_off || !d->num_boxes ||
        !d->dnorm || !d->part_n) return RT_ERR_BADARG;
    if (d->E != QE || d->NL <= 0 || d->B <= 0 || d->P <= 0 || d->K <= 0) return RT_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(head_loss_kernel, dim3(d->NL), dim3(QT), 0, (hipStream_t)stream, *d);
    RT_CHECK_LAUNCH();
    return RT_OK;
}

extern "C" int rt_qenc_bwd(const rt_qenc_bwd_desc* d, rt_stream_t stream) {
    if (!d || !d->ga || !d->dqpos || !d->dmem || !d->part5 || !d->part1 || !d->partc) return RT_ERR_BADARG;
    if (d->E != QE || d->L <= 0 || d->L > QLMAX || d->P <= 0 || d->P > 16 || d->B <= 0) return RT_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(qenc_bwd_kernel, dim3(d->B), dim3(QT), 0, (hipStream_t)stream, *d);
    RT_CHECK_LAUNCH();
    return RT_OK;
}

extern "C" int rt_qregion_trace(unsigned long long* host_out) {
    if (!host_out) return RT_ERR_BADARG;
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_qtrace), sizeof(unsigned long long) * 3 * 24);
}
