// The row-local part of a TransformerEncoderLayer (models/modeling/transformer.py:168-181, forward_post) as ONE launch per
// direction, for the reference's width E = 256:
//
//   rt_enc_tail_fwd   o (attention output) -> out_proj + dropout1 + residual -> norm1 -> linear1 + ReLU + dropout -> linear2 +
//                     dropout2 + residual -> norm2 (+ pos) -> [the NEXT layer's q/k and v projections]
//   rt_enc_tail_bwd   norm2 backward -> linear2^T (ReLU / dropout gate) -> linear1^T + residual -> norm1 backward -> out_proj^T
//
// Everything between two attention launches of the encoder is row-local: a workgroup owns 32 rows of the [B*S, 256] sequence and
// walks the whole chain on them, the intermediate row blocks living in LDS as bf16 MFMA operands.  As launches this was 6 + 5 kernels
// per layer (53 + 68 launches for the six layers, 0.58 + 0.62 ms of the step with nothing else on the GPU -- profiles/
// r04a_concurrent_timeline.txt: every one of them a ~2 us dependent boundary plus a cold ramp for 5-20 us of work, and ~100 LayerNorm
// launches of 5-7 us that do 2 us of work).  The feed-forward pair is walked in chunks of 256 hidden units -- linear1's chunk is
// consumed by linear2's partial product at once -- so the [32, 2048] hidden block is never an operand larger than 16 KB.
//
// Every product is "32 rows x K=256 times a 256 x 256 weight block" (a UNIT): the weight block streams global -> LDS by DMA in four
// 64-wide K tiles of 32 KB (buffer_load_dwordx4 ... lds, swizzle on the source side as in rt_gemm_dma.h), three stages, one barrier
// per tile; the tile sequence of the whole chain (80 tiles forward, 68 backward) is ONE pipeline -- the loader runs ahead across
// unit boundaries and under the LayerNorm / epilogue code, because it depends on nothing the kernel computes.  512 threads: wave w
// owns the 16 rows (w >> 2) x 64 columns (w & 3) of a unit's output (four 16x16x32 MFMA accumulators), so a LayerNorm row statistic
// is a 4-lane-group shuffle plus a 4-wave LDS exchange, with the values still in the accumulator layout.
// Arithmetic, rounding points and dropout sites are those of the launched chain (rt_conv_gemm epilogues + rt_layernorm_*): the
// saved tensors are interchangeable with the chain's, the weight-gradient launches stay outside and read what this writes.
#include "rt_common.h"
#include <stdlib.h>

namespace {

constexpr int EE = 256, RB = 32, ENT = 512, ENS = 3;
constexpr int STAGE_BYTES = 256 * 128, ACT_BYTES = RB * EE * 2, ELPT = 4;      // 256 weight rows x 64 k; DMA instructions per thread per tile
constexpr int TOUCH_OFF = ENS * STAGE_BYTES + 3 * ACT_BYTES + 2 * RB * 4 * 4 + 2 * 2 * EE * 4;      // 256 scratch bytes per wave for the L2 touches
constexpr int SMEM_BYTES = TOUCH_OFF + 8 * 256;

typedef __attribute__((ext_vector_type(4))) int e_i32x4;
typedef bf16_t bf16x4_t __attribute__((ext_vector_type(4)));
template <int N> __device__ __forceinline__ void e_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ e_i32x4 e_rsrc(const void* ptr, unsigned bytes) {
    const uint64_t a = (uint64_t)ptr;
    return e_i32x4{__builtin_amdgcn_readfirstlane((int)(uint32_t)a), __builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32)),
                   __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000};
}
__device__ __forceinline__ void e_dma16(const e_i32x4 rsrc, unsigned lds_base, int voff, int soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 ::"s"(lds_base), "v"(voff), "s"(rsrc), "s"(soff)
                 : "memory", "m0");
}

// 4 bytes per lane into a scratch LDS word: pulls a 128-byte line into this XCD's L2 without a register destination
__device__ __forceinline__ void e_dma4(const e_i32x4 rsrc, unsigned lds_base, int voff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds"
                 ::"s"(lds_base), "v"(voff), "s"(rsrc)
                 : "memory", "m0");
}
// Outputs are written through and dropped from L2 (sc1): a layer's kernel writes ~4.5 MB per XCD that nothing in it reads again,
// next to the 2.6 MB weight stream that all of the XCD's workgroups re-read tile by tile from the 4 MB L2.
constexpr int E_WT = 16;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t e_out_rsrc(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}
typedef unsigned e_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned e_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void e_st_f32x4(float* base, size_t elem, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<e_u32x4*>(&v), e_out_rsrc(base), (int)(elem * 4), 0, E_WT);
}
__device__ __forceinline__ void e_st_bf16x4(void* base, size_t elem, bf16x4_t v) {
    __builtin_amdgcn_raw_buffer_store_b64(*reinterpret_cast<e_u32x2*>(&v), e_out_rsrc(base), (int)(elem * 2), 0, E_WT);
}

struct EUnit { const bf16_t* W; unsigned bytes; int n_base, ldw, k_base; };

// Shared machinery of both kernels: thread coordinates, the weight-tile pipeline, the unit product, row statistics.
struct ECtx {
    unsigned char* smem;
    int t, lane, wave, wn, wm, li, lg, srow, chunk;
    unsigned lds_stage0;
    int next_issue, next_use, n_tiles;
};

__device__ __forceinline__ void ectx_init(ECtx& c, unsigned char* smem, int n_tiles) {
    typedef __attribute__((address_space(3))) void* lds_ptr;
    c.smem = smem;
    c.t = threadIdx.x; c.lane = c.t & 63; c.wave = __builtin_amdgcn_readfirstlane(c.t >> 6);
    c.wn = c.wave & 3; c.wm = c.wave >> 2; c.li = c.lane & 15; c.lg = c.lane >> 4;
    c.srow = c.t >> 3; c.chunk = (c.t & 7) ^ (c.srow & 7);
    c.lds_stage0 = (unsigned)(uintptr_t)(lds_ptr)smem + (unsigned)c.wave * 1024u;
    c.next_issue = 0; c.next_use = 0; c.n_tiles = n_tiles;
}
__device__ __forceinline__ unsigned char* e_act(const ECtx& c, int i) { return c.smem + ENS * STAGE_BYTES + i * ACT_BYTES; }
__device__ __forceinline__ float* e_red(const ECtx& c, int i) { return reinterpret_cast<float*>(c.smem + ENS * STAGE_BYTES + 3 * ACT_BYTES) + i * RB * 4; }
__device__ __forceinline__ float* e_colbuf(const ECtx& c) { return reinterpret_cast<float*>(c.smem + ENS * STAGE_BYTES + 3 * ACT_BYTES + 2 * RB * 4 * 4); }

// The weight stream is COLD when a layer's kernel starts (every layer has its own weights; AdamW rewrote them milliseconds ago): with
// three 32 KB stages in flight a tile that misses L2 costs ~1 us (measured: 80 tiles -> 81 us, profiles/r04e_*), 14 workgroups per XCD
// waiting on the same miss.  So the first thing the kernel does is ask for the WHOLE stream: the workgroups of an XCD (blocks b, b + 8,
// ...: the hardware deals them round-robin -- for speed only) split its 128-byte lines among them, 4-byte LDS-DMA touches into a scratch
// word, ~3 per thread, never read.  They are the oldest entries of every wave's vmcnt queue, i.e. the first tile's wait covers them.
__device__ __forceinline__ void e_touch(const ECtx& c, const void* base, unsigned bytes) {
    typedef __attribute__((address_space(3))) void* lds_ptr;
    const int nx = ((int)gridDim.x + 7) >> 3, xw = (int)blockIdx.x >> 3;
    const bool short_xcd = ((int)gridDim.x & 7) != 0 && ((int)blockIdx.x & 7) >= ((int)gridDim.x & 7);     // this XCD has nx - 1 workgroups
    const e_i32x4 rs = e_rsrc(base, bytes);
    const unsigned scratch = (unsigned)(uintptr_t)(lds_ptr)(c.smem + TOUCH_OFF) + (unsigned)c.wave * 256u;
    const int lines = (int)((bytes + 127) >> 7);
    for (int l0 = 0; l0 < lines; l0 += nx * ENT) {
        const int l = l0 + xw * ENT + c.t;
        e_dma4(rs, scratch, l < lines ? l * 128 : 0x7fffffff);
        if (short_xcd && xw == 0) { const int l2 = l0 + (nx - 1) * ENT + c.t; e_dma4(rs, scratch, l2 < lines ? l2 * 128 : 0x7fffffff); }
    }
}

// tile `ti` of the kernel's weight stream -> its stage (ti % ENS).  Past the end: out-of-range offsets (the DMA writes zeros into a
// stage nobody reads any more), so that every thread's vmcnt bookkeeping stays uniform.
template <typename UnitOf>
__device__ __forceinline__ void e_issue(ECtx& c, const UnitOf& unit_of) {
    const int ti = c.next_issue++;
    constexpr int OOB = 0x7fffffff;
    const bool live = ti < c.n_tiles;
    const EUnit u = unit_of(live ? (ti >> 2) : 0);
    const e_i32x4 rs = e_rsrc(u.W, u.bytes);
    const int soff = (u.k_base + (ti & 3) * 64) * 2;
    const unsigned base = c.lds_stage0 + (unsigned)(ti % ENS) * STAGE_BYTES;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = c.srow + 64 * j;
        e_dma16(rs, base + j * 8192, live ? ((u.n_base + row) * u.ldw + c.chunk * 8) * 2 : OOB, soff);
    }
}

// acc[a] (+)= act[32 x 256] (bf16, LDS, four swizzled [32][64] sub-tiles) x the next four weight tiles of the stream
template <typename UnitOf>
__device__ __forceinline__ void e_unit(ECtx& c, const UnitOf& unit_of, const unsigned char* act, f32x4 (&acc)[4]) {
#pragma unroll 1
    for (int kt = 0; kt < 4; ++kt) {
        e_wait_vmcnt<(ENS - 2) * ELPT>();          // tile next_use has landed (this thread's pieces; later loads / stores only make this stricter)
        __syncthreads();                           // ... everyone's; and nobody reads the stage of tile next_use - 1 any more
        e_issue(c, unit_of);                       // tile next_use + ENS - 1 goes into that stage
        const unsigned char* bA = c.smem + (c.next_use % ENS) * STAGE_BYTES;
        const unsigned char* bB = act + kt * (RB * 128);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int slot = ((kk * 4 + c.lg) ^ (c.li & 7)) << 4;
            const bf16x8 bfr = *reinterpret_cast<const bf16x8*>(bB + (c.wm * 16 + c.li) * 128 + slot);
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const bf16x8 af = *reinterpret_cast<const bf16x8*>(bA + (c.wn * 64 + a * 16 + c.li) * 128 + slot);
                acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfr, acc[a], 0, 0, 0);
            }
        }
        ++c.next_use;
    }
}

// this thread's 4 x 4 values (row m = wm * 16 + li, columns n = wn * 64 + a * 16 + lg * 4 + r) -> bf16 operand rows of act buffer `dst`
__device__ __forceinline__ void e_store_act(const ECtx& c, unsigned char* dst, const bf16x4_t (&v)[4]) {
    const int m = c.wm * 16 + c.li;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int k8 = a * 2 + (c.lg >> 1);         // 8-element slot inside the 64-wide sub-tile wn
        *reinterpret_cast<bf16x4_t*>(dst + c.wn * (RB * 128) + m * 128 + ((k8 ^ (m & 7)) << 4) + (c.lg & 1) * 8) = v[a];
    }
}
__device__ __forceinline__ int e_col(const ECtx& c, int a) { return c.wn * 64 + a * 16 + c.lg * 4; }

// sum over the 256 columns of row m of a per-thread partial (4 lane groups x 4 waves); `buf`: 32 x 4 floats, one barrier
__device__ __forceinline__ float e_row_sum(const ECtx& c, float s, float* buf) {
    s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
    const int m = c.wm * 16 + c.li;
    if (c.lg == 0) buf[m * 4 + c.wn] = s;
    __syncthreads();
    const f32x4 q = *reinterpret_cast<const f32x4*>(buf + m * 4);
    return (q[0] + q[1]) + (q[2] + q[3]);
}

__device__ __forceinline__ f32x4 e_ld4(const float* p, bool ok) { return ok ? *reinterpret_cast<const f32x4*>(p) : f32x4{0.f, 0.f, 0.f, 0.f}; }

// rows of a bf16 [M, 256] tensor -> act buffer (plain loads + swizzled LDS stores; rows past M are zeros)
__device__ __forceinline__ void e_load_rows(const ECtx& c, const bf16_t* src, int m0, int M, unsigned char* dst) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int p = c.t + ENT * j;                   // 1024 16-byte pieces: row = p >> 5, 8-column slot = p & 31
        const int row = p >> 5, s8 = p & 31;
        bf16x8 v = bf16x8{};
        if (m0 + row < M) v = *reinterpret_cast<const bf16x8*>(src + (size_t)(m0 + row) * EE + s8 * 8);
        *reinterpret_cast<bf16x8*>(dst + (s8 >> 3) * (RB * 128) + row * 128 + (((s8 & 7) ^ (row & 7)) << 4)) = v;
    }
}

// ------------------------------------------------------------------------------------------------------------------ forward
__global__ __launch_bounds__(ENT, 1) void enc_tail_fwd_kernel(const rt_enc_tail_fwd_desc p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int C = p.F >> 8;                             // chunks of 256 hidden units
    const bool proj = p.qk != nullptr && p.mode == 0;
    ECtx c;
    ectx_init(c, smem, p.mode == 1 ? 4 : 4 * (1 + 2 * C + (proj ? 3 : 0)));
    auto unit_of = [&](int u) __attribute__((always_inline)) -> EUnit {
        if (u == 0) return EUnit{(const bf16_t*)p.Wo, (unsigned)(EE * EE * 2), 0, EE, 0};
        if (u <= 2 * C) {
            const int cc = (u - 1) >> 1;
            if ((u - 1) & 1) return EUnit{(const bf16_t*)p.W2, (unsigned)(EE * p.F * 2), 0, p.F, cc * 256};
            return EUnit{(const bf16_t*)p.W1, (unsigned)(p.F * EE * 2), cc * 256, EE, 0};
        }
        const int v = u - 2 * C - 1;
        if (v < 2) return EUnit{(const bf16_t*)p.Wqk, (unsigned)(2 * EE * EE * 2), v * 256, EE, 0};
        return EUnit{(const bf16_t*)p.Wv, (unsigned)(EE * EE * 2), 0, EE, 0};
    };
    const int m0 = blockIdx.x * RB;
    const int m = m0 + c.wm * 16 + c.li;
    const bool row_ok = m < p.M;
    // the loads that depend on nothing computed here go first (oldest in the memory queue: they return under the first tiles)
    f32x4 xres[4], posv[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        xres[a] = e_ld4(p.x32 + (size_t)m * EE + e_col(c, a), row_ok);
        posv[a] = p.pos ? e_ld4(p.pos + (size_t)m * EE + e_col(c, a), row_ok) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    e_touch(c, p.Wo, EE * EE * 2);
    if (p.mode == 0) { e_touch(c, p.W1, (unsigned)(p.F * EE * 2)); e_touch(c, p.W2, (unsigned)(EE * p.F * 2)); }
    if (proj) { e_touch(c, p.Wqk, 2 * EE * EE * 2); e_touch(c, p.Wv, EE * EE * 2); }
    e_load_rows(c, (const bf16_t*)p.o, m0, p.M, e_act(c, 0));
#pragma unroll
    for (int s = 0; s < ENS - 1; ++s) e_issue(c, unit_of);

    const bool do_drop = p.drop_p > 0.f;
    const uint32_t thresh = rt_drop_thresh(p.drop_p);
    const float ks = do_drop ? 1.f / (1.f - p.drop_p) : 1.f;
    const uint32_t s_d1 = do_drop ? rt_site_seed(p.seed_dev, p.seed_d1) : 0u, s_dh = do_drop ? rt_site_seed(p.seed_dev, p.seed_dh) : 0u,
                   s_d2 = do_drop ? rt_site_seed(p.seed_dev, p.seed_d2) : 0u;
    const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};

    // LayerNorm of this thread's values of row m (statistics as rt_layernorm_fwd: mean, then the centred second moment)
    auto layer_norm = [&](const f32x4 (&v)[4], const float* gamma, const float* beta, float* mean_out, float* rstd_out, f32x4 (&y)[4]) __attribute__((always_inline)) {
        float s = 0.f;
#pragma unroll
        for (int a = 0; a < 4; ++a) s += (v[a][0] + v[a][1]) + (v[a][2] + v[a][3]);
        const float mean = e_row_sum(c, s, e_red(c, 0)) * (1.f / EE);
        float ss = 0.f;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = v[a][e] - mean; ss += d * d; }
        const float rstd = rsqrtf(e_row_sum(c, ss, e_red(c, 1)) * (1.f / EE) + p.eps);
        if (row_ok && c.lg == 0 && c.wn == 0) { mean_out[m] = mean; rstd_out[m] = rstd; }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + e_col(c, a)), b = *reinterpret_cast<const f32x4*>(beta + e_col(c, a));
#pragma unroll
            for (int e = 0; e < 4; ++e) y[a][e] = (v[a][e] - mean) * rstd * g[e] + b[e];
        }
    };
    auto to_bf16 = [&](const f32x4 (&v)[4], bf16x4_t (&o)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int e = 0; e < 4; ++e) o[a][e] = (bf16_t)v[a][e];
    };
    auto st_bf16 = [&](void* dst, int ld, int col0, const bf16x4_t (&o)[4]) __attribute__((always_inline)) {
        if (!row_ok || !dst) return;
#pragma unroll
        for (int a = 0; a < 4; ++a) e_st_bf16x4(dst, (size_t)m * ld + col0 + e_col(c, a), o[a]);
    };
    auto st_f32 = [&](float* dst, const f32x4 (&v)[4]) __attribute__((always_inline)) {
        if (!row_ok || !dst) return;
#pragma unroll
        for (int a = 0; a < 4; ++a) e_st_f32x4(dst, (size_t)m * EE + e_col(c, a), v[a]);
    };

    // ---- out_proj + dropout1 + residual (rt_conv_gemm epilogue order: +bias -> dropout -> +res) -> norm1
    f32x4 acc[4] = {z4, z4, z4, z4};
    e_unit(c, unit_of, e_act(c, 0), acc);
    f32x4 tv[4], x1[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(p.bo + e_col(c, a));
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = acc[a][e] + b[e];
            if (do_drop) v = (rt_hash32(s_d1, (uint32_t)(m * EE + e_col(c, a) + e)) >= thresh) ? v * ks : 0.f;
            tv[a][e] = v + xres[a][e];
        }
    }
    st_f32(p.t, tv);
    layer_norm(tv, p.g1, p.be1, p.mean1, p.rstd1, x1);
    bf16x4_t hb[4];
    to_bf16(x1, hb);
    st_bf16(p.x1_16, EE, 0, hb);
    if (p.mode == 1) { st_f32(p.x1_32, x1); e_wait_vmcnt<0>(); return; }
    e_store_act(c, e_act(c, 1), hb);                    // linear1's operand (read behind the first barrier of the next unit)

    // ---- feed-forward, 256 hidden units at a time: h_c = dropout(relu(x1 W1_c^T + b1_c)); t2 += h_c W2[:, c]^T
    f32x4 t2[4] = {z4, z4, z4, z4};
#pragma unroll 1
    for (int cc = 0; cc < C; ++cc) {
        f32x4 h[4] = {z4, z4, z4, z4};
        e_unit(c, unit_of, e_act(c, 1), h);
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int f = cc * 256 + e_col(c, a);
            const f32x4 b = *reinterpret_cast<const f32x4*>(p.b1 + f);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = fmaxf(h[a][e] + b[e], 0.f);
                if (do_drop) v = (rt_hash32(s_dh, (uint32_t)(m * p.F + f + e)) >= thresh) ? v * ks : 0.f;
                hb[a][e] = (bf16_t)v;
            }
        }
        st_bf16(p.hdn, p.F, cc * 256, hb);
        e_store_act(c, e_act(c, 2), hb);                // (the previous chunk's linear2 reads of this buffer ended four barriers ago)
        e_unit(c, unit_of, e_act(c, 2), t2);
    }
    // ---- + bias, dropout2, residual -> norm2 (+ pos)
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(p.b2 + e_col(c, a));
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = t2[a][e] + b[e];
            if (do_drop) v = (rt_hash32(s_d2, (uint32_t)(m * EE + e_col(c, a) + e)) >= thresh) ? v * ks : 0.f;
            t2[a][e] = v + x1[a][e];
        }
    }
    st_f32(p.t2, t2);
    f32x4 x2[4];
    layer_norm(t2, p.g2, p.be2, p.mean2, p.rstd2, x2);
    st_f32(p.x2_32, x2);
    bf16x4_t xb[4], xpb[4];
    to_bf16(x2, xb);
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int e = 0; e < 4; ++e) xpb[a][e] = (bf16_t)(x2[a][e] + posv[a][e]);
    st_bf16(p.x2_16, EE, 0, xb);
    st_bf16(p.x2p16, EE, 0, xpb);
    if (!proj) { e_wait_vmcnt<0>(); return; }           // (the over-issued tail tiles must land before the LDS is released)
    // ---- the next layer's projections: q | k from x2 + pos, v from x2
    e_store_act(c, e_act(c, 0), xb);
    e_store_act(c, e_act(c, 1), xpb);                   // (x1's operand rows are dead: the last linear1 unit ended four barriers ago)
#pragma unroll 1
    for (int u = 0; u < 3; ++u) {
        f32x4 q[4] = {z4, z4, z4, z4};
        e_unit(c, unit_of, e_act(c, u < 2 ? 1 : 0), q);
        const float* bias = u < 2 ? p.bqk + u * 256 : p.bv;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(bias + e_col(c, a));
#pragma unroll
            for (int e = 0; e < 4; ++e) hb[a][e] = (bf16_t)(q[a][e] + b[e]);
        }
        if (u < 2) st_bf16(p.qk, 2 * EE, u * 256, hb); else st_bf16(p.v, EE, 0, hb);
    }
    e_wait_vmcnt<0>();
}

// ------------------------------------------------------------------------------------------------------------------ backward
__global__ __launch_bounds__(ENT, 1) void enc_tail_bwd_kernel(const rt_enc_tail_bwd_desc p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int C = p.mode == 1 ? 0 : p.F >> 8;
    ECtx c;
    ectx_init(c, smem, 4 * (2 * C + 1));
    auto unit_of = [&](int u) __attribute__((always_inline)) -> EUnit {
        if (u < 2 * C) {
            const int cc = u >> 1;
            if (u & 1) return EUnit{(const bf16_t*)p.WT1, (unsigned)(EE * p.F * 2), 0, p.F, cc * 256};      // dx1 += dhdn_c WT1[:, c]^T  (WT1: [256][F])
            return EUnit{(const bf16_t*)p.WT2, (unsigned)(p.F * EE * 2), cc * 256, EE, 0};                  // dhdn_c = dt2b WT2_c^T      (WT2: [F][256])
        }
        return EUnit{(const bf16_t*)p.WTo, (unsigned)(EE * EE * 2), 0, EE, 0};
    };
    const int m0 = blockIdx.x * RB;
    const int m = m0 + c.wm * 16 + c.li;
    const bool row_ok = m < p.M;
    const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool full = p.mode == 0;
    f32x4 dy[4], xv2[4], xv1[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const size_t o = (size_t)m * EE + e_col(c, a);
        dy[a] = e_ld4((full ? p.dy : p.dx1) + o, row_ok);            // mode 1: dy carries dx1
        if (full && p.dy2) dy[a] += e_ld4(p.dy2 + o, row_ok);
        xv2[a] = full ? e_ld4(p.t2 + o, row_ok) : z4;
        xv1[a] = e_ld4(p.t + o, row_ok);
    }
    const float mean2 = row_ok && full ? p.mean2[m] : 0.f, rstd2 = row_ok && full ? p.rstd2[m] : 0.f;
    const float mean1 = row_ok ? p.mean1[m] : 0.f, rstd1 = row_ok ? p.rstd1[m] : 0.f;
    if (full) { e_touch(c, p.WT2, (unsigned)(p.F * EE * 2)); e_touch(c, p.WT1, (unsigned)(EE * p.F * 2)); }
    e_touch(c, p.WTo, EE * EE * 2);
#pragma unroll
    for (int s = 0; s < ENS - 1; ++s) e_issue(c, unit_of);

    const bool do_drop = p.drop_p > 0.f;
    const uint32_t thresh = rt_drop_thresh(p.drop_p);
    const float ks = do_drop ? 1.f / (1.f - p.drop_p) : 1.f;
    const uint32_t s_d1 = do_drop ? rt_site_seed(p.seed_dev, p.seed_d1) : 0u, s_d2 = do_drop ? rt_site_seed(p.seed_dev, p.seed_d2) : 0u;

    // LayerNorm backward of this thread's values of row m (rt_layernorm_bwd's arithmetic); the parameter-gradient column sums of
    // the workgroup's 32 rows go to partials[block][0 / 1][256] (rt_ln_param_grad_grouped reduces them)
    auto ln_bwd = [&](const f32x4 (&g_in)[4], const f32x4 (&xin)[4], float mean, float rstd, const float* gamma, float* partials,
                      f32x4 (&dx)[4]) __attribute__((always_inline)) {
#pragma clang fp contract(off)
        f32x4 xh[4], g[4];
        float s1 = 0.f, s2 = 0.f;
        float* cb = e_colbuf(c);                          // [2 (wm)][2][256]
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const f32x4 gam = *reinterpret_cast<const f32x4*>(gamma + e_col(c, a));
            f32x4 cg, cbeta;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                xh[a][e] = (xin[a][e] - mean) * rstd;
                const float de = row_ok ? g_in[a][e] : 0.f;
                cg[e] = de * xh[a][e]; cbeta[e] = de;
                g[a][e] = de * gam[e];
                s1 += g[a][e]; s2 += g[a][e] * xh[a][e];
            }
            // column sums over the wave's 16 rows (the 16 lanes li of a lane group)
#pragma unroll
            for (int o = 1; o < 16; o <<= 1)
#pragma unroll
                for (int e = 0; e < 4; ++e) { cg[e] += __shfl_xor(cg[e], o, 64); cbeta[e] += __shfl_xor(cbeta[e], o, 64); }
            if (c.li == 0) {
                *reinterpret_cast<f32x4*>(cb + (c.wm * 2 + 0) * EE + e_col(c, a)) = cg;
                *reinterpret_cast<f32x4*>(cb + (c.wm * 2 + 1) * EE + e_col(c, a)) = cbeta;
            }
        }
        s1 = e_row_sum(c, s1, e_red(c, 0)) * (1.f / EE);
        s2 = e_row_sum(c, s2, e_red(c, 1)) * (1.f / EE);      // (both barriers are behind the column-sum stores)
        if (partials && c.t < EE * 2) {
            const int which = c.t >> 8, col = c.t & 255;
            partials[((size_t)blockIdx.x * 2 + which) * EE + col] = cb[(0 * 2 + which) * EE + col] + cb[(1 * 2 + which) * EE + col];
        }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int e = 0; e < 4; ++e) dx[a][e] = rstd * (g[a][e] - s1 - xh[a][e] * s2);
    };
    auto drop_bf16 = [&](const f32x4 (&v)[4], uint32_t seed, bf16x4_t (&o)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float d = v[a][e];
                if (do_drop) d = (rt_hash32(seed, (uint32_t)(m * EE + e_col(c, a) + e)) >= thresh) ? d * ks : 0.f;
                o[a][e] = (bf16_t)d;
            }
    };
    auto st_bf16 = [&](void* dst, int ld, int col0, const bf16x4_t (&o)[4]) __attribute__((always_inline)) {
        if (!row_ok || !dst) return;
#pragma unroll
        for (int a = 0; a < 4; ++a) e_st_bf16x4(dst, (size_t)m * ld + col0 + e_col(c, a), o[a]);
    };

    bf16x4_t ob[4];
    f32x4 dx1[4] = {z4, z4, z4, z4};
    f32x4 dt2[4] = {z4, z4, z4, z4};
    if (full) {
    // ---- norm2 backward: dt2 (fp32, the residual path) and dt2b = bf16(dt2 through dropout2): linear2's output gradient
    ln_bwd(dy, xv2, mean2, rstd2, p.g2, p.part2, dt2);
    drop_bf16(dt2, s_d2, ob);
    st_bf16(p.dt2b, EE, 0, ob);
    e_store_act(c, e_act(c, 0), ob);
    // ---- feed-forward backward, chunk by chunk: dhdn_c = gate(dt2b W2[:, c]) ; dx1 += dhdn_c W1_c
#pragma unroll 1
    for (int cc = 0; cc < C; ++cc) {
        bf16x4_t gate[4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
            gate[a] = row_ok ? *reinterpret_cast<const bf16x4_t*>((const bf16_t*)p.hdn + (size_t)m * p.F + cc * 256 + e_col(c, a))
                             : bf16x4_t{(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
        f32x4 dh[4] = {z4, z4, z4, z4};
        e_unit(c, unit_of, e_act(c, 0), dh);
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int e = 0; e < 4; ++e) ob[a][e] = (bf16_t)(((float)gate[a][e] > 0.f) ? dh[a][e] * p.gate_scale : 0.f);
        st_bf16(p.dhdn, p.F, cc * 256, ob);
        e_store_act(c, e_act(c, 1), ob);
        e_unit(c, unit_of, e_act(c, 1), dx1);
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) dx1[a] += dt2[a];           // linear1's input gradient + the residual path
    } else {
#pragma unroll
        for (int a = 0; a < 4; ++a) dx1[a] = dy[a];          // mode 1: the caller's launches produced it
    }
    // ---- norm1 backward -> dt (fp32, passed on: the layer input's residual path) and dtb = out_proj's output gradient
    f32x4 dt[4];
    ln_bwd(dx1, xv1, mean1, rstd1, p.g1, p.part1, dt);
    if (row_ok)
#pragma unroll
        for (int a = 0; a < 4; ++a) e_st_f32x4(p.dt, (size_t)m * EE + e_col(c, a), dt[a]);
    drop_bf16(dt, s_d1, ob);
    st_bf16(p.dtb, EE, 0, ob);
    e_store_act(c, e_act(c, 2), ob);
    f32x4 dO[4] = {z4, z4, z4, z4};
    e_unit(c, unit_of, e_act(c, 2), dO);
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int e = 0; e < 4; ++e) ob[a][e] = (bf16_t)dO[a][e];
    st_bf16(p.d_o, EE, 0, ob);
    e_wait_vmcnt<0>();
}

}  // namespace

static int enc_common_ok(int M, int F) { return (M > 0 && F >= 256 && (F & 255) == 0 && (long long)M * F < 0x3fffffffLL) ? RT_OK : RT_ERR_UNSUPPORTED; }

extern "C" int rt_enc_tail_fwd(const rt_enc_tail_fwd_desc* d, rt_stream_t stream) {
    if (!d || !d->o || !d->x32 || !d->Wo || !d->bo || !d->g1 || !d->be1 || !d->t || !d->mean1 || !d->rstd1 || !d->x1_16) return RT_ERR_BADARG;
    if (d->mode != 0 && d->mode != 1) return RT_ERR_BADARG;
    if (d->mode == 1 && !d->x1_32) return RT_ERR_BADARG;
    if (d->mode == 0 && (!d->W1 || !d->W2 || !d->b1 || !d->b2 || !d->g2 || !d->be2 || !d->hdn || !d->t2 || !d->mean2 || !d->rstd2 || !d->x2_32 || !d->x2_16))
        return RT_ERR_BADARG;
    if ((d->qk != nullptr) != (d->v != nullptr) || (d->qk && (!d->Wqk || !d->Wv || !d->bqk || !d->bv))) return RT_ERR_BADARG;
    if (d->x2p16 && !d->pos) return RT_ERR_BADARG;
    const int rc = enc_common_ok(d->M, d->F);
    if (rc != RT_OK) return rc;
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(enc_tail_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
        if (e != hipSuccess) return (int)e;
        attr = true;
    }
    hipLaunchKernelGGL(enc_tail_fwd_kernel, dim3((unsigned)((d->M + RB - 1) / RB)), dim3(ENT), SMEM_BYTES, (hipStream_t)stream, *d);
    RT_CHECK_LAUNCH();
    return RT_OK;
}

extern "C" int rt_enc_tail_bwd(const rt_enc_tail_bwd_desc* d, rt_stream_t stream) {
    if (!d || !d->t || !d->mean1 || !d->rstd1 || !d->g1 || !d->WTo || !d->dt || !d->dtb || !d->d_o) return RT_ERR_BADARG;
    if (d->mode != 0 && d->mode != 1) return RT_ERR_BADARG;
    if (d->mode == 1 && !d->dx1) return RT_ERR_BADARG;
    if (d->mode == 0 && (!d->dy || !d->t2 || !d->mean2 || !d->rstd2 || !d->g2 || !d->hdn || !d->WT2 || !d->WT1 || !d->dt2b || !d->dhdn)) return RT_ERR_BADARG;
    const int rc = enc_common_ok(d->M, d->F);
    if (rc != RT_OK) return rc;
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(enc_tail_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
        if (e != hipSuccess) return (int)e;
        attr = true;
    }
    hipLaunchKernelGGL(enc_tail_bwd_kernel, dim3((unsigned)((d->M + RB - 1) / RB)), dim3(ENT), SMEM_BYTES, (hipStream_t)stream, *d);
    RT_CHECK_LAUNCH();
    return RT_OK;
}
