// The decoder's query chain as ONE cooperative launch (transformer.py:231-252 x dec_layers, one query per image).
//
// With a single referring query per image the decoder works on M = B rows: every Linear is a 16-column-tile GEMV whose cost
// as a launch is the launch itself (~5 us per link, ~66 links per forward).  Here DEC_G workgroups stay resident for the whole
// stack and hand the M x 256 row block from stage to stage through global memory:
//   * the hand-off carries its own validity: a producer writes its columns as 8-byte units {4 bytes of data, 4-byte tag} with
//     write-through stores (sc0 sc1) and moves on -- no store acknowledgement, no counter; a consumer reads the units with
//     cache-bypassing loads and simply re-reads until every tag is the one of (this launch, this layer, this stage).  One memory
//     round trip per hand-off instead of three (store ack -> counter -> poll -> data read: the first version of this kernel,
//     5.2 us per stage measured with the stage trace below; a counter-only barrier probes at 1.6-2.1 us,
//     benchmarks/probes/grid_barrier_probe.hip) against ~4-5 us per launched link.  The tag's launch part is a word in the
//     hand-off buffer that workgroup 0 advances at the very end, so the buffer never needs clearing;
//   * everything that does not depend on the row block -- the stage's weight fragments, biases, the (b, h) K / V rows of the
//     cross-attention -- is requested BEFORE the wait, so it arrives while the workgroup polls;
//   * LayerNorm is not a stage: each consumer recomputes it on the M <= 16 rows (one wave per row, the arithmetic of
//     layernorm_fwd_vec_kernel) and keeps the fp32 result in LDS as the residual of the stage after next;
//   * four waves per workgroup (one per SIMD: the whole 512-register file per lane, the K / V rows of an attention stage stay in
//     registers across the wait); thread 0 polls -- its first poll returns behind its own prefetches, which the stage needs anyway.
// Arithmetic is the launched chain's, operation for operation (the K split over four waves and the reduction order of
// skinny_gemm_kernel, attn_q1_fwd_kernel's softmax, the same dropout sites and indices): the outputs are bit-identical to the
// chain's and the launched backward consumes the saved tensors unchanged (tests/test_decoder_coop_gpu.py).
// Stage map of one layer:  S1 v-proj + head dropout -> S2 out_proj + residual -> [LN1] S3 q-proj -> S4 cross-attention ->
// S5 out_proj + residual -> [LN2] S6 linear1 + relu -> S7 linear2 + residual -> [LN3] next layer's S1.  The 256-wide products
// (S1-S3, S5, S7) run on the 16 "core" workgroups (one 16-column tile each); the cross-attention (one (b, h) per workgroup) and
// linear1 (F / 16 tiles) on the G - 16 "workers", which therefore hold their K / V rows and weight fragments long before the
// rows they wait for exist.
#include "rt_common.h"
#include <stdlib.h>

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

constexpr int DEC_E = 256;            // model width (16 column tiles: the "core" workgroups 0..15 own one each)
constexpr int DEC_CORE = DEC_E / 16;
constexpr int DEC_COH = 17;           // buffer cache policy sc0 | sc1
constexpr int DEC_MAXK = 3;           // keys per thread of the attention stage: S <= 768
constexpr int DEC_NT = 4;             // linear1 column tiles per workgroup (F / 16 / G <= 4)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t dec_rsrc(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ u32x4 dec_ld16(const void* base, int byte_off) {          // coherent (cache-bypassing) 16-B load
    return __builtin_amdgcn_raw_buffer_load_b128(dec_rsrc(base), byte_off, 0, DEC_COH);
}
__device__ __forceinline__ void dec_st16(void* base, int byte_off, u32x4 v) {        // write-through stores
    __builtin_amdgcn_raw_buffer_store_b128(v, dec_rsrc(base), byte_off, 0, DEC_COH);
}
__device__ __forceinline__ void dec_st8(void* base, int byte_off, u32x2 v) {
    __builtin_amdgcn_raw_buffer_store_b64(v, dec_rsrc(base), byte_off, 0, DEC_COH);
}
__device__ __forceinline__ void dec_st2(void* base, int byte_off, bf16_t v) {
    __builtin_amdgcn_raw_buffer_store_b16(*reinterpret_cast<unsigned short*>(&v), dec_rsrc(base), byte_off, 0, DEC_COH);
}

struct DecSmem {
    f32x4 red[4][64];
    float ln32[16][DEC_E];           // fp32 LayerNorm output of the stage before: the residual of the next product
    float sm32[4][32];
    float redf[8];
    int err;
};

// ---- hand-off through tagged units --------------------------------------------------------------------------------------------
// Region layout (bytes from the buffer start; words 0 / 1 = launch epoch / failure flag):
constexpr int LL_HDR = 256;
constexpr int LL_O = LL_HDR, LL_U = LL_O + 16 * 256 * 4, LL_Q2 = LL_U + 16 * 256 * 8, LL_O2 = LL_Q2 + 16 * 256 * 4,
              LL_U2 = LL_O2 + 16 * 256 * 4, LL_U3 = LL_U2 + 16 * 256 * 8, LL_HDN = LL_U3 + 16 * 256 * 8;      // hdn: 16 * F * 4 bytes
__device__ __forceinline__ unsigned ll_tag(unsigned epoch, int layer, int stage) { return (epoch << 8) | (unsigned)(layer * 8 + stage + 1); }

// bf16 row block [M][K] (unit = 2 bf16 + tag) -> LDS operand rows; NL 16-byte loads per thread in flight per round trip
template <int NL>
__device__ __forceinline__ void ll_rows_to_lds(unsigned char* ll, int region, unsigned tag, int M, int K, bf16_t* xa, int ld,
                                               unsigned* err, int spin) {
    const int t = threadIdx.x;
    const int per_row = K >> 2, pieces = M * per_row;            // 16-byte pieces: 2 units = 4 bf16
    for (int i0 = 0; i0 < pieces; i0 += 256 * NL) {
        u32x4 v[NL];
        int guard = 0;
        bool ok;
        do {
            asm volatile("" ::: "memory");       // the loads below must be re-issued on every pass
#pragma unroll
            for (int j = 0; j < NL; ++j) {
                const int i = i0 + j * 256 + t;
                if (i < pieces) v[j] = dec_ld16(ll + region, i * 16);
            }
            ok = true;
#pragma unroll
            for (int j = 0; j < NL; ++j) {
                const int i = i0 + j * 256 + t;
                if (i < pieces) ok = ok && v[j][1] == tag && v[j][3] == tag;
            }
        } while (!ok && ++guard < spin);
        if (!ok) *err = 1u;
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            const int i = i0 + j * 256 + t;
            if (i < pieces) {
                const int r = i / per_row, c = i - r * per_row;
                *reinterpret_cast<u32x2*>(xa + r * ld + c * 4) = u32x2{v[j][0], v[j][2]};
            }
        }
    }
}
// producer side: 4 consecutive bf16 features of row m (one 16-byte store), 4 consecutive fp32 features (two)
__device__ __forceinline__ void ll_put_bf16x4(unsigned char* ll, int region, unsigned tag, int elem, u32x2 packed) {
    dec_st16(ll + region, elem * 4, u32x4{packed[0], tag, packed[1], tag});
}
__device__ __forceinline__ void ll_put_f32x4(unsigned char* ll, int region, unsigned tag, int elem, f32x4 v) {
    dec_st16(ll + region, elem * 8, u32x4{__float_as_uint(v[0]), tag, __float_as_uint(v[1]), tag});
    dec_st16(ll + region, elem * 8 + 16, u32x4{__float_as_uint(v[2]), tag, __float_as_uint(v[3]), tag});
}

template <bool COH>
__device__ __forceinline__ void dec_rows_to_lds(const bf16_t* src, int M, int K, bf16_t* xa, int ld) {
    const int t = threadIdx.x;
    const int per_row = K >> 3, pieces = M * per_row;
    for (int i = t; i < pieces; i += 256) {
        const int r = i / per_row, c = i - r * per_row;
        *reinterpret_cast<u32x4*>(xa + r * ld + c * 8) = *reinterpret_cast<const u32x4*>(src + (size_t)i * 8);
    }
}

// ---- LayerNorm of the M rows (layernorm_fwd_vec_kernel<1>'s arithmetic): fp32 result -> ln32, bf16(y [+ pos]) -> xa ----------
struct DecLnOut { float* y_f32; bf16_t* y_bf16; bf16_t* ypos_bf16; float* mean; float* rstd; };
// `ll` != nullptr: the rows come from a tagged fp32 region (4 units per lane), else from `u` (plain memory)
__device__ __forceinline__ void dec_ln_rows(const float* u, unsigned char* ll, int region, unsigned tag, unsigned* err, int spin,
                                            int M, const float* gamma, const float* beta, const float* pos, float eps,
                                            DecSmem& sm, bf16_t* xa, int ld, bool writer, const DecLnOut& o) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane * 4;
    const f32x4 gam = *reinterpret_cast<const f32x4*>(gamma + c), bet = *reinterpret_cast<const f32x4*>(beta + c);
    for (int row = wave; row < M; row += 4) {
        f32x4 v;
        if (ll) {
            u32x4 a, b;
            int guard = 0;
            bool ok;
            do {
            asm volatile("" ::: "memory");       // the loads below must be re-issued on every pass
                a = dec_ld16(ll + region, (row * DEC_E + c) * 8);
                b = dec_ld16(ll + region, (row * DEC_E + c) * 8 + 16);
                ok = a[1] == tag && a[3] == tag && b[1] == tag && b[3] == tag;
            } while (!ok && ++guard < spin);
            if (!ok) *err = 1u;
            v = f32x4{__uint_as_float(a[0]), __uint_as_float(a[2]), __uint_as_float(b[0]), __uint_as_float(b[2])};
        } else {
            v = *reinterpret_cast<const f32x4*>(u + (size_t)row * DEC_E + c);
        }
        const f32x4 ps = pos ? *reinterpret_cast<const f32x4*>(pos + (size_t)row * DEC_E + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        const float s = (v[0] + v[1]) + (v[2] + v[3]);
        const float mean = rt_wave_sum(s) * (1.f / DEC_E);
        float ss = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = v[e] - mean; ss += d * d; }
        const float rstd = rsqrtf(rt_wave_sum(ss) * (1.f / DEC_E) + eps);
        f32x4 y;
        bf16x4 yb, yp;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            y[e] = (v[e] - mean) * rstd * gam[e] + bet[e];
            yb[e] = (bf16_t)y[e];
            yp[e] = (bf16_t)(y[e] + ps[e]);
        }
        *reinterpret_cast<f32x4*>(&sm.ln32[row][c]) = y;
        *reinterpret_cast<bf16x4*>(xa + row * ld + c) = pos ? yp : yb;
        if (writer) {
            const size_t off = (size_t)row * DEC_E + c;
            if (lane == 0) { if (o.mean) o.mean[row] = mean; if (o.rstd) o.rstd[row] = rstd; }
            if (o.y_f32) *reinterpret_cast<f32x4*>(o.y_f32 + off) = y;
            if (o.y_bf16) *reinterpret_cast<bf16x4*>(o.y_bf16 + off) = yb;
            if (o.ypos_bf16) *reinterpret_cast<bf16x4*>(o.ypos_bf16 + off) = yp;
        }
    }
}

// ---- weight fragments of one 16-column tile: lane (li, lg) holds W[n0 + li][32 ks + 8 lg .. + 7] for its wave's K steps ------
template <int PER>
__device__ __forceinline__ void dec_load_w(const bf16_t* W, int K, int n0, u32x4 (&wv)[PER]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave < 4) {
        const bf16_t* row = W + (size_t)(n0 + (lane & 15)) * K + (lane >> 4) * 8 + wave * PER * 32;
#pragma unroll
        for (int j = 0; j < PER; ++j) wv[j] = *reinterpret_cast<const u32x4*>(row + j * 32);
    }
}

// one tile: K split over the four compute waves (PER 32-wide steps each, ascending), partial sums added wave 0 + 1 + 2 + 3, the
// epilogue on wave 0: lane (li, lg) owns row m = li, features n0 + 4 lg .. + 3 -- skinny_gemm_kernel's schedule exactly.
template <int PER, class Epi>
__device__ __forceinline__ void dec_tile(const u32x4 (&wv)[PER], const bf16_t* xa, int ld, int M, int n0, DecSmem& sm, Epi epi) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, lg = lane >> 4;
    if (wave < 4) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const bf16_t* xr = xa + li * ld + lg * 8 + wave * PER * 32;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const bf16x8 xv = *reinterpret_cast<const bf16x8*>(xr + j * 32);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&wv[j]), xv, acc, 0, 0, 0);
        }
        sm.red[wave][lane] = acc;
    }
    __syncthreads();
    if (wave == 0 && li < M) {
        const f32x4 acc = sm.red[0][lane] + sm.red[1][lane] + sm.red[2][lane] + sm.red[3][lane];
        epi(li, n0 + lg * 4, acc);
    }
    __syncthreads();
}

__device__ __forceinline__ f32x4 dec_dropout(f32x4 v, float p, uint32_t seed, uint32_t idx0, int shift) {
    const uint32_t thresh = rt_drop_thresh(p);
    const float ks = 1.0f / (1.0f - p);
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = (rt_hash32(seed, (idx0 + r) >> shift) >= thresh) ? v[r] * ks : 0.f;
    return v;
}
__device__ __forceinline__ u32x2 dec_pack4(f32x4 v) {
    bf16x4 b;
#pragma unroll
    for (int r = 0; r < 4; ++r) b[r] = (bf16_t)v[r];
    return *reinterpret_cast<u32x2*>(&b);
}

// ---- cross-attention of one (b, h): attn_q1_fwd_kernel's arithmetic; K / V rows arrive before the wait ------------------------
struct DecRow { bf16x8 c[4]; };
struct DecKV { DecRow k[DEC_MAXK], v[DEC_MAXK]; bool ok[DEC_MAXK]; };
__device__ __forceinline__ void dec_attn_prefetch(const rt_decoder_fwd_desc& p, const rt_decoder_layer_fwd& L, int bh, DecKV& kv) {
    const int t = threadIdx.x;
    if (t < 256) {
        const int b = bh / p.H, h = bh - b * p.H;
#pragma unroll
        for (int i = 0; i < DEC_MAXK; ++i) {
            const int j = t + i * 256;
            const int jj = j < p.S ? j : 0;
            kv.ok[i] = j < p.S && !(p.kpm && p.kpm[(size_t)b * p.S + jj]);
            const bf16_t* kr = (const bf16_t*)L.k2 + ((size_t)b * p.S + jj) * p.ldkv + h * 32;
            const bf16_t* vr = (const bf16_t*)L.v2 + ((size_t)b * p.S + jj) * p.ldkv + h * 32;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                kv.k[i].c[c] = *reinterpret_cast<const bf16x8*>(kr + c * 8);
                kv.v[i].c[c] = *reinterpret_cast<const bf16x8*>(vr + c * 8);
            }
        }
    }
}
__device__ __forceinline__ void dec_attn(const rt_decoder_fwd_desc& p, const rt_decoder_layer_fwd& L, int bh, const DecKV& kv,
                                         DecSmem& sm, uint32_t dseed, unsigned char* ll, unsigned tag_q, unsigned tag_o, unsigned* err, int spin) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int b = bh / p.H, h = bh - b * p.H;
    float q[32], sc[DEC_MAXK], o[32];
    float m = -INFINITY;
    {   // the 32 query features of (b, h): 16 tagged units, the same 128 bytes for every lane
        u32x4 raw[8];
        int guard = 0;
        bool ok;
        do {
            asm volatile("" ::: "memory");       // the loads below must be re-issued on every pass
            ok = true;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                raw[c] = dec_ld16(ll + LL_Q2, (b * DEC_E + h * 32 + c * 4) * 4);
                ok = ok && raw[c][1] == tag_q && raw[c][3] == tag_q;
            }
        } while (!ok && ++guard < spin);
        if (!ok) *err = 1u;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const unsigned w0 = raw[c][0], w1 = raw[c][2];
            const bf16x2 lo = *reinterpret_cast<const bf16x2*>(&w0), hi = *reinterpret_cast<const bf16x2*>(&w1);
            q[c * 4 + 0] = (float)lo[0]; q[c * 4 + 1] = (float)lo[1]; q[c * 4 + 2] = (float)hi[0]; q[c * 4 + 3] = (float)hi[1];
        }
    }
#pragma unroll
    for (int i = 0; i < DEC_MAXK; ++i) {
        float a = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int e = 0; e < 8; ++e) a += (float)kv.k[i].c[c][e] * q[c * 8 + e];
        sc[i] = kv.ok[i] ? a * p.scale : -INFINITY;
        m = fmaxf(m, sc[i]);
    }
    m = rt_wave_max(m);
    if (lane == 0) sm.redf[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(sm.redf[0], sm.redf[1]), fmaxf(sm.redf[2], sm.redf[3]));
    const float ms = (m == -INFINITY) ? 0.f : m;
    float l = 0.f;
#pragma unroll
    for (int i = 0; i < DEC_MAXK; ++i) { sc[i] = __expf(sc[i] - ms); l += sc[i]; }
    l = rt_wave_sum(l);
    if (lane == 0) sm.redf[4 + wave] = l;
    __syncthreads();
    l = sm.redf[4] + sm.redf[5] + sm.redf[6] + sm.redf[7];
    const float inv_l = 1.f / l;                 // fully masked row: NaN below, as the reference
    if (t == 0) L.lse2[bh] = ms + __logf(l);
    const bool do_drop = p.drop_p > 0.f;
    const uint32_t thresh = rt_drop_thresh(p.drop_p);
    const float ks = do_drop ? 1.f / (1.f - p.drop_p) : 1.f;
#pragma unroll
    for (int d = 0; d < 32; ++d) o[d] = 0.f;
#pragma unroll
    for (int i = 0; i < DEC_MAXK; ++i) {
        const int j = t + i * 256;
        if (j >= p.S) continue;
        float pr = sc[i] * inv_l;
        if (do_drop) pr = (rt_hash32(dseed, (uint32_t)((size_t)bh * p.S + j)) >= thresh) ? pr * ks : 0.f;
        if (pr != 0.f || pr != pr) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int e = 0; e < 8; ++e) o[c * 8 + e] += pr * (float)kv.v[i].c[c][e];
        }
    }
#pragma unroll
    for (int d = 0; d < 32; ++d) o[d] = rt_wave_sum(o[d]);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int d = 0; d < 32; ++d) sm.sm32[wave][d] = o[d];
    }
    __syncthreads();
    if (t < 16) {                                // thread t: features 2t, 2t + 1 of the head -> one unit + the saved bf16 pair
        bf16x2 pr;
        pr[0] = (bf16_t)(sm.sm32[0][2 * t] + sm.sm32[1][2 * t] + sm.sm32[2][2 * t] + sm.sm32[3][2 * t]);
        pr[1] = (bf16_t)(sm.sm32[0][2 * t + 1] + sm.sm32[1][2 * t + 1] + sm.sm32[2][2 * t + 1] + sm.sm32[3][2 * t + 1]);
        const int e = b * DEC_E + h * 32 + 2 * t;
        const unsigned bits = *reinterpret_cast<const unsigned*>(&pr);
        dec_st8(ll + LL_O2, e * 4, u32x2{bits, tag_o});
        *reinterpret_cast<unsigned*>((bf16_t*)L.o2 + e) = bits;
    }
}

__global__ __launch_bounds__(256) void decoder_fwd_kernel(const rt_decoder_fwd_desc p, const int G, const int spin, unsigned* trace) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dec_smem_raw[];
    DecSmem& sm = *reinterpret_cast<DecSmem*>(dec_smem_raw);
    bf16_t* xa = reinterpret_cast<bf16_t*>(dec_smem_raw + sizeof(DecSmem));       // [16][K + 8] bf16 operand rows of the current product
    const int wg = blockIdx.x, t = threadIdx.x;
    const int M = p.M, F = p.F;
    const bool core = wg < DEC_CORE, writer = wg == 0;
    const int wk = wg - DEC_CORE, NW = G - DEC_CORE;         // worker id / count: attention and linear1 run on the workers
    const int ldE = DEC_E + 8, ldF = F + 8;
    const int n0 = wg * 16;                                  // a core workgroup's column tile of every 256-wide product
    const int n_bh = M * p.H;
    const int f_tiles = F >> 4;
    const bool drop = p.drop_p > 0.f;
    unsigned char* ll = reinterpret_cast<unsigned char*>(p.handoff);
    unsigned* err = p.handoff + 1;
    const unsigned epoch = p.handoff[0];                     // advanced by workgroup 0 when the whole stack is done
    const int lane = t & 63, lg = lane >> 4, wave = t >> 6;

    // REFTR_DEC_TRACE: workgroups 0 (core) and G - 1 stamp the 100 MHz wall clock at every stage boundary
    int tr_i = 0;
    unsigned* tr = (trace && t == 0 && (wg == 0 || wg == G - 1)) ? trace + (wg == 0 ? 0 : 512) : nullptr;
#define DEC_STAMP() do { if (tr) tr[tr_i++] = (unsigned)wall_clock64(); } while (0)
    DEC_STAMP();
    u32x4 w2[2];            // K = 256: two 32-wide steps per wave
    if (core) dec_load_w<2>((const bf16_t*)p.layer[0].Wv, DEC_E, n0, w2);
    for (int l = 0; l < p.n_layers; ++l) {
        const rt_decoder_layer_fwd& L = p.layer[l];
        if (core) {
            // ================= S1: o = headdrop(t16 Wv^T + bv); t = LN3 of the layer before (or the stack's input)
            f32x4 bias = {0.f, 0.f, 0.f, 0.f};
            if (wave == 0) bias = *reinterpret_cast<const f32x4*>(L.bv + n0 + lg * 4);
            if (l == 0) {
                dec_rows_to_lds<false>((const bf16_t*)p.t16, M, DEC_E, xa, ldE);
                for (int i = t; i < M * (DEC_E / 4); i += 256)
                    *reinterpret_cast<f32x4*>(&sm.ln32[0][0] + i * 4) = *reinterpret_cast<const f32x4*>(p.t32 + (size_t)i * 4);
            } else {
                const rt_decoder_layer_fwd& Lp = p.layer[l - 1];
                const DecLnOut out{Lp.t3_f32, (bf16_t*)Lp.t3_16, nullptr, Lp.mean3, Lp.rstd3};
                dec_ln_rows(nullptr, ll, LL_U3, ll_tag(epoch, l - 1, 6), err, spin, M, Lp.g3, Lp.be3, nullptr, p.eps, sm, xa, ldE, writer, out);
            }
            DEC_STAMP();
            __syncthreads();
            {
                const uint32_t seed = drop ? rt_site_seed(p.seed_dev, L.seed_ad) : 0u;
                const unsigned tag = ll_tag(epoch, l, 0);
                dec_tile<2>(w2, xa, ldE, M, n0, sm, [&](int m, int n, f32x4 v) {
                    v += bias;
                    if (drop) v = dec_dropout(v, p.drop_p, seed, (uint32_t)(m * DEC_E + n), 5);
                    const u32x2 pk = dec_pack4(v);
                    ll_put_bf16x4(ll, LL_O, tag, m * DEC_E + n, pk);
                    *reinterpret_cast<u32x2*>((bf16_t*)L.o + m * DEC_E + n) = pk;
                });
            }
            DEC_STAMP();
            // ================= S2: u = t + drop(o Wo^T + bo)
            dec_load_w<2>((const bf16_t*)L.Wo, DEC_E, n0, w2);
            if (wave == 0) bias = *reinterpret_cast<const f32x4*>(L.bo + n0 + lg * 4);
            ll_rows_to_lds<4>(ll, LL_O, ll_tag(epoch, l, 0), M, DEC_E, xa, ldE, err, spin);
            DEC_STAMP();
            __syncthreads();
            {
                const uint32_t seed = drop ? rt_site_seed(p.seed_dev, L.seed_d1) : 0u;
                const unsigned tag = ll_tag(epoch, l, 1);
                dec_tile<2>(w2, xa, ldE, M, n0, sm, [&](int m, int n, f32x4 v) {
                    v += bias;
                    if (drop) v = dec_dropout(v, p.drop_p, seed, (uint32_t)(m * DEC_E + n), 0);
                    v += *reinterpret_cast<const f32x4*>(&sm.ln32[m][n]);
                    ll_put_f32x4(ll, LL_U, tag, m * DEC_E + n, v);
                    *reinterpret_cast<f32x4*>(L.u + m * DEC_E + n) = v;
                });
            }
            DEC_STAMP();
            // ================= S3: q2 = (LN1(u) + query_pos) Wq^T + bq
            dec_load_w<2>((const bf16_t*)L.Wq, DEC_E, n0, w2);
            if (wave == 0) bias = *reinterpret_cast<const f32x4*>(L.bq + n0 + lg * 4);
            {
                const DecLnOut out{nullptr, nullptr, (bf16_t*)L.t1q16, L.mean1, L.rstd1};
                dec_ln_rows(nullptr, ll, LL_U, ll_tag(epoch, l, 1), err, spin, M, L.g1, L.be1, p.qpos, p.eps, sm, xa, ldE, writer, out);
            }
            DEC_STAMP();
            __syncthreads();
            {
                const unsigned tag = ll_tag(epoch, l, 2);
                dec_tile<2>(w2, xa, ldE, M, n0, sm, [&](int m, int n, f32x4 v) {
                    v += bias;
                    const u32x2 pk = dec_pack4(v);
                    ll_put_bf16x4(ll, LL_Q2, tag, m * DEC_E + n, pk);
                    *reinterpret_cast<u32x2*>((bf16_t*)L.q2 + m * DEC_E + n) = pk;
                });
            }
            DEC_STAMP();
        }
        // ================= S4: cross-attention, (b, h) = worker id, + number of workers, ...
        if (!core && wk < n_bh) {
            const uint32_t seed = drop ? rt_site_seed(p.seed_dev, L.seed_ad2) : 0u;
            DecKV kv;
            for (int bh = wk; bh < n_bh; bh += NW) {
                dec_attn_prefetch(p, L, bh, kv);
                dec_attn(p, L, bh, kv, sm, seed, ll, ll_tag(epoch, l, 2), ll_tag(epoch, l, 3), err, spin);
                __syncthreads();
            }
            DEC_STAMP();
        }
        if (core) {
            // ================= S5: u2 = t1 + drop(o2 Wo2^T + bo2)
            f32x4 bias = {0.f, 0.f, 0.f, 0.f};
            dec_load_w<2>((const bf16_t*)L.Wo2, DEC_E, n0, w2);
            if (wave == 0) bias = *reinterpret_cast<const f32x4*>(L.bo2 + n0 + lg * 4);
            ll_rows_to_lds<4>(ll, LL_O2, ll_tag(epoch, l, 3), M, DEC_E, xa, ldE, err, spin);
            DEC_STAMP();
            __syncthreads();
            {
                const uint32_t seed = drop ? rt_site_seed(p.seed_dev, L.seed_d2) : 0u;
                const unsigned tag = ll_tag(epoch, l, 4);
                dec_tile<2>(w2, xa, ldE, M, n0, sm, [&](int m, int n, f32x4 v) {
                    v += bias;
                    if (drop) v = dec_dropout(v, p.drop_p, seed, (uint32_t)(m * DEC_E + n), 0);
                    v += *reinterpret_cast<const f32x4*>(&sm.ln32[m][n]);
                    ll_put_f32x4(ll, LL_U2, tag, m * DEC_E + n, v);
                    *reinterpret_cast<f32x4*>(L.u2 + m * DEC_E + n) = v;
                });
            }
            DEC_STAMP();
        }
        // ================= S6: hdn = drop(relu(LN2(u2) W1^T + b1)), column tiles worker id, + number of workers, ...
        if (!core && wk < f_tiles) {
            u32x4 w1[DEC_NT][2];
            f32x4 b1[DEC_NT];
#pragma unroll
            for (int i = 0; i < DEC_NT; ++i) {
                const int tile = wk + i * NW;
                if (tile < f_tiles) {
                    dec_load_w<2>((const bf16_t*)L.W1, DEC_E, tile * 16, w1[i]);
                    if (wave == 0) b1[i] = *reinterpret_cast<const f32x4*>(L.b1 + tile * 16 + lg * 4);
                }
            }
            {
                const DecLnOut out{nullptr, (bf16_t*)L.t2_16, nullptr, L.mean2, L.rstd2};
                dec_ln_rows(nullptr, ll, LL_U2, ll_tag(epoch, l, 4), err, spin, M, L.g2, L.be2, nullptr, p.eps, sm, xa, ldE, wk == 0, out);
            }
            DEC_STAMP();
            __syncthreads();
            const uint32_t seed = drop ? rt_site_seed(p.seed_dev, L.seed_dh) : 0u;
            const unsigned tag = ll_tag(epoch, l, 5);
#pragma unroll
            for (int i = 0; i < DEC_NT; ++i) {
                const int tile = wk + i * NW;
                if (tile < f_tiles) {
                    const f32x4 bb = b1[i];
                    dec_tile<2>(w1[i], xa, ldE, M, tile * 16, sm, [&](int m, int n, f32x4 v) {
                        v += bb;
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                        if (drop) v = dec_dropout(v, p.drop_p, seed, (uint32_t)(m * F + n), 0);
                        const u32x2 pk = dec_pack4(v);
                        ll_put_bf16x4(ll, LL_HDN, tag, m * F + n, pk);
                        *reinterpret_cast<u32x2*>((bf16_t*)L.hdn + m * F + n) = pk;
                    });
                }
            }
            DEC_STAMP();
        }
        if (core) {
            // ================= S7: u3 = t2 + drop(hdn W2^T + b2)      (K = F: 16 steps per wave at F = 2048)
            u32x4 wf[16];
            f32x4 bias = {0.f, 0.f, 0.f, 0.f};
            dec_load_w<16>((const bf16_t*)L.W2, F, n0, wf);
            if (wave == 0) bias = *reinterpret_cast<const f32x4*>(L.b2 + n0 + lg * 4);
            {   // t2 = LN2(u2): this product's residual (the workers recompute it as linear1's operand)
                const DecLnOut none{nullptr, nullptr, nullptr, nullptr, nullptr};
                dec_ln_rows(nullptr, ll, LL_U2, ll_tag(epoch, l, 4), err, spin, M, L.g2, L.be2, nullptr, p.eps, sm, xa, ldE, false, none);
                __syncthreads();         // xa is reused for the hdn rows below
            }
            ll_rows_to_lds<16>(ll, LL_HDN, ll_tag(epoch, l, 5), M, F, xa, ldF, err, spin);
            DEC_STAMP();
            __syncthreads();
            {
                const uint32_t seed = drop ? rt_site_seed(p.seed_dev, L.seed_d3) : 0u;
                const unsigned tag = ll_tag(epoch, l, 6);
                dec_tile<16>(wf, xa, ldF, M, n0, sm, [&](int m, int n, f32x4 v) {
                    v += bias;
                    if (drop) v = dec_dropout(v, p.drop_p, seed, (uint32_t)(m * DEC_E + n), 0);
                    v += *reinterpret_cast<const f32x4*>(&sm.ln32[m][n]);
                    ll_put_f32x4(ll, LL_U3, tag, m * DEC_E + n, v);
                    *reinterpret_cast<f32x4*>(L.u3 + m * DEC_E + n) = v;
                });
            }
            DEC_STAMP();
            if (l + 1 < p.n_layers) dec_load_w<2>((const bf16_t*)p.layer[l + 1].Wv, DEC_E, n0, w2);
        }
    }
    // ---- the last layer's norm3 (statistics, bf16 rows, the fp32 rows the shared decoder norm reads); then the launch epoch moves on:
    // every workgroup read it at its start, and nothing of this launch is in flight once the last u3 is complete
    if (writer && p.n_layers > 0) {
        const rt_decoder_layer_fwd& Lp = p.layer[p.n_layers - 1];
        const DecLnOut out{Lp.t3_f32, (bf16_t*)Lp.t3_16, nullptr, Lp.mean3, Lp.rstd3};
        dec_ln_rows(nullptr, ll, LL_U3, ll_tag(epoch, p.n_layers - 1, 6), err, spin, M, Lp.g3, Lp.be3, nullptr, p.eps, sm, xa, ldE, true, out);
        if (t == 0) p.handoff[0] = epoch + 1;
        DEC_STAMP();
    }
}

// =====================================================================================================================================
// Backward of the stack, same machinery.  Per layer (last to first), hand-offs in brackets:
//   B1 LayerNorm3 backward of (dnorm + dta of the layer above) -> du3 (fp32, kept in LDS) / du3b     [every workgroup recomputes it]
//   B2 workers: dhdn = relu-gate(du3b W2) * 1/(1-p)                                                  [dhdn]
//   B3 core:    dt2 = du3 + dhdn W1                                                                  [dt2]
//   B4 LayerNorm2 backward -> du2 (LDS, until B9) / du2b       B5 core: do2 = du2b Wo2              [do2]
//   B6 workers: one-query attention backward of (b, h): dK, dV rows to memory, dq                    [dq2]
//   B8 core:    dt1q = dq2 Wq (+= into d query_pos)                                                  [dt1q]
//   B9 LayerNorm1 backward of (du2 + dt1q) -> du (LDS) / dub   B10 core: dv = head-mask(dub Wo)      [dv]
//   B11 core:   dta = du + dv Wv                                                                     [dta -> B1 of the layer below]
// Everything the launched chain leaves for the launches that stay outside is written in the chain's format and values.
struct DecSmemB {
    DecSmem a;
    float res2[16][DEC_E];           // du2: LayerNorm2's input gradient waits here for LayerNorm1's backward
    float pstage[4][2][DEC_E];       // one block of LayerNorm parameter-gradient partials (4 rows = 4 waves)
};

struct DecLnbSrc { const float* plain; const float* lds; unsigned char* ll; int region; unsigned tag; };

// LayerNorm backward of the M rows, layernorm_bwd_vec_kernel<1>'s arithmetic (one wave per row; block b of that kernel = rows 4b..4b+3 =
// waves 0..3 here, so the parameter-gradient partials are the same sums in the same order).
__device__ __forceinline__ void dec_lnb_rows(const DecLnbSrc& src, const float* x, const float* mean_p, const float* rstd_p, const float* gamma,
                                             int M, float drop2_p, uint32_t seed2, float (*res)[DEC_E], bf16_t* xa, int ld, bool writer,
                                             bf16_t* dxb_plain, float* partials, float (*pstage)[2][DEC_E], unsigned* err, int spin) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane * 4;
    const f32x4 gam = *reinterpret_cast<const f32x4*>(gamma + c);
    const bool do_drop2 = drop2_p > 0.f;
    const uint32_t thresh2 = rt_drop_thresh(drop2_p);
    const float ks2 = do_drop2 ? 1.f / (1.f - drop2_p) : 1.f;
    f32x4 dg[4], db[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) { dg[b] = f32x4{0.f, 0.f, 0.f, 0.f}; db[b] = dg[b]; }
#pragma unroll
    for (int b = 0; b < 4; ++b) {
#pragma clang fp contract(off)      // as in layernorm_bwd_vec_kernel
        const int row = b * 4 + wave;
        if (row >= M) continue;
        f32x4 d = {0.f, 0.f, 0.f, 0.f};
        bool have = false;
        if (src.plain) { d = *reinterpret_cast<const f32x4*>(src.plain + (size_t)row * DEC_E + c); have = true; }
        if (src.lds) { d = *reinterpret_cast<const f32x4*>(src.lds + row * DEC_E + c); have = true; }
        if (src.ll) {
            u32x4 a, q;
            int guard = 0;
            bool ok;
            do {
                asm volatile("" ::: "memory");
                a = dec_ld16(src.ll + src.region, (row * DEC_E + c) * 8);
                q = dec_ld16(src.ll + src.region, (row * DEC_E + c) * 8 + 16);
                ok = a[1] == src.tag && a[3] == src.tag && q[1] == src.tag && q[3] == src.tag;
            } while (!ok && ++guard < spin);
            if (!ok) *err = 1u;
            const f32x4 v = f32x4{__uint_as_float(a[0]), __uint_as_float(a[2]), __uint_as_float(q[0]), __uint_as_float(q[2])};
            d = have ? d + v : v;
        }
        const float mean = mean_p[row], rstd = rstd_p[row];
        const f32x4 xv = *reinterpret_cast<const f32x4*>(x + (size_t)row * DEC_E + c);
        f32x4 xh, g;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            xh[e] = (xv[e] - mean) * rstd;
            const float de = d[e];
            dg[b][e] += de * xh[e]; db[b][e] += de;
            g[e] = de * gam[e];
            s1 += g[e]; s2 += g[e] * xh[e];
        }
        s1 = rt_wave_sum(s1) * (1.f / DEC_E); s2 = rt_wave_sum(s2) * (1.f / DEC_E);
        f32x4 dx;
        bf16x4 bb;
        const int o = row * DEC_E + c;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            dx[e] = rstd * (g[e] - s1 - xh[e] * s2);
            float d2 = dx[e];
            if (do_drop2) d2 = (rt_hash32(seed2, (uint32_t)(o + e)) >= thresh2) ? d2 * ks2 : 0.f;
            bb[e] = (bf16_t)d2;
        }
        *reinterpret_cast<f32x4*>(&res[row][c]) = dx;
        *reinterpret_cast<bf16x4*>(xa + row * ld + c) = bb;
        if (writer && dxb_plain) *reinterpret_cast<bf16x4*>(dxb_plain + o) = bb;
    }
    if (writer && partials) {                      // uniform per workgroup
        const int nb = (M + 3) >> 2;
        for (int b = 0; b < nb; ++b) {
            f32x4 a = dg[0], q = db[0];
            if (b == 1) { a = dg[1]; q = db[1]; } else if (b == 2) { a = dg[2]; q = db[2]; } else if (b == 3) { a = dg[3]; q = db[3]; }
            *reinterpret_cast<f32x4*>(&pstage[wave][0][c]) = a;
            *reinterpret_cast<f32x4*>(&pstage[wave][1][c]) = q;
            __syncthreads();
            const int cc = threadIdx.x;
            partials[((size_t)b * 2) * DEC_E + cc] = pstage[0][0][cc] + pstage[1][0][cc] + pstage[2][0][cc] + pstage[3][0][cc];
            partials[((size_t)b * 2 + 1) * DEC_E + cc] = pstage[0][1][cc] + pstage[1][1][cc] + pstage[2][1][cc] + pstage[3][1][cc];
            __syncthreads();
        }
    }
}

// one-query attention backward of (b, h): attn_q1_bwd_kernel's arithmetic; do2 arrives as 16 tagged units
__device__ __forceinline__ void dec_attn_bwd(const rt_decoder_bwd_desc& p, const rt_decoder_layer_bwd& L, int bh, const DecKV& kv,
                                             DecSmem& sm, uint32_t dseed, unsigned char* ll, unsigned tag_in, unsigned tag_out,
                                             unsigned* err, int spin) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int b = bh / p.H, h = bh - b * p.H;
    float q[32], go[32], ov[32], dq[32];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const bf16x8 qv = *reinterpret_cast<const bf16x8*>((const bf16_t*)L.q2 + b * DEC_E + h * 32 + c * 8);
        const bf16x8 o8 = *reinterpret_cast<const bf16x8*>((const bf16_t*)L.o2 + b * DEC_E + h * 32 + c * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) { q[c * 8 + e] = (float)qv[e]; ov[c * 8 + e] = (float)o8[e]; }
    }
    {
        u32x4 raw[8];
        int guard = 0;
        bool ok;
        do {
            asm volatile("" ::: "memory");
            ok = true;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                raw[c] = dec_ld16(ll + LL_O2, (b * DEC_E + h * 32 + c * 4) * 4);
                ok = ok && raw[c][1] == tag_in && raw[c][3] == tag_in;
            }
        } while (!ok && ++guard < spin);
        if (!ok) *err = 1u;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const unsigned w0 = raw[c][0], w1 = raw[c][2];
            const bf16x2 lo = *reinterpret_cast<const bf16x2*>(&w0), hi = *reinterpret_cast<const bf16x2*>(&w1);
            go[c * 4 + 0] = (float)lo[0]; go[c * 4 + 1] = (float)lo[1]; go[c * 4 + 2] = (float)hi[0]; go[c * 4 + 3] = (float)hi[1];
        }
    }
    float delta = 0.f;
#pragma unroll
    for (int d = 0; d < 32; ++d) delta += go[d] * ov[d];
    const float lse = L.lse2[bh];
    const bool do_drop = p.drop_p > 0.f;
    const uint32_t thresh = rt_drop_thresh(p.drop_p);
    const float ks = do_drop ? 1.f / (1.f - p.drop_p) : 1.f;
#pragma unroll
    for (int d = 0; d < 32; ++d) dq[d] = 0.f;
#pragma unroll
    for (int i = 0; i < DEC_MAXK; ++i) {
        const int j = t + i * 256;
        if (j >= p.S) continue;
        float dotk = 0.f, dotv = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int e = 0; e < 8; ++e) dotk += (float)kv.k[i].c[c][e] * q[c * 8 + e];
        const float pr = kv.ok[i] ? __expf(dotk * p.scale - lse) : 0.f;
        float mk = 1.f;
        if (do_drop) mk = (rt_hash32(dseed, (uint32_t)((size_t)bh * p.S + j)) >= thresh) ? ks : 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int e = 0; e < 8; ++e) dotv += (float)kv.v[i].c[c][e] * go[c * 8 + e];
        const float ds = pr * (mk * dotv - delta) * p.scale;
        const float pd = pr * mk;
        bf16_t* dkr = (bf16_t*)L.dk2 + ((size_t)b * p.S + j) * p.ldkv + h * 32;
        bf16_t* dvr = (bf16_t*)L.dv2 + ((size_t)b * p.S + j) * p.ldkv + h * 32;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            bf16x8 a8, b8;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                a8[e] = (bf16_t)(ds * q[c * 8 + e]); b8[e] = (bf16_t)(pd * go[c * 8 + e]);
                dq[c * 8 + e] += ds * (float)kv.k[i].c[c][e];
            }
            *reinterpret_cast<bf16x8*>(dkr + c * 8) = a8;
            *reinterpret_cast<bf16x8*>(dvr + c * 8) = b8;
            if (L.dk2p) {
                *reinterpret_cast<bf16x8*>((bf16_t*)L.dk2p + ((size_t)b * p.S + j) * p.ldkvp + h * 32 + c * 8) = a8;
                *reinterpret_cast<bf16x8*>((bf16_t*)L.dv2p + ((size_t)b * p.S + j) * p.ldkvp + h * 32 + c * 8) = b8;
            }
        }
    }
#pragma unroll
    for (int d = 0; d < 32; ++d) dq[d] = rt_wave_sum(dq[d]);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int d = 0; d < 32; ++d) sm.sm32[wave][d] = dq[d];
    }
    __syncthreads();
    if (t < 16) {
        bf16x2 pr;
        pr[0] = (bf16_t)(sm.sm32[0][2 * t] + sm.sm32[1][2 * t] + sm.sm32[2][2 * t] + sm.sm32[3][2 * t]);
        pr[1] = (bf16_t)(sm.sm32[0][2 * t + 1] + sm.sm32[1][2 * t + 1] + sm.sm32[2][2 * t + 1] + sm.sm32[3][2 * t + 1]);
        const int e = b * DEC_E + h * 32 + 2 * t;
        const unsigned bits = *reinterpret_cast<const unsigned*>(&pr);
        dec_st8(ll + LL_Q2, e * 4, u32x2{bits, tag_out});
        *reinterpret_cast<unsigned*>((bf16_t*)L.dq2 + e) = bits;
    }
}

template <class LT>
__device__ __forceinline__ void dec_kv_prefetch(const LT& L, const uint8_t* kpm, int H, int S, int ldkv, int bh, DecKV& kv) {
    const int t = threadIdx.x;
    const int b = bh / H, h = bh - b * H;
#pragma unroll
    for (int i = 0; i < DEC_MAXK; ++i) {
        const int j = t + i * 256;
        const int jj = j < S ? j : 0;
        kv.ok[i] = j < S && !(kpm && kpm[(size_t)b * S + jj]);
        const bf16_t* kr = (const bf16_t*)L.k2 + ((size_t)b * S + jj) * ldkv + h * 32;
        const bf16_t* vr = (const bf16_t*)L.v2 + ((size_t)b * S + jj) * ldkv + h * 32;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            kv.k[i].c[c] = *reinterpret_cast<const bf16x8*>(kr + c * 8);
            kv.v[i].c[c] = *reinterpret_cast<const bf16x8*>(vr + c * 8);
        }
    }
}

__global__ __launch_bounds__(256) void decoder_bwd_kernel(const rt_decoder_bwd_desc p, const int G, const int spin) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dec_smem_raw[];
    DecSmemB& smb = *reinterpret_cast<DecSmemB*>(dec_smem_raw);
    DecSmem& sm = smb.a;
    bf16_t* xa = reinterpret_cast<bf16_t*>(dec_smem_raw + sizeof(DecSmemB));
    const int wg = blockIdx.x, t = threadIdx.x;
    const int M = p.M, F = p.F;
    const bool core = wg < DEC_CORE, writer = wg == 0;
    const int wk = wg - DEC_CORE, NW = G - DEC_CORE;
    const int ldE = DEC_E + 8, ldF = F + 8;
    const int n0 = wg * 16;
    const int n_bh = M * p.H;
    const int f_tiles = F >> 4;
    const bool drop = p.drop_p > 0.f;
    unsigned char* ll = reinterpret_cast<unsigned char*>(p.handoff);
    unsigned* err = p.handoff + 1;
    const unsigned epoch = p.handoff[0];
    const int lane = t & 63, lg = lane >> 4, wave = t >> 6;
    (void)wave;
    const int NL = p.n_layers;

    for (int l = NL - 1; l >= 0; --l) {
        const rt_decoder_layer_bwd& L = p.layer[l];
        // ---- requests that depend on nothing computed here
        u32x4 wA[DEC_NT][2];          // workers: linear2^T tiles; core: reused per product below
        u32x4 wf[16];
        if (!core) {
#pragma unroll
            for (int i = 0; i < DEC_NT; ++i) {
                const int tile = wk + i * NW;
                if (tile < f_tiles) dec_load_w<2>((const bf16_t*)L.WT2, DEC_E, tile * 16, wA[i]);
            }
        } else {
            dec_load_w<16>((const bf16_t*)L.WT1, F, n0, wf);
        }
        // ================= B1: LayerNorm3 backward
        {
            const DecLnbSrc src{L.dnorm, nullptr, l + 1 < NL ? ll : nullptr, LL_U3, ll_tag(epoch, l + 1, 6)};
            const uint32_t seed = drop ? rt_site_seed(p.seed_dev, L.seed_d3) : 0u;
            dec_lnb_rows(src, L.u3, L.mean3, L.rstd3, L.g3, M, p.drop_p, seed, sm.ln32, xa, ldE, writer, (bf16_t*)L.du3b, L.part3,
                         smb.pstage, err, spin);
        }
        __syncthreads();
        if (!core) {
            // ================= B2 (workers): dhdn = gate(du3b W2) / (1 - p)
            if (wk < f_tiles) {
                const unsigned tag = ll_tag(epoch, l, 0);
#pragma unroll
                for (int i = 0; i < DEC_NT; ++i) {
                    const int tile = wk + i * NW;
                    if (tile < f_tiles) {
                        dec_tile<2>(wA[i], xa, ldE, M, tile * 16, sm, [&](int m, int n, f32x4 v) {
                            const bf16x4 gg = *reinterpret_cast<const bf16x4*>((const bf16_t*)L.hdn + m * F + n);
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] = ((float)gg[r] > 0.f) ? v[r] * p.gate_scale : 0.f;
                            const u32x2 pk = dec_pack4(v);
                            ll_put_bf16x4(ll, LL_HDN, tag, m * F + n, pk);
                            *reinterpret_cast<u32x2*>((bf16_t*)L.dhdn + m * F + n) = pk;
                        });
                    }
                }
            }
            // ================= B6 (workers): attention backward of (b, h) = worker id, + number of workers, ...
            if (wk < n_bh) {
                const uint32_t seed = drop ? rt_site_seed(p.seed_dev, L.seed_ad2) : 0u;
                DecKV kv;
                for (int bh = wk; bh < n_bh; bh += NW) {
                    dec_kv_prefetch(L, p.kpm, p.H, p.S, p.ldkv, bh, kv);
                    dec_attn_bwd(p, L, bh, kv, sm, seed, ll, ll_tag(epoch, l, 2), ll_tag(epoch, l, 3), err, spin);
                    __syncthreads();
                }
            }
            continue;
        }
        // ================= B3 (core): dt2 = du3 + dhdn W1
        u32x4 w2[2];
        ll_rows_to_lds<16>(ll, LL_HDN, ll_tag(epoch, l, 0), M, F, xa, ldF, err, spin);
        __syncthreads();
        {
            const unsigned tag = ll_tag(epoch, l, 1);
            dec_tile<16>(wf, xa, ldF, M, n0, sm, [&](int m, int n, f32x4 v) {
                v += *reinterpret_cast<const f32x4*>(&sm.ln32[m][n]);
                ll_put_f32x4(ll, LL_U, tag, m * DEC_E + n, v);
            });
        }
        // ================= B4: LayerNorm2 backward; B5: do2 = du2b Wo2
        dec_load_w<2>((const bf16_t*)L.WTo2, DEC_E, n0, w2);
        {
            const DecLnbSrc src{nullptr, nullptr, ll, LL_U, ll_tag(epoch, l, 1)};
            const uint32_t seed = drop ? rt_site_seed(p.seed_dev, L.seed_d2) : 0u;
            dec_lnb_rows(src, L.u2, L.mean2, L.rstd2, L.g2, M, p.drop_p, seed, smb.res2, xa, ldE, writer, (bf16_t*)L.du2b, L.part2,
                         smb.pstage, err, spin);
        }
        __syncthreads();
        {
            const unsigned tag = ll_tag(epoch, l, 2);
            dec_tile<2>(w2, xa, ldE, M, n0, sm, [&](int m, int n, f32x4 v) {
                ll_put_bf16x4(ll, LL_O2, tag, m * DEC_E + n, dec_pack4(v));
            });
        }
        // ================= B8: dt1q = dq2 Wq, accumulated into d query_pos
        dec_load_w<2>((const bf16_t*)L.WTq, DEC_E, n0, w2);
        ll_rows_to_lds<4>(ll, LL_Q2, ll_tag(epoch, l, 3), M, DEC_E, xa, ldE, err, spin);
        __syncthreads();
        {
            const unsigned tag = ll_tag(epoch, l, 4);
            dec_tile<2>(w2, xa, ldE, M, n0, sm, [&](int m, int n, f32x4 v) {
                ll_put_f32x4(ll, LL_U2, tag, m * DEC_E + n, v);
                f32x4* acc = reinterpret_cast<f32x4*>(p.dqpos + m * DEC_E + n);
                *acc += v;
            });
        }
        // ================= B9: LayerNorm1 backward of (du2 + dt1q); B10: dv = head-mask(dub Wo)
        dec_load_w<2>((const bf16_t*)L.WTo, DEC_E, n0, w2);
        {
            const DecLnbSrc src{nullptr, &smb.res2[0][0], ll, LL_U2, ll_tag(epoch, l, 4)};
            const uint32_t seed = drop ? rt_site_seed(p.seed_dev, L.seed_d1) : 0u;
            dec_lnb_rows(src, L.u, L.mean1, L.rstd1, L.g1, M, p.drop_p, seed, sm.ln32, xa, ldE, writer, (bf16_t*)L.dub, L.part1,
                         smb.pstage, err, spin);
        }
        __syncthreads();
        {
            const unsigned tag = ll_tag(epoch, l, 5);
            const uint32_t seed = drop ? rt_site_seed(p.seed_dev, L.seed_ad) : 0u;
            dec_tile<2>(w2, xa, ldE, M, n0, sm, [&](int m, int n, f32x4 v) {
                if (drop) v = dec_dropout(v, p.drop_p, seed, (uint32_t)(m * DEC_E + n), 5);
                const u32x2 pk = dec_pack4(v);
                ll_put_bf16x4(ll, LL_O, tag, m * DEC_E + n, pk);
                *reinterpret_cast<u32x2*>((bf16_t*)L.dv + m * DEC_E + n) = pk;
            });
        }
        // ================= B11: dta = du + dv Wv
        dec_load_w<2>((const bf16_t*)L.WTv, DEC_E, n0, w2);
        ll_rows_to_lds<4>(ll, LL_O, ll_tag(epoch, l, 5), M, DEC_E, xa, ldE, err, spin);
        __syncthreads();
        {
            const unsigned tag = ll_tag(epoch, l, 6);
            dec_tile<2>(w2, xa, ldE, M, n0, sm, [&](int m, int n, f32x4 v) {
                v += *reinterpret_cast<const f32x4*>(&sm.ln32[m][n]);
                if (l > 0) ll_put_f32x4(ll, LL_U3, tag, m * DEC_E + n, v);
                else *reinterpret_cast<f32x4*>(p.dta + m * DEC_E + n) = v;
            });
        }
    }
    // ---- the launch epoch moves on once nothing of this launch can still be read: workgroup 0 finished the first layer's B11 after
    // every hand-off it consumed; the other workgroups' last reads are of tags it has seen complete ...
    if (writer) {
        __syncthreads();
        if (t == 0) p.handoff[0] = epoch + 1;
    }
}

int g_spin_override = 0;              // rt_decoder_set_spin (tests force a hand-off timeout with 1)
int dec_spin() {
    static const int v = getenv("REFTR_DEC_SPIN") ? atoi(getenv("REFTR_DEC_SPIN")) : (1 << 20);
    return g_spin_override > 0 ? g_spin_override : v;
}
unsigned* dec_trace_buf() {           // REFTR_DEC_TRACE=1: 1024 words, read back by rt_decoder_trace
    static unsigned* buf = nullptr;
    static const int on = getenv("REFTR_DEC_TRACE") ? atoi(getenv("REFTR_DEC_TRACE")) : 0;
    static bool tried = false;
    if (on && !tried) {
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(hipStreamPerThread, &st);
        tried = true;
        if (hipMalloc(&buf, 4096) != hipSuccess) { buf = nullptr; (void)hipGetLastError(); } else (void)hipMemset(buf, 0, 4096);
    }
    return buf;
}
int dec_groups() {
    static const int g = getenv("REFTR_DEC_G") ? atoi(getenv("REFTR_DEC_G")) : 80;
    return g;
}
// Co-residency of the G spin-waiting workgroups is a REQUIREMENT of both kernels (a consumer polls a producer that must be running).
// They are launched with plain launches (a cooperative launch costs +17-20 us per replay and applies no check under graph replay
// anyway, MI355X_MICROARCH.md "Residency and cooperative launch"), so the check is made here, once per device: the occupancy query
// must admit at least one workgroup of each kernel per compute unit at its dynamic LDS size, and the device must have at least 2 G
// compute units (a margin of 2x against the query being one block per CU high and against other streams' resident workgroups:
// the BERT branch runs beside the decoder).  Anything else answers RT_ERR_UNSUPPORTED and the caller keeps the launched chain.
int dec_residency_ok(int F) {
    static int cached_dev = -1, cached_F = 0, cached = RT_ERR_UNSUPPORTED;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return RT_ERR_UNSUPPORTED;
    if (dev == cached_dev && F == cached_F) return cached;
    const int G = dec_groups();
    hipDeviceProp_t prop;
    int rc = RT_OK;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) rc = RT_ERR_UNSUPPORTED;
    else {
        const size_t smf = sizeof(DecSmem) + (size_t)16 * (F + 8) * 2, smb = sizeof(DecSmemB) + (size_t)16 * (F + 8) * 2;
        int nf = 0, nb = 0;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(decoder_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(decoder_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&nf, decoder_fwd_kernel, 256, smf) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, decoder_bwd_kernel, 256, smb) != hipSuccess) { (void)hipGetLastError(); rc = RT_ERR_UNSUPPORTED; }
        else if (nf < 1 || nb < 1 || prop.multiProcessorCount < 2 * G) rc = RT_ERR_UNSUPPORTED;
    }
    cached_dev = dev; cached_F = F; cached = rc;
    return rc;
}

}  // namespace

extern "C" int rt_decoder_fwd(const rt_decoder_fwd_desc* d, rt_stream_t stream) {
    if (!d || !d->t32 || !d->t16 || !d->qpos || !d->handoff) return RT_ERR_BADARG;
    if (d->n_layers < 1 || d->n_layers > RT_DEC_MAX_LAYERS) return RT_ERR_UNSUPPORTED;
    const int G = dec_groups();
    if (G <= DEC_CORE || G > 144) return RT_ERR_UNSUPPORTED;
    if (d->M < 1 || d->M > 16 || d->H * 32 != DEC_E || d->S < 1 || d->S > 256 * DEC_MAXK) return RT_ERR_UNSUPPORTED;
    if (d->F != 2048 || (d->F >> 4) > DEC_NT * (G - DEC_CORE) || (d->ldkv & 7)) return RT_ERR_UNSUPPORTED;
    for (int l = 0; l < d->n_layers; ++l) {
        const rt_decoder_layer_fwd& L = d->layer[l];
        if (!L.Wv || !L.Wo || !L.Wq || !L.Wo2 || !L.W1 || !L.W2 || !L.k2 || !L.v2 || !L.o || !L.u || !L.q2 || !L.o2 || !L.u2 ||
            !L.hdn || !L.u3 || !L.lse2 || !L.t1q16 || !L.t2_16 || !L.t3_16) return RT_ERR_BADARG;
    }
    if (dec_residency_ok(d->F) != RT_OK) return RT_ERR_UNSUPPORTED;
    const size_t smem = sizeof(DecSmem) + (size_t)16 * (d->F + 8) * 2;
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(decoder_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr = true;
    }
    hipLaunchKernelGGL(decoder_fwd_kernel, dim3(G), dim3(256), smem, (hipStream_t)stream, *d, G, dec_spin(), dec_trace_buf());
    RT_CHECK_LAUNCH();
    return RT_OK;
}

extern "C" int rt_decoder_bwd(const rt_decoder_bwd_desc* d, rt_stream_t stream) {
    if (!d || !d->dta || !d->dqpos || !d->handoff) return RT_ERR_BADARG;
    if (d->n_layers < 1 || d->n_layers > RT_DEC_MAX_LAYERS) return RT_ERR_UNSUPPORTED;
    const int G = dec_groups();
    if (G <= DEC_CORE || G > 144) return RT_ERR_UNSUPPORTED;
    if (d->M < 1 || d->M > 16 || d->H * 32 != DEC_E || d->S < 1 || d->S > 256 * DEC_MAXK) return RT_ERR_UNSUPPORTED;
    if (d->F != 2048 || (d->F >> 4) > DEC_NT * (G - DEC_CORE) || (d->ldkv & 7)) return RT_ERR_UNSUPPORTED;
    for (int l = 0; l < d->n_layers; ++l) {
        const rt_decoder_layer_bwd& L = d->layer[l];
        if (!L.WT2 || !L.WT1 || !L.WTo2 || !L.WTq || !L.WTo || !L.WTv || !L.g1 || !L.g2 || !L.g3 || !L.u || !L.u2 || !L.u3 || !L.mean1 ||
            !L.rstd1 || !L.mean2 || !L.rstd2 || !L.mean3 || !L.rstd3 || !L.hdn || !L.q2 || !L.k2 || !L.v2 || !L.o2 || !L.lse2 || !L.dnorm ||
            !L.du3b || !L.dhdn || !L.du2b || !L.dq2 || !L.dub || !L.dv || !L.dk2 || !L.dv2 || (!L.dk2p != !L.dv2p) ||
            (L.dk2p && (d->ldkvp < DEC_E || (d->ldkvp & 7)))) return RT_ERR_BADARG;
    }
    if (dec_residency_ok(d->F) != RT_OK) return RT_ERR_UNSUPPORTED;
    const size_t smem = sizeof(DecSmemB) + (size_t)16 * (d->F + 8) * 2;
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(decoder_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr = true;
    }
    hipLaunchKernelGGL(decoder_bwd_kernel, dim3(G), dim3(256), smem, (hipStream_t)stream, *d, G, dec_spin());
    RT_CHECK_LAUNCH();
    return RT_OK;
}

/* RT_OK when the cooperative decoder launches may be used on the current device (see dec_residency_ok), else RT_ERR_UNSUPPORTED. */
extern "C" int rt_decoder_supported(int F) {
    if (F != 2048) return RT_ERR_UNSUPPORTED;
    const int G = dec_groups();
    if (G <= DEC_CORE || G > 144 || (F >> 4) > DEC_NT * (G - DEC_CORE)) return RT_ERR_UNSUPPORTED;
    return dec_residency_ok(F);
}

/* Overrides the number of polls a consumer makes before it gives up and raises the failure word (<= 0: back to REFTR_DEC_SPIN /
   the default 2^20).  Tests use 1 to force a timeout. */
extern "C" int rt_decoder_set_spin(int spin) { g_spin_override = spin; return RT_OK; }

/* debugging aid (REFTR_DEC_TRACE=1): copies the 1024 stage time stamps of the last launch (100 MHz ticks; words 0.. = workgroup 0,
   words 512.. = the last workgroup) to `out`; returns RT_ERR_UNSUPPORTED when tracing is off. */
extern "C" int rt_decoder_trace(uint32_t* out) {
    unsigned* b = dec_trace_buf();
    if (!b) return RT_ERR_UNSUPPORTED;
    if (!out) return RT_OK;                  // allocation only (call once before any stream capture)
    if (hipDeviceSynchronize() != hipSuccess) return RT_ERR_BADARG;
    return (int)hipMemcpy(out, b, 4096, hipMemcpyDeviceToHost);
}
