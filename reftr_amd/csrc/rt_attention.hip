// Masked multi-head attention forward / backward on MFMA for the RefTR sequence lengths
// (VL encoder S = L + HW/32^2, dh = 32; BERT L <= 128, dh = 64; decoder cross-attention n_q*n_ph x S).
//
// One workgroup (8 waves) owns one (batch, head) and 128 rows of the "outer" axis, 16 per wave; the whole
// inner-axis operand pair of that head is staged once in LDS in its natural [row][dh] layout (row stride
// padded by 32 B, conflict-free for both access patterns below).  The score matrix never leaves registers:
//
//   forward / dQ : S^T tile [16 keys x 16 queries] = mfma(K rows, Q^T)  ("swapped" product): a lane then holds,
//     for ONE query (lane & 15), keys 4g..4g+3 of the tile (g = lane >> 4).  Two such tiles (32 keys) are exactly
//     the B-operand fragment of the next MFMA  O^T[d][q] += V^T[d][keys] P^T[keys][q]  — no cross-lane traffic for
//     P; V^T / K^T fragments come from the natural V / K rows through the gfx950 LDS transpose read
//     (ds_read_b64_tr_b16), with the same key -> k-slot assignment on both operands.
//     Softmax: two streaming passes over the keys (pass 1: running max / sum, pass 2: normalised P and PV), so
//     any S fits without holding S scores in registers; row statistics are combined across the 4 lane groups
//     with two shuffles.
//   dK / dV      : the roles swap — a wave owns 16 keys, S tile [16 queries x 16 keys] = mfma(Q rows, K^T), and
//     dV^T[d][key] += dO^T[d][q] P[q][key],  dK^T[d][key] += Q^T[d][q] dS[q][key]  with dO^T / Q^T via transpose reads.
//
// Dropout acts on the probabilities with the shared counter hash (index ((b*H+h)*Sq + q)*Sk + key); a fully
// masked row gives NaN like the reference's softmax over -inf.
#include "rt_common.h"
#include <stdlib.h>

namespace {

typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_ptr;

__device__ __forceinline__ bf16x8 tr_pair(const unsigned char* p0, const unsigned char* p1) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)p0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)p1);
    union { struct { s16x4 a, b; } s; bf16x8 v; } u;
    u.s.a = lo; u.s.b = hi;
    return u.v;
}

__device__ __forceinline__ bf16x8 pack8(const float (&v)[8]) {
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (bf16_t)v[e];
    return o;
}

template <int DH> struct Geo {
    // LDS row stride (bytes): + 32 B, conflict-free for both access patterns (a 16-B pad for dh = 32 would fit two workgroups per CU at
    // S = 440; measured identical -- the kernels are bound by the CU's VALU work, profiles/r03_side_stream_probes.txt)
    static constexpr int RS = DH * 2 + 32;
    static constexpr int KH = DH / 32;         // MFMA k-steps over the head dim
    static constexpr int DT = DH / 16;         // 16-wide output tiles over the head dim
};

// rows [0, rows) of a [*, ld] bf16 matrix (head slice) -> LDS rows of stride RS; rows [rows, rows_pad) zeroed
template <int DH>
__device__ __forceinline__ void stage_rows(unsigned char* dst, const bf16_t* src, int rows, int rows_pad, int ld, int tid, int nthreads) {
    constexpr int RS = Geo<DH>::RS, CPR = DH / 8, U = 4;
    // U independent 16-B loads in flight per thread before the first LDS store: the copy is one or two memory round trips, not
    // one per 16 B (a load -> store loop serialises on the load latency: 4-8 us for a 28 KB head slice)
    const int total = rows_pad * CPR;
    for (int c0 = tid; c0 < total; c0 += nthreads * U) {
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = c0 + u * nthreads;
            const int r = c / CPR, cc = c % CPR;
            v[u] = make_uint4(0, 0, 0, 0);
            if (c < total && r < rows) v[u] = *reinterpret_cast<const uint4*>(src + (size_t)r * ld + cc * 8);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = c0 + u * nthreads;
            const int r = c / CPR, cc = c % CPR;
            if (c < total) *reinterpret_cast<uint4*>(dst + r * RS + cc * 16) = v[u];
        }
    }
}

// two matrices at once (K and V, or Q and dO): both sets of loads are in flight together
template <int DH>
__device__ __forceinline__ void stage_rows2(unsigned char* dst0, const bf16_t* src0, int ld0, unsigned char* dst1, const bf16_t* src1, int ld1,
                                            int rows, int rows_pad, int tid, int nthreads) {
    constexpr int RS = Geo<DH>::RS, CPR = DH / 8, U = 4;
    const int total = rows_pad * CPR;
    for (int c0 = tid; c0 < total; c0 += nthreads * U) {
        uint4 a[U], b[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = c0 + u * nthreads;
            const int r = c / CPR, cc = c % CPR;
            a[u] = make_uint4(0, 0, 0, 0); b[u] = make_uint4(0, 0, 0, 0);
            if (c < total && r < rows) {
                a[u] = *reinterpret_cast<const uint4*>(src0 + (size_t)r * ld0 + cc * 8);
                b[u] = *reinterpret_cast<const uint4*>(src1 + (size_t)r * ld1 + cc * 8);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = c0 + u * nthreads;
            const int r = c / CPR, cc = c % CPR;
            if (c < total) {
                *reinterpret_cast<uint4*>(dst0 + r * RS + cc * 16) = a[u];
                *reinterpret_cast<uint4*>(dst1 + r * RS + cc * 16) = b[u];
            }
        }
    }
}

// fragment of a [rows][DH] global matrix used as MFMA B operand: lane (i, g) <- row (r0 + i), cols 32*kh + 8g..+7
template <int DH>
__device__ __forceinline__ void load_bfrag(const bf16_t* base, int row, int nrows, int ld, int lg, bf16x8 (&f)[Geo<DH>::KH]) {
#pragma unroll
    for (int kh = 0; kh < Geo<DH>::KH; ++kh) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (row < nrows) v = *reinterpret_cast<const uint4*>(base + (size_t)row * ld + kh * 32 + lg * 8);
        f[kh] = *reinterpret_cast<bf16x8*>(&v);
    }
}

// S tile = sum_kh mfma(A rows from LDS (row0 + li), B frag)
template <int DH>
__device__ __forceinline__ f32x4 tile_dot(const unsigned char* sA, int row0, int li, int lg, const bf16x8 (&bf)[Geo<DH>::KH]) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kh = 0; kh < Geo<DH>::KH; ++kh) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(sA + (row0 + li) * Geo<DH>::RS + kh * 64 + lg * 16);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bf[kh], acc, 0, 0, 0);
    }
    return acc;
}

// ------------------------------------------------------------------------------------------------ forward
template <int DH, int NW>
__global__ __launch_bounds__(64 * NW) void attn_fwd_kernel(const rt_attn_desc p) {
    constexpr int RS = Geo<DH>::RS, DT = Geo<DH>::DT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int Skp = (p.Sk + 31) & ~31;
    unsigned char* sK = smem;
    unsigned char* sV = sK + (size_t)Skp * RS;
    float* sBias = reinterpret_cast<float*>(sV + (size_t)Skp * RS);
    const int bh = blockIdx.x, b = bh / p.H, h = bh % p.H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, lg = lane >> 4;
    stage_rows2<DH>(sK, (const bf16_t*)p.k + (size_t)b * p.Sk * p.ldk + h * DH, p.ldk,
                    sV, (const bf16_t*)p.v + (size_t)b * p.Sk * p.ldv + h * DH, p.ldv, p.Sk, Skp, threadIdx.x, 64 * NW);
    for (int j = threadIdx.x; j < Skp; j += 64 * NW)
        sBias[j] = (j < p.Sk && !(p.kpm && p.kpm[(size_t)b * p.Sk + j])) ? 0.f : -INFINITY;
    __syncthreads();

    const int q = blockIdx.y * (16 * NW) + wave * 16 + li;          // this lane's query
    if (blockIdx.y * (16 * NW) + wave * 16 >= p.Sq) return;
    bf16x8 qf[Geo<DH>::KH];
    load_bfrag<DH>((const bf16_t*)p.q + (size_t)b * p.Sq * p.ldq + h * DH, q, p.Sq, p.ldq, lg, qf);
    const int nblk = Skp >> 4;

    // pass 1: running max / sum over this lane's keys (4 per 16-key tile)
    float m = -INFINITY, l = 0.f;
    for (int blk = 0; blk < nblk; ++blk) {
        const f32x4 acc = tile_dot<DH>(sK, blk * 16, li, lg, qf);
        const f32x4 bias = *reinterpret_cast<const f32x4*>(sBias + blk * 16 + lg * 4);
        float s[4], mx = m;
#pragma unroll
        for (int r = 0; r < 4; ++r) { s[r] = acc[r] * p.scale + bias[r]; mx = fmaxf(mx, s[r]); }
        const float ms = (mx == -INFINITY) ? 0.f : mx;
        float add = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) add += __expf(s[r] - ms);
        l = l * __expf(m - ms) + add;
        m = mx;
    }
    float M = fmaxf(m, __shfl_xor(m, 16, 64));
    M = fmaxf(M, __shfl_xor(M, 32, 64));
    const float Ms = (M == -INFINITY) ? 0.f : M;
    l *= __expf(m - Ms);
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv_l = 1.f / l;                 // fully masked row: 0 * inf = NaN below, as the reference

    // pass 2: normalised probabilities -> P^T fragments -> O^T += V^T P^T
    const bool do_drop = p.drop_p > 0.f;
    const uint32_t thresh = rt_drop_thresh(p.drop_p);
    const float ks = do_drop ? 1.f / (1.f - p.drop_p) : 1.f;
    const uint32_t dseed = rt_site_seed(p.seed_dev, p.drop_seed);
    const uint32_t drop_row = (uint32_t)(((size_t)bh * p.Sq + q) * p.Sk);
    f32x4 o[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t) o[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int tr_r = 4 * lg + (li >> 2), tr_c = (li & 3) * 8;
    for (int c = 0; c < (Skp >> 5); ++c) {
        float pv[8];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int k0 = c * 32 + half * 16;
            const f32x4 acc = tile_dot<DH>(sK, k0, li, lg, qf);
            const f32x4 bias = *reinterpret_cast<const f32x4*>(sBias + k0 + lg * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float pr = __expf(acc[r] * p.scale + bias[r] - Ms) * inv_l;
                if (do_drop) pr = (rt_hash32(dseed, drop_row + (uint32_t)(k0 + lg * 4 + r)) >= thresh) ? pr * ks : 0.f;
                pv[half * 4 + r] = pr;
            }
        }
        const bf16x8 pf = pack8(pv);
        const unsigned char* v0 = sV + (c * 32 + tr_r) * RS + tr_c;
#pragma unroll
        for (int t = 0; t < DT; ++t) {
            const bf16x8 vf = tr_pair(v0 + t * 32, v0 + 16 * RS + t * 32);
            o[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, o[t], 0, 0, 0);
        }
    }
    if (q < p.Sq) {
        bf16_t* orow = (bf16_t*)p.out + ((size_t)b * p.Sq + q) * p.ldo + h * DH;
#pragma unroll
        for (int t = 0; t < DT; ++t) {
            bf16x4 ov;
#pragma unroll
            for (int r = 0; r < 4; ++r) ov[r] = (bf16_t)o[t][r];
            *reinterpret_cast<bf16x4*>(orow + t * 16 + lg * 4) = ov;
        }
        if (lg == 0 && p.lse) p.lse[(size_t)bh * p.Sq + q] = M + __logf(l);
    }
}

// Forward with the score row kept in registers (NT 16-key tiles, fully unrolled): the NT score MFMAs of a wave are
// independent and issue back to back, the softmax runs over registers, and the scores are not recomputed for the P V pass.
// The two-pass kernel above walks 2 x NT dependent [LDS read -> MFMA -> exp] steps per wave and is latency-bound at one
// workgroup per CU (88 KB of LDS for S = 440).
template <int DH, int NW, int NT>
__global__ __launch_bounds__(64 * NW) void attn_fwd_reg_kernel(const rt_attn_desc p) {
    constexpr int RS = Geo<DH>::RS, DT = Geo<DH>::DT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int Skp = (p.Sk + 31) & ~31;
    unsigned char* sK = smem;
    unsigned char* sV = sK + (size_t)Skp * RS;
    float* sBias = reinterpret_cast<float*>(sV + (size_t)Skp * RS);
    const int bh = blockIdx.x, b = bh / p.H, h = bh % p.H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, lg = lane >> 4;
    stage_rows2<DH>(sK, (const bf16_t*)p.k + (size_t)b * p.Sk * p.ldk + h * DH, p.ldk,
                    sV, (const bf16_t*)p.v + (size_t)b * p.Sk * p.ldv + h * DH, p.ldv, p.Sk, Skp, threadIdx.x, 64 * NW);
    for (int j = threadIdx.x; j < Skp; j += 64 * NW)
        sBias[j] = (j < p.Sk && !(p.kpm && p.kpm[(size_t)b * p.Sk + j])) ? 0.f : -INFINITY;
    __syncthreads();

    const int q = blockIdx.y * (16 * NW) + wave * 16 + li;
    if (blockIdx.y * (16 * NW) + wave * 16 >= p.Sq) return;
    bf16x8 qf[Geo<DH>::KH];
    load_bfrag<DH>((const bf16_t*)p.q + (size_t)b * p.Sq * p.ldq + h * DH, q, p.Sq, p.ldq, lg, qf);
    const int nblk = Skp >> 4;                  // <= NT, even

    f32x4 sc[NT];
    float m = -INFINITY;
#pragma unroll
    for (int blk = 0; blk < NT; ++blk) {
        if (blk < nblk) {
            const f32x4 acc = tile_dot<DH>(sK, blk * 16, li, lg, qf);
            const f32x4 bias = *reinterpret_cast<const f32x4*>(sBias + blk * 16 + lg * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) { sc[blk][r] = acc[r] * p.scale + bias[r]; m = fmaxf(m, sc[blk][r]); }
        } else {
            sc[blk] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        }
    }
    float M = fmaxf(m, __shfl_xor(m, 16, 64));
    M = fmaxf(M, __shfl_xor(M, 32, 64));
    const float Ms = (M == -INFINITY) ? 0.f : M;
    float l = 0.f;
#pragma unroll
    for (int blk = 0; blk < NT; ++blk)
#pragma unroll
        for (int r = 0; r < 4; ++r) { sc[blk][r] = __expf(sc[blk][r] - Ms); l += sc[blk][r]; }
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv_l = 1.f / l;                 // fully masked row: 0 * inf = NaN below, as the reference

    const bool do_drop = p.drop_p > 0.f;
    const uint32_t thresh = rt_drop_thresh(p.drop_p);
    const float ks = do_drop ? 1.f / (1.f - p.drop_p) : 1.f;
    const uint32_t dseed = rt_site_seed(p.seed_dev, p.drop_seed);
    const uint32_t drop_row = (uint32_t)(((size_t)bh * p.Sq + q) * p.Sk);
    f32x4 o[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t) o[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int tr_r = 4 * lg + (li >> 2), tr_c = (li & 3) * 8;
#pragma unroll
    for (int c = 0; c < NT / 2; ++c) {
        if (2 * c < nblk) {
            float pv[8];
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int k0 = c * 32 + half * 16;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float pr = sc[2 * c + half][r] * inv_l;
                    if (do_drop) pr = (rt_hash32(dseed, drop_row + (uint32_t)(k0 + lg * 4 + r)) >= thresh) ? pr * ks : 0.f;
                    pv[half * 4 + r] = pr;
                }
            }
            const bf16x8 pf = pack8(pv);
            const unsigned char* v0 = sV + (c * 32 + tr_r) * RS + tr_c;
#pragma unroll
            for (int t = 0; t < DT; ++t) {
                const bf16x8 vf = tr_pair(v0 + t * 32, v0 + 16 * RS + t * 32);
                o[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, o[t], 0, 0, 0);
            }
        }
    }
    if (q < p.Sq) {
        bf16_t* orow = (bf16_t*)p.out + ((size_t)b * p.Sq + q) * p.ldo + h * DH;
#pragma unroll
        for (int t = 0; t < DT; ++t) {
            bf16x4 ov;
#pragma unroll
            for (int r = 0; r < 4; ++r) ov[r] = (bf16_t)o[t][r];
            *reinterpret_cast<bf16x4*>(orow + t * 16 + lg * 4) = ov;
        }
        if (lg == 0 && p.lse) p.lse[(size_t)bh * p.Sq + q] = M + __logf(l);
    }
}

// ------------------------------------------------------------------------------------------------ dQ (+ delta)
template <int DH, int NW>
__device__ __forceinline__ void attn_bwd_dq_body(const rt_attn_bwd_desc& p, const int by) {
    constexpr int RS = Geo<DH>::RS, DT = Geo<DH>::DT, KH = Geo<DH>::KH;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int Skp = (p.Sk + 31) & ~31;
    unsigned char* sK = smem;
    unsigned char* sV = sK + (size_t)Skp * RS;
    float* sBias = reinterpret_cast<float*>(sV + (size_t)Skp * RS);
    const int bh = blockIdx.x, b = bh / p.H, h = bh % p.H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, lg = lane >> 4;
    stage_rows2<DH>(sK, (const bf16_t*)p.k + (size_t)b * p.Sk * p.ldk + h * DH, p.ldk,
                    sV, (const bf16_t*)p.v + (size_t)b * p.Sk * p.ldv + h * DH, p.ldv, p.Sk, Skp, threadIdx.x, 64 * NW);
    for (int j = threadIdx.x; j < Skp; j += 64 * NW)
        sBias[j] = (j < p.Sk && !(p.kpm && p.kpm[(size_t)b * p.Sk + j])) ? 0.f : -INFINITY;
    __syncthreads();

    const int q = by * (16 * NW) + wave * 16 + li;
    if (by * (16 * NW) + wave * 16 >= p.Sq) return;
    bf16x8 qf[KH], dof[KH], of[KH];
    load_bfrag<DH>((const bf16_t*)p.q + (size_t)b * p.Sq * p.ldq + h * DH, q, p.Sq, p.ldq, lg, qf);
    load_bfrag<DH>((const bf16_t*)p.dout + (size_t)b * p.Sq * p.ldo + h * DH, q, p.Sq, p.ldo, lg, dof);
    load_bfrag<DH>((const bf16_t*)p.out + (size_t)b * p.Sq * p.ldo + h * DH, q, p.Sq, p.ldo, lg, of);
    float delta = 0.f;
#pragma unroll
    for (int kh = 0; kh < KH; ++kh)
#pragma unroll
        for (int e = 0; e < 8; ++e) delta += (float)dof[kh][e] * (float)of[kh][e];
    delta += __shfl_xor(delta, 16, 64);
    delta += __shfl_xor(delta, 32, 64);
    const float lse = (q < p.Sq) ? p.lse[(size_t)bh * p.Sq + q] : INFINITY;
    if (lg == 0 && q < p.Sq) p.delta[(size_t)bh * p.Sq + q] = delta;

    const bool do_drop = p.drop_p > 0.f;
    const uint32_t thresh = rt_drop_thresh(p.drop_p);
    const float ks = do_drop ? 1.f / (1.f - p.drop_p) : 1.f;
    const uint32_t dseed = rt_site_seed(p.seed_dev, p.drop_seed);
    const uint32_t drop_row = (uint32_t)(((size_t)bh * p.Sq + q) * p.Sk);
    f32x4 dq[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t) dq[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int tr_r = 4 * lg + (li >> 2), tr_c = (li & 3) * 8;
    for (int c = 0; c < (Skp >> 5); ++c) {
        float dsv[8];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int k0 = c * 32 + half * 16;
            const f32x4 s = tile_dot<DH>(sK, k0, li, lg, qf);
            const f32x4 dp = tile_dot<DH>(sV, k0, li, lg, dof);
            const f32x4 bias = *reinterpret_cast<const f32x4*>(sBias + k0 + lg * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pr = __expf(s[r] * p.scale + bias[r] - lse);
                float d = dp[r];
                if (do_drop) d = (rt_hash32(dseed, drop_row + (uint32_t)(k0 + lg * 4 + r)) >= thresh) ? d * ks : 0.f;
                dsv[half * 4 + r] = pr * (d - delta) * p.scale;
            }
        }
        const bf16x8 dsf = pack8(dsv);
        const unsigned char* k0p = sK + (c * 32 + tr_r) * RS + tr_c;
#pragma unroll
        for (int t = 0; t < DT; ++t) {
            const bf16x8 kf = tr_pair(k0p + t * 32, k0p + 16 * RS + t * 32);
            dq[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, dsf, dq[t], 0, 0, 0);
        }
    }
    if (q < p.Sq) {
        bf16_t* drow = (bf16_t*)p.dq + ((size_t)b * p.Sq + q) * p.lddq + h * DH;
#pragma unroll
        for (int t = 0; t < DT; ++t) {
            bf16x4 ov;
#pragma unroll
            for (int r = 0; r < 4; ++r) ov[r] = (bf16_t)dq[t][r];
            *reinterpret_cast<bf16x4*>(drow + t * 16 + lg * 4) = ov;
        }
    }
}

// ------------------------------------------------------------------------------------------------ dK, dV
// OWN_DELTA: delta[i] = sum_d dO[i, d] * O[i, d] is recomputed here for the head's query rows (64-128 B of O and dO per row)
// instead of being read from the dQ kernel's output: the two halves of the backward then do not depend on each other and run
// as ONE launch (attn_bwd_fused_kernel) -- on the encoder's chain that is a 17 us kernel and a graph-node boundary less per layer.
template <int DH, int NW, bool OWN_DELTA>
__device__ __forceinline__ void attn_bwd_dkv_body(const rt_attn_bwd_desc& p, const int by) {
    constexpr int RS = Geo<DH>::RS, DT = Geo<DH>::DT, KH = Geo<DH>::KH;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int Sqp = (p.Sq + 31) & ~31;
    unsigned char* sQ = smem;
    unsigned char* sD = sQ + (size_t)Sqp * RS;
    float* sL = reinterpret_cast<float*>(sD + (size_t)Sqp * RS);
    float* sDel = sL + Sqp;
    const int bh = blockIdx.x, b = bh / p.H, h = bh % p.H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, lg = lane >> 4;
    stage_rows2<DH>(sQ, (const bf16_t*)p.q + (size_t)b * p.Sq * p.ldq + h * DH, p.ldq,
                    sD, (const bf16_t*)p.dout + (size_t)b * p.Sq * p.ldo + h * DH, p.ldo, p.Sq, Sqp, threadIdx.x, 64 * NW);
    for (int i = threadIdx.x; i < Sqp; i += 64 * NW) {
        sL[i] = (i < p.Sq) ? p.lse[(size_t)bh * p.Sq + i] : INFINITY;       // padded query rows: p = exp(-inf) = 0
        float del = 0.f;
        if (i < p.Sq) {
            if (OWN_DELTA) {
                const bf16_t* orow = (const bf16_t*)p.out + ((size_t)b * p.Sq + i) * p.ldo + h * DH;
                const bf16_t* drow = (const bf16_t*)p.dout + ((size_t)b * p.Sq + i) * p.ldo + h * DH;
#pragma unroll
                for (int c = 0; c < DH; c += 8) {
                    const bf16x8 ov = *reinterpret_cast<const bf16x8*>(orow + c), dv8 = *reinterpret_cast<const bf16x8*>(drow + c);
#pragma unroll
                    for (int e = 0; e < 8; ++e) del += (float)dv8[e] * (float)ov[e];
                }
            } else {
                del = p.delta[(size_t)bh * p.Sq + i];
            }
        }
        sDel[i] = del;
    }
    __syncthreads();

    const int key = by * (16 * NW) + wave * 16 + li;         // this lane's key (MFMA column)
    if (by * (16 * NW) + wave * 16 >= p.Sk) return;
    bf16x8 kf[KH], vf[KH];
    load_bfrag<DH>((const bf16_t*)p.k + (size_t)b * p.Sk * p.ldk + h * DH, key, p.Sk, p.ldk, lg, kf);
    load_bfrag<DH>((const bf16_t*)p.v + (size_t)b * p.Sk * p.ldv + h * DH, key, p.Sk, p.ldv, lg, vf);
    const float kbias = (key < p.Sk && !(p.kpm && p.kpm[(size_t)b * p.Sk + key])) ? 0.f : -INFINITY;

    const bool do_drop = p.drop_p > 0.f;
    const uint32_t thresh = rt_drop_thresh(p.drop_p);
    const float ks = do_drop ? 1.f / (1.f - p.drop_p) : 1.f;
    const uint32_t dseed = rt_site_seed(p.seed_dev, p.drop_seed);
    f32x4 dk[DT], dv[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t) { dk[t] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    const int tr_r = 4 * lg + (li >> 2), tr_c = (li & 3) * 8;
    for (int c = 0; c < (Sqp >> 5); ++c) {
        float pv[8], dsv[8];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int q0 = c * 32 + half * 16;
            const f32x4 s = tile_dot<DH>(sQ, q0, li, lg, kf);          // [query 4g+r][key li]
            const f32x4 dp = tile_dot<DH>(sD, q0, li, lg, vf);
            const f32x4 lse = *reinterpret_cast<const f32x4*>(sL + q0 + lg * 4);
            const f32x4 del = *reinterpret_cast<const f32x4*>(sDel + q0 + lg * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pr = __expf(s[r] * p.scale + kbias - lse[r]);
                float d = dp[r], pd = pr;
                if (do_drop) {
                    const int qq = q0 + lg * 4 + r;
                    const bool keep = rt_hash32(dseed, (uint32_t)(((size_t)bh * p.Sq + qq) * p.Sk + key)) >= thresh;
                    d = keep ? d * ks : 0.f; pd = keep ? pr * ks : 0.f;
                }
                pv[half * 4 + r] = pd;
                dsv[half * 4 + r] = pr * (d - del[r]) * p.scale;
            }
        }
        const bf16x8 pf = pack8(pv), dsf = pack8(dsv);
        const unsigned char* d0 = sD + (c * 32 + tr_r) * RS + tr_c;
        const unsigned char* q0p = sQ + (c * 32 + tr_r) * RS + tr_c;
#pragma unroll
        for (int t = 0; t < DT; ++t) {
            const bf16x8 dof = tr_pair(d0 + t * 32, d0 + 16 * RS + t * 32);
            dv[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dof, pf, dv[t], 0, 0, 0);
            const bf16x8 qtf = tr_pair(q0p + t * 32, q0p + 16 * RS + t * 32);
            dk[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qtf, dsf, dk[t], 0, 0, 0);
        }
    }
    if (key < p.Sk) {
        bf16_t* kro = (bf16_t*)p.dk + ((size_t)b * p.Sk + key) * p.lddk + h * DH;
        bf16_t* vro = (bf16_t*)p.dv + ((size_t)b * p.Sk + key) * p.lddv + h * DH;
#pragma unroll
        for (int t = 0; t < DT; ++t) {
            bf16x4 a, c2;
#pragma unroll
            for (int r = 0; r < 4; ++r) { a[r] = (bf16_t)dk[t][r]; c2[r] = (bf16_t)dv[t][r]; }
            *reinterpret_cast<bf16x4*>(kro + t * 16 + lg * 4) = a;
            *reinterpret_cast<bf16x4*>(vro + t * 16 + lg * 4) = c2;
        }
    }
}

template <int DH, int NW>
__global__ __launch_bounds__(64 * NW) void attn_bwd_dq_kernel(const rt_attn_bwd_desc p) {
    attn_bwd_dq_body<DH, NW>(p, (int)blockIdx.y);
}
template <int DH, int NW>
__global__ __launch_bounds__(64 * NW) void attn_bwd_dkv_kernel(const rt_attn_bwd_desc p) {
    attn_bwd_dkv_body<DH, NW, false>(p, (int)blockIdx.y);
}
// both halves in one launch: blockIdx.y < ny_dq -> the dQ row blocks, the rest -> the dK / dV key blocks
template <int DH, int NW>
__global__ __launch_bounds__(64 * NW) void attn_bwd_fused_kernel(const rt_attn_bwd_desc p, const int ny_dq) {
    if ((int)blockIdx.y < ny_dq) attn_bwd_dq_body<DH, NW>(p, (int)blockIdx.y);
    else attn_bwd_dkv_body<DH, NW, true>(p, (int)blockIdx.y - ny_dq);
}

// ------------------------------------------------------------------------------------------------ long inner axes
// Sequences whose K / V (forward, dQ) or Q / dO (dK, dV) rows do not fit the CU's 160 KB at once (--dilation at 640 x 640: c5 at
// stride 16, S = 1600 + L): the inner axis is walked in chunks of LONG_CH rows, each staged into the same LDS region behind a
// barrier.  Same lane <-> (query, key) assignment and the same key order per lane as the whole-axis kernels above -- the forward
// is the two-pass kernel with each pass running over the chunks (pass 1 stages K only) -- so the results are bit-identical to
// attn_fwd_kernel / the dq / dkv bodies wherever both apply (tests force this path on short sequences with a small chunk).
template <int DH> struct LongGeo { static constexpr int CH = DH == 32 ? 768 : 448; };     // 2 * CH * RS + 8 * CH <= 160 KB

template <int DH, int NW>
__global__ __launch_bounds__(64 * NW) void attn_fwd_long_kernel(const rt_attn_desc p, const int ch) {
    constexpr int RS = Geo<DH>::RS, DT = Geo<DH>::DT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int Skp = (p.Sk + 31) & ~31;
    unsigned char* sK = smem;
    unsigned char* sV = sK + (size_t)ch * RS;
    float* sBias = reinterpret_cast<float*>(sV + (size_t)ch * RS);
    const int bh = blockIdx.x, b = bh / p.H, h = bh % p.H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const bf16_t* kbase = (const bf16_t*)p.k + (size_t)b * p.Sk * p.ldk + h * DH;
    const bf16_t* vbase = (const bf16_t*)p.v + (size_t)b * p.Sk * p.ldv + h * DH;
    const int q = blockIdx.y * (16 * NW) + wave * 16 + li;          // this lane's query (rows past Sq: zero fragments, never stored)
    bf16x8 qf[Geo<DH>::KH];
    load_bfrag<DH>((const bf16_t*)p.q + (size_t)b * p.Sq * p.ldq + h * DH, q, p.Sq, p.ldq, lg, qf);

    float m = -INFINITY, l = 0.f;
    for (int c0 = 0; c0 < Skp; c0 += ch) {
        const int rows = (Skp - c0 < ch) ? Skp - c0 : ch;
        const int valid = (p.Sk - c0 < rows) ? p.Sk - c0 : rows;
        __syncthreads();
        stage_rows<DH>(sK, kbase + (size_t)c0 * p.ldk, valid, rows, p.ldk, threadIdx.x, 64 * NW);
        for (int j = threadIdx.x; j < rows; j += 64 * NW)
            sBias[j] = (c0 + j < p.Sk && !(p.kpm && p.kpm[(size_t)b * p.Sk + c0 + j])) ? 0.f : -INFINITY;
        __syncthreads();
        for (int blk = 0; blk < (rows >> 4); ++blk) {
            const f32x4 acc = tile_dot<DH>(sK, blk * 16, li, lg, qf);
            const f32x4 bias = *reinterpret_cast<const f32x4*>(sBias + blk * 16 + lg * 4);
            float sc[4], mx = m;
#pragma unroll
            for (int r = 0; r < 4; ++r) { sc[r] = acc[r] * p.scale + bias[r]; mx = fmaxf(mx, sc[r]); }
            const float ms = (mx == -INFINITY) ? 0.f : mx;
            float add = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) add += __expf(sc[r] - ms);
            l = l * __expf(m - ms) + add;
            m = mx;
        }
    }
    float M = fmaxf(m, __shfl_xor(m, 16, 64));
    M = fmaxf(M, __shfl_xor(M, 32, 64));
    const float Ms = (M == -INFINITY) ? 0.f : M;
    l *= __expf(m - Ms);
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv_l = 1.f / l;

    const bool do_drop = p.drop_p > 0.f;
    const uint32_t thresh = rt_drop_thresh(p.drop_p);
    const float ks = do_drop ? 1.f / (1.f - p.drop_p) : 1.f;
    const uint32_t dseed = rt_site_seed(p.seed_dev, p.drop_seed);
    const uint32_t drop_row = (uint32_t)(((size_t)bh * p.Sq + q) * p.Sk);
    f32x4 o[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t) o[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int tr_r = 4 * lg + (li >> 2), tr_c = (li & 3) * 8;
    for (int c0 = 0; c0 < Skp; c0 += ch) {
        const int rows = (Skp - c0 < ch) ? Skp - c0 : ch;
        const int valid = (p.Sk - c0 < rows) ? p.Sk - c0 : rows;
        __syncthreads();
        stage_rows2<DH>(sK, kbase + (size_t)c0 * p.ldk, p.ldk, sV, vbase + (size_t)c0 * p.ldv, p.ldv, valid, rows, threadIdx.x, 64 * NW);
        for (int j = threadIdx.x; j < rows; j += 64 * NW)
            sBias[j] = (c0 + j < p.Sk && !(p.kpm && p.kpm[(size_t)b * p.Sk + c0 + j])) ? 0.f : -INFINITY;
        __syncthreads();
        for (int c = 0; c < (rows >> 5); ++c) {
            float pv[8];
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int k0 = c * 32 + half * 16;
                const f32x4 acc = tile_dot<DH>(sK, k0, li, lg, qf);
                const f32x4 bias = *reinterpret_cast<const f32x4*>(sBias + k0 + lg * 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float pr = __expf(acc[r] * p.scale + bias[r] - Ms) * inv_l;
                    if (do_drop) pr = (rt_hash32(dseed, drop_row + (uint32_t)(c0 + k0 + lg * 4 + r)) >= thresh) ? pr * ks : 0.f;
                    pv[half * 4 + r] = pr;
                }
            }
            const bf16x8 pf = pack8(pv);
            const unsigned char* v0 = sV + (c * 32 + tr_r) * RS + tr_c;
#pragma unroll
            for (int t = 0; t < DT; ++t) {
                const bf16x8 vf = tr_pair(v0 + t * 32, v0 + 16 * RS + t * 32);
                o[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, o[t], 0, 0, 0);
            }
        }
    }
    if (q < p.Sq) {
        bf16_t* orow = (bf16_t*)p.out + ((size_t)b * p.Sq + q) * p.ldo + h * DH;
#pragma unroll
        for (int t = 0; t < DT; ++t) {
            bf16x4 ov;
#pragma unroll
            for (int r = 0; r < 4; ++r) ov[r] = (bf16_t)o[t][r];
            *reinterpret_cast<bf16x4*>(orow + t * 16 + lg * 4) = ov;
        }
        if (lg == 0 && p.lse) p.lse[(size_t)bh * p.Sq + q] = M + __logf(l);
    }
}

template <int DH, int NW>
__device__ __forceinline__ void attn_bwd_dq_long_body(const rt_attn_bwd_desc& p, const int by, const int ch) {
    constexpr int RS = Geo<DH>::RS, DT = Geo<DH>::DT, KH = Geo<DH>::KH;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int Skp = (p.Sk + 31) & ~31;
    unsigned char* sK = smem;
    unsigned char* sV = sK + (size_t)ch * RS;
    float* sBias = reinterpret_cast<float*>(sV + (size_t)ch * RS);
    const int bh = blockIdx.x, b = bh / p.H, h = bh % p.H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const bf16_t* kbase = (const bf16_t*)p.k + (size_t)b * p.Sk * p.ldk + h * DH;
    const bf16_t* vbase = (const bf16_t*)p.v + (size_t)b * p.Sk * p.ldv + h * DH;
    const int q = by * (16 * NW) + wave * 16 + li;
    bf16x8 qf[KH], dof[KH], of[KH];
    load_bfrag<DH>((const bf16_t*)p.q + (size_t)b * p.Sq * p.ldq + h * DH, q, p.Sq, p.ldq, lg, qf);
    load_bfrag<DH>((const bf16_t*)p.dout + (size_t)b * p.Sq * p.ldo + h * DH, q, p.Sq, p.ldo, lg, dof);
    load_bfrag<DH>((const bf16_t*)p.out + (size_t)b * p.Sq * p.ldo + h * DH, q, p.Sq, p.ldo, lg, of);
    float delta = 0.f;
#pragma unroll
    for (int kh = 0; kh < KH; ++kh)
#pragma unroll
        for (int e = 0; e < 8; ++e) delta += (float)dof[kh][e] * (float)of[kh][e];
    delta += __shfl_xor(delta, 16, 64);
    delta += __shfl_xor(delta, 32, 64);
    const float lse = (q < p.Sq) ? p.lse[(size_t)bh * p.Sq + q] : INFINITY;
    if (lg == 0 && q < p.Sq) p.delta[(size_t)bh * p.Sq + q] = delta;

    const bool do_drop = p.drop_p > 0.f;
    const uint32_t thresh = rt_drop_thresh(p.drop_p);
    const float ks = do_drop ? 1.f / (1.f - p.drop_p) : 1.f;
    const uint32_t dseed = rt_site_seed(p.seed_dev, p.drop_seed);
    const uint32_t drop_row = (uint32_t)(((size_t)bh * p.Sq + q) * p.Sk);
    f32x4 dq[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t) dq[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int tr_r = 4 * lg + (li >> 2), tr_c = (li & 3) * 8;
    for (int c0 = 0; c0 < Skp; c0 += ch) {
        const int rows = (Skp - c0 < ch) ? Skp - c0 : ch;
        const int valid = (p.Sk - c0 < rows) ? p.Sk - c0 : rows;
        __syncthreads();
        stage_rows2<DH>(sK, kbase + (size_t)c0 * p.ldk, p.ldk, sV, vbase + (size_t)c0 * p.ldv, p.ldv, valid, rows, threadIdx.x, 64 * NW);
        for (int j = threadIdx.x; j < rows; j += 64 * NW)
            sBias[j] = (c0 + j < p.Sk && !(p.kpm && p.kpm[(size_t)b * p.Sk + c0 + j])) ? 0.f : -INFINITY;
        __syncthreads();
        for (int c = 0; c < (rows >> 5); ++c) {
            float dsv[8];
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int k0 = c * 32 + half * 16;
                const f32x4 sc = tile_dot<DH>(sK, k0, li, lg, qf);
                const f32x4 dp = tile_dot<DH>(sV, k0, li, lg, dof);
                const f32x4 bias = *reinterpret_cast<const f32x4*>(sBias + k0 + lg * 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pr = __expf(sc[r] * p.scale + bias[r] - lse);
                    float d = dp[r];
                    if (do_drop) d = (rt_hash32(dseed, drop_row + (uint32_t)(c0 + k0 + lg * 4 + r)) >= thresh) ? d * ks : 0.f;
                    dsv[half * 4 + r] = pr * (d - delta) * p.scale;
                }
            }
            const bf16x8 dsf = pack8(dsv);
            const unsigned char* k0p = sK + (c * 32 + tr_r) * RS + tr_c;
#pragma unroll
            for (int t = 0; t < DT; ++t) {
                const bf16x8 kf = tr_pair(k0p + t * 32, k0p + 16 * RS + t * 32);
                dq[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, dsf, dq[t], 0, 0, 0);
            }
        }
    }
    if (q < p.Sq) {
        bf16_t* drow = (bf16_t*)p.dq + ((size_t)b * p.Sq + q) * p.lddq + h * DH;
#pragma unroll
        for (int t = 0; t < DT; ++t) {
            bf16x4 ov;
#pragma unroll
            for (int r = 0; r < 4; ++r) ov[r] = (bf16_t)dq[t][r];
            *reinterpret_cast<bf16x4*>(drow + t * 16 + lg * 4) = ov;
        }
    }
}

template <int DH, int NW>
__device__ __forceinline__ void attn_bwd_dkv_long_body(const rt_attn_bwd_desc& p, const int by, const int ch) {
    constexpr int RS = Geo<DH>::RS, DT = Geo<DH>::DT, KH = Geo<DH>::KH;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int Sqp = (p.Sq + 31) & ~31;
    unsigned char* sQ = smem;
    unsigned char* sD = sQ + (size_t)ch * RS;
    float* sL = reinterpret_cast<float*>(sD + (size_t)ch * RS);
    float* sDel = sL + ch;
    const int bh = blockIdx.x, b = bh / p.H, h = bh % p.H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const bf16_t* qbase = (const bf16_t*)p.q + (size_t)b * p.Sq * p.ldq + h * DH;
    const bf16_t* dbase = (const bf16_t*)p.dout + (size_t)b * p.Sq * p.ldo + h * DH;
    const int key = by * (16 * NW) + wave * 16 + li;
    bf16x8 kf[KH], vf[KH];
    load_bfrag<DH>((const bf16_t*)p.k + (size_t)b * p.Sk * p.ldk + h * DH, key, p.Sk, p.ldk, lg, kf);
    load_bfrag<DH>((const bf16_t*)p.v + (size_t)b * p.Sk * p.ldv + h * DH, key, p.Sk, p.ldv, lg, vf);
    const float kbias = (key < p.Sk && !(p.kpm && p.kpm[(size_t)b * p.Sk + key])) ? 0.f : -INFINITY;
    const bool do_drop = p.drop_p > 0.f;
    const uint32_t thresh = rt_drop_thresh(p.drop_p);
    const float ks = do_drop ? 1.f / (1.f - p.drop_p) : 1.f;
    const uint32_t dseed = rt_site_seed(p.seed_dev, p.drop_seed);
    f32x4 dk[DT], dv[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t) { dk[t] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    const int tr_r = 4 * lg + (li >> 2), tr_c = (li & 3) * 8;
    for (int c0 = 0; c0 < Sqp; c0 += ch) {
        const int rows = (Sqp - c0 < ch) ? Sqp - c0 : ch;
        const int valid = (p.Sq - c0 < rows) ? p.Sq - c0 : rows;
        __syncthreads();
        stage_rows2<DH>(sQ, qbase + (size_t)c0 * p.ldq, p.ldq, sD, dbase + (size_t)c0 * p.ldo, p.ldo, valid, rows, threadIdx.x, 64 * NW);
        for (int i = threadIdx.x; i < rows; i += 64 * NW) {
            const int qi = c0 + i;
            sL[i] = (qi < p.Sq) ? p.lse[(size_t)bh * p.Sq + qi] : INFINITY;
            float del = 0.f;
            if (qi < p.Sq) {                       // delta recomputed from the head's O / dO rows: independent of the dQ blocks
                const bf16_t* orow = (const bf16_t*)p.out + ((size_t)b * p.Sq + qi) * p.ldo + h * DH;
                const bf16_t* drow = dbase + (size_t)qi * p.ldo;
#pragma unroll
                for (int c = 0; c < DH; c += 8) {
                    const bf16x8 ov = *reinterpret_cast<const bf16x8*>(orow + c), dv8 = *reinterpret_cast<const bf16x8*>(drow + c);
#pragma unroll
                    for (int e = 0; e < 8; ++e) del += (float)dv8[e] * (float)ov[e];
                }
            }
            sDel[i] = del;
        }
        __syncthreads();
        for (int c = 0; c < (rows >> 5); ++c) {
            float pv[8], dsv[8];
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int q0 = c * 32 + half * 16;
                const f32x4 sc = tile_dot<DH>(sQ, q0, li, lg, kf);
                const f32x4 dp = tile_dot<DH>(sD, q0, li, lg, vf);
                const f32x4 lse = *reinterpret_cast<const f32x4*>(sL + q0 + lg * 4);
                const f32x4 del = *reinterpret_cast<const f32x4*>(sDel + q0 + lg * 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pr = __expf(sc[r] * p.scale + kbias - lse[r]);
                    float d = dp[r], pd = pr;
                    if (do_drop) {
                        const int qq = c0 + q0 + lg * 4 + r;
                        const bool keep = rt_hash32(dseed, (uint32_t)(((size_t)bh * p.Sq + qq) * p.Sk + key)) >= thresh;
                        d = keep ? d * ks : 0.f; pd = keep ? pr * ks : 0.f;
                    }
                    pv[half * 4 + r] = pd;
                    dsv[half * 4 + r] = pr * (d - del[r]) * p.scale;
                }
            }
            const bf16x8 pf = pack8(pv), dsf = pack8(dsv);
            const unsigned char* d0 = sD + (c * 32 + tr_r) * RS + tr_c;
            const unsigned char* q0p = sQ + (c * 32 + tr_r) * RS + tr_c;
#pragma unroll
            for (int t = 0; t < DT; ++t) {
                const bf16x8 dof = tr_pair(d0 + t * 32, d0 + 16 * RS + t * 32);
                dv[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dof, pf, dv[t], 0, 0, 0);
                const bf16x8 qtf = tr_pair(q0p + t * 32, q0p + 16 * RS + t * 32);
                dk[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qtf, dsf, dk[t], 0, 0, 0);
            }
        }
    }
    if (key < p.Sk) {
        bf16_t* kro = (bf16_t*)p.dk + ((size_t)b * p.Sk + key) * p.lddk + h * DH;
        bf16_t* vro = (bf16_t*)p.dv + ((size_t)b * p.Sk + key) * p.lddv + h * DH;
#pragma unroll
        for (int t = 0; t < DT; ++t) {
            bf16x4 a, c2;
#pragma unroll
            for (int r = 0; r < 4; ++r) { a[r] = (bf16_t)dk[t][r]; c2[r] = (bf16_t)dv[t][r]; }
            *reinterpret_cast<bf16x4*>(kro + t * 16 + lg * 4) = a;
            *reinterpret_cast<bf16x4*>(vro + t * 16 + lg * 4) = c2;
        }
    }
}

// both halves in one launch, as attn_bwd_fused_kernel: blockIdx.y < ny_dq -> the dQ row blocks, the rest -> the dK / dV key blocks
template <int DH, int NW>
__global__ __launch_bounds__(64 * NW) void attn_bwd_long_kernel(const rt_attn_bwd_desc p, const int ny_dq, const int ch) {
    if ((int)blockIdx.y < ny_dq) attn_bwd_dq_long_body<DH, NW>(p, (int)blockIdx.y, ch);
    else attn_bwd_dkv_long_body<DH, NW>(p, (int)blockIdx.y - ny_dq, ch);
}

// Opt a kernel into the full 160 KiB of dynamic LDS once per kernel (not per launch: keeps launches capturable in a
// hipGraph).  Keyed by the function pointer (kernels of equal signature share a C++ type).
template <typename K>
int set_smem(K kernel, size_t bytes) {
    if (bytes > 160 * 1024) return RT_ERR_UNSUPPORTED;
    static const void* seen[16];
    static int nseen = 0;
    const void* fp = reinterpret_cast<const void*>(kernel);
    for (int i = 0; i < nseen; ++i) if (seen[i] == fp) return RT_OK;
    hipError_t e = hipFuncSetAttribute(fp, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    if (nseen < 16) seen[nseen++] = fp;
    return RT_OK;
}

size_t smem_bytes(int inner, int dh) {
    const size_t ip = (size_t)((inner + 31) & ~31);
    return 2 * ip * (dh * 2 + 32) + 2 * sizeof(float) * ip;
}

// rows of the inner axis per LDS pass of the long-axis kernels; 0 = the whole-axis kernels apply.  REFTR_ATTN_CHUNK (a multiple of
// 32) forces the long-axis kernels with that chunk on every shape (tests: bit-identity with the whole-axis kernels)
int long_chunk(int inner, int dh) {
    static const int env = getenv("REFTR_ATTN_CHUNK") ? atoi(getenv("REFTR_ATTN_CHUNK")) : 0;
    const int cap = dh == 32 ? LongGeo<32>::CH : LongGeo<64>::CH;
    if (env >= 32) return ((env < cap ? env : cap) + 31) & ~31;
    return smem_bytes(inner, dh) > 160 * 1024 ? cap : 0;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// One query per (batch, head): the decoder's cross-attention with a single referring query per image (Sq = 1, Sk = L + h*w).
// The MFMA kernels above are built for 16-query tiles and spend ~15 us of latency on this case; here a 256-thread workgroup
// owns one (b, h): each thread takes keys t, t + 256, ... (a 64-B K row and V row per key), the softmax is a block
// reduction and the output / dq a 32-wide reduction over the block.  Same arithmetic order of the definition, fp32.
// Backward produces dq, dk and dv in the same launch.
// ------------------------------------------------------------------------------------------------
constexpr int Q1_MAXK = 3;            // keys per thread: Sk <= 768

__device__ __forceinline__ void q1_load_row(const bf16_t* row, float* out32) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(row + c * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) out32[c * 8 + e] = (float)v[e];
    }
}

// block-wide sum of a 32-vector held per thread; result (all 32 values) valid in every thread
__device__ __forceinline__ void q1_block_sum32(float* v, float (*sm)[32]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int d = 0; d < 32; ++d) v[d] = rt_wave_sum(v[d]);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int d = 0; d < 32; ++d) sm[wave][d] = v[d];
    }
    __syncthreads();
#pragma unroll
    for (int d = 0; d < 32; ++d) v[d] = sm[0][d] + sm[1][d] + sm[2][d] + sm[3][d];
}

struct Q1Row { bf16x8 c[4]; };
__device__ __forceinline__ Q1Row q1_raw(const bf16_t* row) {
    Q1Row r;
#pragma unroll
    for (int c = 0; c < 4; ++c) r.c[c] = *reinterpret_cast<const bf16x8*>(row + c * 8);
    return r;
}
__device__ __forceinline__ float q1_dot(const Q1Row& r, const float* x) {
    float a = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int e = 0; e < 8; ++e) a += (float)r.c[c][e] * x[c * 8 + e];
    return a;
}

__global__ __launch_bounds__(256) void attn_q1_fwd_kernel(const rt_attn_desc p) {
    __shared__ float sm32[4][32];
    __shared__ float red[8];
    const int bh = blockIdx.x, b = bh / p.H, h = bh % p.H, t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    // every global load of this thread is issued up front (K and V rows of its <= 3 keys): one memory round trip
    Q1Row kr[Q1_MAXK], vr[Q1_MAXK];
    bool ok[Q1_MAXK];
#pragma unroll
    for (int i = 0; i < Q1_MAXK; ++i) {
        const int j = t + i * 256;
        const int jj = j < p.Sk ? j : 0;
        ok[i] = j < p.Sk && !(p.kpm && p.kpm[(size_t)b * p.Sk + jj]);
        kr[i] = q1_raw((const bf16_t*)p.k + ((size_t)b * p.Sk + jj) * p.ldk + h * 32);
        vr[i] = q1_raw((const bf16_t*)p.v + ((size_t)b * p.Sk + jj) * p.ldv + h * 32);
    }
    float q[32];
    q1_load_row((const bf16_t*)p.q + (size_t)b * p.ldq + h * 32, q);
    float sc[Q1_MAXK];
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < Q1_MAXK; ++i) {
        sc[i] = ok[i] ? q1_dot(kr[i], q) * p.scale : -INFINITY;
        m = fmaxf(m, sc[i]);
    }
    m = rt_wave_max(m);
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float ms = (m == -INFINITY) ? 0.f : m;
    float l = 0.f;
#pragma unroll
    for (int i = 0; i < Q1_MAXK; ++i) { sc[i] = __expf(sc[i] - ms); l += sc[i]; }
    l = rt_wave_sum(l);
    if (lane == 0) red[4 + wave] = l;
    __syncthreads();
    l = red[4] + red[5] + red[6] + red[7];
    const float inv_l = 1.f / l;                 // fully masked row: NaN below, as the reference
    if (t == 0 && p.lse) p.lse[bh] = ms + __logf(l);
    const bool do_drop = p.drop_p > 0.f;
    const uint32_t thresh = rt_drop_thresh(p.drop_p);
    const float ks = do_drop ? 1.f / (1.f - p.drop_p) : 1.f;
    const uint32_t dseed = rt_site_seed(p.seed_dev, p.drop_seed);
    float o[32];
#pragma unroll
    for (int d = 0; d < 32; ++d) o[d] = 0.f;
#pragma unroll
    for (int i = 0; i < Q1_MAXK; ++i) {
        const int j = t + i * 256;
        if (j >= p.Sk) continue;
        float pr = sc[i] * inv_l;
        if (do_drop) pr = (rt_hash32(dseed, (uint32_t)((size_t)bh * p.Sk + j)) >= thresh) ? pr * ks : 0.f;
        if (pr != 0.f || pr != pr) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int e = 0; e < 8; ++e) o[c * 8 + e] += pr * (float)vr[i].c[c][e];
        }
    }
    q1_block_sum32(o, sm32);
    if (t < 32) {
        float mine = 0.f;
#pragma unroll
        for (int d = 0; d < 32; ++d) mine = (t == d) ? o[d] : mine;
        ((bf16_t*)p.out)[(size_t)b * p.ldo + h * 32 + t] = (bf16_t)mine;
    }
}

__global__ __launch_bounds__(256) void attn_q1_bwd_kernel(const rt_attn_bwd_desc p) {
    __shared__ float sm32[4][32];
    const int bh = blockIdx.x, b = bh / p.H, h = bh % p.H, t = threadIdx.x;
    Q1Row kr[Q1_MAXK], vr[Q1_MAXK];
    bool ok[Q1_MAXK];
#pragma unroll
    for (int i = 0; i < Q1_MAXK; ++i) {
        const int j = t + i * 256;
        const int jj = j < p.Sk ? j : 0;
        ok[i] = j < p.Sk && !(p.kpm && p.kpm[(size_t)b * p.Sk + jj]);
        kr[i] = q1_raw((const bf16_t*)p.k + ((size_t)b * p.Sk + jj) * p.ldk + h * 32);
        vr[i] = q1_raw((const bf16_t*)p.v + ((size_t)b * p.Sk + jj) * p.ldv + h * 32);
    }
    float q[32], go[32], ov[32];
    q1_load_row((const bf16_t*)p.q + (size_t)b * p.ldq + h * 32, q);
    q1_load_row((const bf16_t*)p.dout + (size_t)b * p.ldo + h * 32, go);
    q1_load_row((const bf16_t*)p.out + (size_t)b * p.ldo + h * 32, ov);
    float delta = 0.f;
#pragma unroll
    for (int d = 0; d < 32; ++d) delta += go[d] * ov[d];
    const float lse = p.lse[bh];
    const bool do_drop = p.drop_p > 0.f;
    const uint32_t thresh = rt_drop_thresh(p.drop_p);
    const float ks = do_drop ? 1.f / (1.f - p.drop_p) : 1.f;
    const uint32_t dseed = rt_site_seed(p.seed_dev, p.drop_seed);
    float dq[32];
#pragma unroll
    for (int d = 0; d < 32; ++d) dq[d] = 0.f;
#pragma unroll
    for (int i = 0; i < Q1_MAXK; ++i) {
        const int j = t + i * 256;
        if (j >= p.Sk) continue;
        const float pr = ok[i] ? __expf(q1_dot(kr[i], q) * p.scale - lse) : 0.f;
        float mk = 1.f;
        if (do_drop) mk = (rt_hash32(dseed, (uint32_t)((size_t)bh * p.Sk + j)) >= thresh) ? ks : 0.f;
        const float dp = q1_dot(vr[i], go);
        const float ds = pr * (mk * dp - delta) * p.scale;
        const float pd = pr * mk;
        bf16_t* dkr = (bf16_t*)p.dk + ((size_t)b * p.Sk + j) * p.lddk + h * 32;
        bf16_t* dvr = (bf16_t*)p.dv + ((size_t)b * p.Sk + j) * p.lddv + h * 32;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            bf16x8 a8, b8;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                a8[e] = (bf16_t)(ds * q[c * 8 + e]); b8[e] = (bf16_t)(pd * go[c * 8 + e]);
                dq[c * 8 + e] += ds * (float)kr[i].c[c][e];
            }
            *reinterpret_cast<bf16x8*>(dkr + c * 8) = a8;
            *reinterpret_cast<bf16x8*>(dvr + c * 8) = b8;
        }
    }
    q1_block_sum32(dq, sm32);
    if (t < 32) {
        float mine = 0.f;
#pragma unroll
        for (int d = 0; d < 32; ++d) mine = (t == d) ? dq[d] : mine;
        ((bf16_t*)p.dq)[(size_t)b * p.lddq + h * 32 + t] = (bf16_t)mine;
    }
}

extern "C" int rt_attn_fwd(const rt_attn_desc* d, rt_stream_t stream) {
    if (!d || !d->q || !d->k || !d->v || !d->out) return RT_ERR_BADARG;
    if ((d->dh != 32 && d->dh != 64) || d->Sk <= 0 || d->Sq <= 0) return RT_ERR_UNSUPPORTED;
    if ((d->ldq | d->ldk | d->ldv | d->ldo) & 7) return RT_ERR_UNSUPPORTED;
    static const int q1_env = RT_TUNE("REFTR_ATTN_Q1", 1);
    if (q1_env && d->Sq == 1 && d->dh == 32 && d->Sk <= 256 * Q1_MAXK) {
        hipLaunchKernelGGL(attn_q1_fwd_kernel, dim3(d->B * d->H), dim3(256), 0, (hipStream_t)stream, *d);
        RT_CHECK_LAUNCH();
        return RT_OK;
    }
    int rc;
    if (const int ch = long_chunk(d->Sk, d->dh)) {
        const size_t lsmem = smem_bytes(ch, d->dh);
        const dim3 lgrid(d->B * d->H, (d->Sq + 127) / 128);
#define RT_ATTN_FWD_LONG(DHV) do { if ((rc = set_smem(attn_fwd_long_kernel<DHV, 8>, lsmem)) != RT_OK) return rc; \
        hipLaunchKernelGGL((attn_fwd_long_kernel<DHV, 8>), lgrid, dim3(512), lsmem, (hipStream_t)stream, *d, ch); } while (0)
        if (d->dh == 32) RT_ATTN_FWD_LONG(32); else RT_ATTN_FWD_LONG(64);
#undef RT_ATTN_FWD_LONG
        RT_CHECK_LAUNCH();
        return RT_OK;
    }
    const size_t smem = smem_bytes(d->Sk, d->dh);
    static const int nw_env = RT_TUNE("REFTR_ATTN_NW", 8);
    const int nw = nw_env == 4 ? 4 : (nw_env == 16 ? 16 : 8);
    // (b, h) on grid.x: the row blocks of one head get ids bh, bh + B*H, ... -> the same XCD whenever B*H is a multiple of 8, so the
    // head's K / V rows cross the fabric once per XCD instead of once per block
    const dim3 grid(d->B * d->H, (d->Sq + 16 * nw - 1) / (16 * nw));
#define RT_ATTN_FWD(DHV, NWV) do { if ((rc = set_smem(attn_fwd_kernel<DHV, NWV>, smem)) != RT_OK) return rc; \
        hipLaunchKernelGGL((attn_fwd_kernel<DHV, NWV>), grid, dim3(64 * NWV), smem, (hipStream_t)stream, *d); } while (0)
    static const int reg_env = RT_TUNE("REFTR_ATTN_REG", 1);
    const int tiles = ((d->Sk + 31) & ~31) >> 4;
    if (reg_env && nw == 8 && tiles <= 28) {          // 28 tiles = 180 VGPRs; 48 would spill: longer rows keep the two-pass kernel
#define RT_ATTN_FWD_REG(DHV, NTV) do { if ((rc = set_smem(attn_fwd_reg_kernel<DHV, 8, NTV>, smem)) != RT_OK) return rc; \
        hipLaunchKernelGGL((attn_fwd_reg_kernel<DHV, 8, NTV>), grid, dim3(512), smem, (hipStream_t)stream, *d); } while (0)
        if (d->dh == 32) { if (tiles <= 8) RT_ATTN_FWD_REG(32, 8); else RT_ATTN_FWD_REG(32, 28); }
        else             { if (tiles <= 8) RT_ATTN_FWD_REG(64, 8); else RT_ATTN_FWD_REG(64, 28); }
#undef RT_ATTN_FWD_REG
        RT_CHECK_LAUNCH();
        return RT_OK;
    }
    if (d->dh == 32) { if (nw == 8) RT_ATTN_FWD(32, 8); else if (nw == 4) RT_ATTN_FWD(32, 4); else RT_ATTN_FWD(32, 16); }
    else             { if (nw == 8) RT_ATTN_FWD(64, 8); else if (nw == 4) RT_ATTN_FWD(64, 4); else RT_ATTN_FWD(64, 16); }
#undef RT_ATTN_FWD
    RT_CHECK_LAUNCH();
    return RT_OK;
}

extern "C" int rt_attn_bwd(const rt_attn_bwd_desc* d, rt_stream_t stream) {
    if (!d || !d->q || !d->k || !d->v || !d->out || !d->dout || !d->lse || !d->delta || !d->dq || !d->dk || !d->dv)
        return RT_ERR_BADARG;
    if ((d->dh != 32 && d->dh != 64) || d->Sk <= 0 || d->Sq <= 0) return RT_ERR_UNSUPPORTED;
    if ((d->ldq | d->ldk | d->ldv | d->ldo | d->lddq | d->lddk | d->lddv) & 7) return RT_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    static const int q1_env = RT_TUNE("REFTR_ATTN_Q1", 1);
    if (q1_env && d->Sq == 1 && d->dh == 32 && d->Sk <= 256 * Q1_MAXK) {
        hipLaunchKernelGGL(attn_q1_bwd_kernel, dim3(d->B * d->H), dim3(256), 0, s, *d);
        RT_CHECK_LAUNCH();
        return RT_OK;
    }
    int rc;
    {
        const int ch1 = long_chunk(d->Sk, d->dh), ch2 = long_chunk(d->Sq, d->dh);
        if (ch1 || ch2) {          // either inner axis too long for one LDS pass: both halves walk their axis in chunks
            const int ch = ch1 > ch2 ? ch1 : ch2;
            const size_t lsmem = smem_bytes(ch, d->dh);
            const int ny_dq = (d->Sq + 127) / 128, ny_kv = (d->Sk + 127) / 128;
#define RT_ATTN_BWD_LONG(DHV) do { if ((rc = set_smem(attn_bwd_long_kernel<DHV, 8>, lsmem)) != RT_OK) return rc; \
            hipLaunchKernelGGL((attn_bwd_long_kernel<DHV, 8>), dim3(d->B * d->H, ny_dq + ny_kv), dim3(512), lsmem, s, *d, ny_dq, ch); } while (0)
            if (d->dh == 32) RT_ATTN_BWD_LONG(32); else RT_ATTN_BWD_LONG(64);
#undef RT_ATTN_BWD_LONG
            RT_CHECK_LAUNCH();
            return RT_OK;
        }
    }
    const size_t smem1 = smem_bytes(d->Sk, d->dh), smem2 = smem_bytes(d->Sq, d->dh);
    static const int nw_env = RT_TUNE("REFTR_ATTN_NW", 8);
    const int nw = nw_env == 4 ? 4 : (nw_env == 16 ? 16 : 8);
    const dim3 g1(d->B * d->H, (d->Sq + 16 * nw - 1) / (16 * nw)), g2(d->B * d->H, (d->Sk + 16 * nw - 1) / (16 * nw));
    static const int fused_env = RT_TUNE("REFTR_ATTN_BWD_FUSED", 1);
    if (fused_env && nw == 8) {
        const size_t smem = smem1 > smem2 ? smem1 : smem2;
        const dim3 g(d->B * d->H, g1.y + g2.y);
#define RT_ATTN_BWD_F(DHV) do { if ((rc = set_smem(attn_bwd_fused_kernel<DHV, 8>, smem)) != RT_OK) return rc; \
        hipLaunchKernelGGL((attn_bwd_fused_kernel<DHV, 8>), g, dim3(512), smem, s, *d, (int)g1.y); } while (0)
        if (d->dh == 32) RT_ATTN_BWD_F(32); else RT_ATTN_BWD_F(64);
#undef RT_ATTN_BWD_F
        RT_CHECK_LAUNCH();
        return RT_OK;
    }
#define RT_ATTN_BWD(DHV, NWV) do { if ((rc = set_smem(attn_bwd_dq_kernel<DHV, NWV>, smem1)) != RT_OK) return rc; \
        if ((rc = set_smem(attn_bwd_dkv_kernel<DHV, NWV>, smem2)) != RT_OK) return rc; \
        hipLaunchKernelGGL((attn_bwd_dq_kernel<DHV, NWV>), g1, dim3(64 * NWV), smem1, s, *d); \
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<DHV, NWV>), g2, dim3(64 * NWV), smem2, s, *d); } while (0)
    if (d->dh == 32) { if (nw == 8) RT_ATTN_BWD(32, 8); else if (nw == 4) RT_ATTN_BWD(32, 4); else RT_ATTN_BWD(32, 16); }
    else             { if (nw == 8) RT_ATTN_BWD(64, 8); else if (nw == 4) RT_ATTN_BWD(64, 4); else RT_ATTN_BWD(64, 16); }
#undef RT_ATTN_BWD
    RT_CHECK_LAUNCH();
    return RT_OK;
}
