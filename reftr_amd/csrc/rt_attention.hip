// Masked multi-head attention forward / backward for the RefTR sequence lengths
// (VL encoder S = L + HW/32^2 <= 768, dh = 32; BERT L <= 128, dh = 64; decoder n_q*n_ph queries).
//
// Work decomposition: a workgroup owns one (batch, head) and a 64-row tile of the "outer" axis; the whole
// inner-axis operand pair (K,V for forward / dQ; Q,dO for dK/dV) is staged once in LDS as bf16 rows padded
// by 16 B (conflict-free ds_read_b128 when lane j reads row j).  One 64-lane wave processes one outer row at
// a time with the inner axis spread across lanes; softmax statistics are wave reductions and the
// [64 lanes x dh] partial outputs are combined with a butterfly reduce-scatter (dh shuffles, no LDS).
// The score matrix never touches HBM; backward recomputes probabilities from the saved log-sum-exp.
#include "rt_common.h"

namespace {

constexpr int TMAX = 12;   // inner axis <= 768

template <int DH>
__device__ __forceinline__ void load_row_f32(const bf16_t* p, float (&r)[DH]) {
#pragma unroll
    for (int c = 0; c < DH / 8; ++c) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(p + c * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) r[c * 8 + e] = (float)v[e];
    }
}

template <int DH>
__device__ __forceinline__ float dot_lds(const unsigned char* row, const float (&q)[DH]) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < DH / 8; ++c) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(row + c * 16);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += q[c * 8 + e] * (float)v[e];
    }
    return s;
}

template <int DH>
__device__ __forceinline__ void axpy_lds(const unsigned char* row, float a, float (&o)[DH]) {
#pragma unroll
    for (int c = 0; c < DH / 8; ++c) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(row + c * 16);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[c * 8 + e] += a * (float)v[e];
    }
}

// Sum o[d] over the 64 lanes; afterwards lane L holds the total for d = (DH == 64 ? L : L >> 1) in o[0].
template <int DH>
__device__ __forceinline__ void butterfly_reduce(float (&o)[DH], int lane) {
    int mask = 32;
#pragma unroll
    for (int h = DH / 2; h >= 1; h >>= 1) {
        const bool up = (lane & mask) != 0;
#pragma unroll
        for (int i = 0; i < h; ++i) {
            const float keep = up ? o[i + h] : o[i];
            const float send = up ? o[i] : o[i + h];
            o[i] = keep + __shfl_xor(send, mask, 64);
        }
        mask >>= 1;
    }
    if (DH == 32) o[0] += __shfl_xor(o[0], 1, 64);
}

template <int DH>
__device__ __forceinline__ void stage_rows(unsigned char* dst, const bf16_t* src, int rows, int ld, int tid) {
    constexpr int RS = DH * 2 + 16;
    constexpr int CPR = DH / 8;
    for (int c = tid; c < rows * CPR; c += 256) {
        const int r = c / CPR, cc = c % CPR;
        *reinterpret_cast<uint4*>(dst + r * RS + cc * 16) = *reinterpret_cast<const uint4*>(src + (size_t)r * ld + cc * 8);
    }
}

template <int DH>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const rt_attn_desc p) {
    constexpr int RS = DH * 2 + 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sK = smem;
    unsigned char* sV = smem + (size_t)p.Sk * RS;
    const int bh = blockIdx.y, b = bh / p.H, h = bh % p.H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bf16_t* kb = (const bf16_t*)p.k + (size_t)b * p.Sk * p.ldk + h * DH;
    const bf16_t* vb = (const bf16_t*)p.v + (size_t)b * p.Sk * p.ldv + h * DH;
    stage_rows<DH>(sK, kb, p.Sk, p.ldk, threadIdx.x);
    stage_rows<DH>(sV, vb, p.Sk, p.ldv, threadIdx.x);
    __syncthreads();
    const bool do_drop = p.drop_p > 0.f;
    const uint32_t thresh = rt_drop_thresh(p.drop_p);
    const float ks = do_drop ? 1.f / (1.f - p.drop_p) : 1.f;
    const int nt = (p.Sk + 63) >> 6;
    const uint8_t* kpm = p.kpm ? p.kpm + (size_t)b * p.Sk : nullptr;

    for (int r = 0; r < 16; ++r) {
        const int i = blockIdx.x * 64 + wave * 16 + r;
        if (i >= p.Sq) break;
        float q[DH];
        load_row_f32<DH>((const bf16_t*)p.q + ((size_t)b * p.Sq + i) * p.ldq + h * DH, q);
        float s[TMAX];
        float m = -INFINITY;
#pragma unroll
        for (int t = 0; t < TMAX; ++t) {
            s[t] = -INFINITY;
            if (t < nt) {
                const int j = lane + (t << 6);
                if (j < p.Sk && !(kpm && kpm[j])) s[t] = dot_lds<DH>(sK + (size_t)j * RS, q) * p.scale;
                m = fmaxf(m, s[t]);
            }
        }
        m = rt_wave_max(m);
        float l = 0.f;
#pragma unroll
        for (int t = 0; t < TMAX; ++t)
            if (t < nt) { s[t] = __expf(s[t] - m); l += s[t]; }   // all-masked row: (-inf) - (-inf) = NaN, as the reference
        l = rt_wave_sum(l);
        const float inv_l = 1.f / l;
        float o[DH];
#pragma unroll
        for (int d = 0; d < DH; ++d) o[d] = 0.f;
#pragma unroll
        for (int t = 0; t < TMAX; ++t) {
            if (t < nt) {
                const int j = lane + (t << 6);
                if (j < p.Sk) {
                    float pj = s[t] * inv_l;
                    if (do_drop)
                        pj = (rt_hash32(p.drop_seed, (uint32_t)(((size_t)bh * p.Sq + i) * p.Sk + j)) >= thresh) ? pj * ks : 0.f;
                    axpy_lds<DH>(sV + (size_t)j * RS, pj, o);
                }
            }
        }
        butterfly_reduce<DH>(o, lane);
        bf16_t* orow = (bf16_t*)p.out + ((size_t)b * p.Sq + i) * p.ldo + h * DH;
        if (DH == 64) orow[lane] = (bf16_t)o[0];
        else if ((lane & 1) == 0) orow[lane >> 1] = (bf16_t)o[0];
        if (lane == 0 && p.lse) p.lse[(size_t)bh * p.Sq + i] = m + __logf(l);
    }
}

// dQ (+ delta = dO . O) : same traversal as forward
template <int DH>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const rt_attn_bwd_desc p) {
    constexpr int RS = DH * 2 + 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sK = smem;
    unsigned char* sV = smem + (size_t)p.Sk * RS;
    const int bh = blockIdx.y, b = bh / p.H, h = bh % p.H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    stage_rows<DH>(sK, (const bf16_t*)p.k + (size_t)b * p.Sk * p.ldk + h * DH, p.Sk, p.ldk, threadIdx.x);
    stage_rows<DH>(sV, (const bf16_t*)p.v + (size_t)b * p.Sk * p.ldv + h * DH, p.Sk, p.ldv, threadIdx.x);
    __syncthreads();
    const bool do_drop = p.drop_p > 0.f;
    const uint32_t thresh = rt_drop_thresh(p.drop_p);
    const float ks = do_drop ? 1.f / (1.f - p.drop_p) : 1.f;
    const int nt = (p.Sk + 63) >> 6;
    const uint8_t* kpm = p.kpm ? p.kpm + (size_t)b * p.Sk : nullptr;

    for (int r = 0; r < 16; ++r) {
        const int i = blockIdx.x * 64 + wave * 16 + r;
        if (i >= p.Sq) break;
        float q[DH], dO[DH];
        load_row_f32<DH>((const bf16_t*)p.q + ((size_t)b * p.Sq + i) * p.ldq + h * DH, q);
        load_row_f32<DH>((const bf16_t*)p.dout + ((size_t)b * p.Sq + i) * p.ldo + h * DH, dO);
        float delta = 0.f;
        {
            float O[DH];
            load_row_f32<DH>((const bf16_t*)p.out + ((size_t)b * p.Sq + i) * p.ldo + h * DH, O);
#pragma unroll
            for (int d = 0; d < DH; ++d) delta += dO[d] * O[d];
        }
        const float lse = p.lse[(size_t)bh * p.Sq + i];
        if (lane == 0) p.delta[(size_t)bh * p.Sq + i] = delta;
        float dq[DH];
#pragma unroll
        for (int d = 0; d < DH; ++d) dq[d] = 0.f;
#pragma unroll
        for (int t = 0; t < TMAX; ++t) {
            if (t < nt) {
                const int j = lane + (t << 6);
                if (j < p.Sk && !(kpm && kpm[j])) {
                    const float sij = dot_lds<DH>(sK + (size_t)j * RS, q) * p.scale;
                    const float pij = __expf(sij - lse);
                    float dp = dot_lds<DH>(sV + (size_t)j * RS, dO);
                    if (do_drop)
                        dp = (rt_hash32(p.drop_seed, (uint32_t)(((size_t)bh * p.Sq + i) * p.Sk + j)) >= thresh) ? dp * ks : 0.f;
                    const float ds = pij * (dp - delta) * p.scale;
                    axpy_lds<DH>(sK + (size_t)j * RS, ds, dq);
                }
            }
        }
        butterfly_reduce<DH>(dq, lane);
        bf16_t* drow = (bf16_t*)p.dq + ((size_t)b * p.Sq + i) * p.lddq + h * DH;
        if (DH == 64) drow[lane] = (bf16_t)dq[0];
        else if ((lane & 1) == 0) drow[lane >> 1] = (bf16_t)dq[0];
    }
}

// dK, dV: outer axis = keys, inner axis = queries (Q and dO staged in LDS)
template <int DH>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const rt_attn_bwd_desc p) {
    constexpr int RS = DH * 2 + 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sQ = smem;
    unsigned char* sD = smem + (size_t)p.Sq * RS;
    float* sL = reinterpret_cast<float*>(smem + 2 * (size_t)p.Sq * RS);
    float* sDel = sL + p.Sq;
    const int bh = blockIdx.y, b = bh / p.H, h = bh % p.H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    stage_rows<DH>(sQ, (const bf16_t*)p.q + (size_t)b * p.Sq * p.ldq + h * DH, p.Sq, p.ldq, threadIdx.x);
    stage_rows<DH>(sD, (const bf16_t*)p.dout + (size_t)b * p.Sq * p.ldo + h * DH, p.Sq, p.ldo, threadIdx.x);
    for (int i = threadIdx.x; i < p.Sq; i += 256) {
        sL[i] = p.lse[(size_t)bh * p.Sq + i];
        sDel[i] = p.delta[(size_t)bh * p.Sq + i];
    }
    __syncthreads();
    const bool do_drop = p.drop_p > 0.f;
    const uint32_t thresh = rt_drop_thresh(p.drop_p);
    const float ks = do_drop ? 1.f / (1.f - p.drop_p) : 1.f;
    const int nt = (p.Sq + 63) >> 6;
    const uint8_t* kpm = p.kpm ? p.kpm + (size_t)b * p.Sk : nullptr;

    for (int r = 0; r < 16; ++r) {
        const int j = blockIdx.x * 64 + wave * 16 + r;
        if (j >= p.Sk) break;
        float dk[DH], dv[DH];
#pragma unroll
        for (int d = 0; d < DH; ++d) { dk[d] = 0.f; dv[d] = 0.f; }
        const bool masked = kpm && kpm[j];
        if (!masked) {
            float kr[DH], vr[DH];
            load_row_f32<DH>((const bf16_t*)p.k + ((size_t)b * p.Sk + j) * p.ldk + h * DH, kr);
            load_row_f32<DH>((const bf16_t*)p.v + ((size_t)b * p.Sk + j) * p.ldv + h * DH, vr);
#pragma unroll
            for (int t = 0; t < TMAX; ++t) {
                if (t < nt) {
                    const int i = lane + (t << 6);
                    if (i < p.Sq) {
                        const float sij = dot_lds<DH>(sQ + (size_t)i * RS, kr) * p.scale;
                        const float pij = __expf(sij - sL[i]);
                        float dp = dot_lds<DH>(sD + (size_t)i * RS, vr);
                        float pd = pij;
                        if (do_drop) {
                            const bool keep = rt_hash32(p.drop_seed, (uint32_t)(((size_t)bh * p.Sq + i) * p.Sk + j)) >= thresh;
                            dp = keep ? dp * ks : 0.f;
                            pd = keep ? pij * ks : 0.f;
                        }
                        axpy_lds<DH>(sD + (size_t)i * RS, pd, dv);
                        const float ds = pij * (dp - sDel[i]) * p.scale;
                        axpy_lds<DH>(sQ + (size_t)i * RS, ds, dk);
                    }
                }
            }
        }
        butterfly_reduce<DH>(dk, lane);
        butterfly_reduce<DH>(dv, lane);
        bf16_t* kro = (bf16_t*)p.dk + ((size_t)b * p.Sk + j) * p.lddk + h * DH;
        bf16_t* vro = (bf16_t*)p.dv + ((size_t)b * p.Sk + j) * p.lddv + h * DH;
        if (DH == 64) { kro[lane] = (bf16_t)dk[0]; vro[lane] = (bf16_t)dv[0]; }
        else if ((lane & 1) == 0) { kro[lane >> 1] = (bf16_t)dk[0]; vro[lane >> 1] = (bf16_t)dv[0]; }
    }
}

template <typename K>
int set_smem(K kernel, size_t bytes) {
    if (bytes > 160 * 1024) return RT_ERR_UNSUPPORTED;
    if (bytes > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return (int)e;
    }
    return RT_OK;
}

}  // namespace

extern "C" int rt_attn_fwd(const rt_attn_desc* d, rt_stream_t stream) {
    if (!d || !d->q || !d->k || !d->v || !d->out) return RT_ERR_BADARG;
    if ((d->dh != 32 && d->dh != 64) || d->Sk <= 0 || d->Sk > 64 * TMAX || d->Sq <= 0) return RT_ERR_UNSUPPORTED;
    if ((d->ldq | d->ldk | d->ldv | d->ldo) & 7) return RT_ERR_UNSUPPORTED;
    const size_t smem = 2 * (size_t)d->Sk * (d->dh * 2 + 16);
    const dim3 grid((d->Sq + 63) / 64, d->B * d->H);
    int rc;
    if (d->dh == 32) {
        if ((rc = set_smem(attn_fwd_kernel<32>, smem)) != RT_OK) return rc;
        hipLaunchKernelGGL(attn_fwd_kernel<32>, grid, dim3(256), smem, (hipStream_t)stream, *d);
    } else {
        if ((rc = set_smem(attn_fwd_kernel<64>, smem)) != RT_OK) return rc;
        hipLaunchKernelGGL(attn_fwd_kernel<64>, grid, dim3(256), smem, (hipStream_t)stream, *d);
    }
    RT_CHECK_LAUNCH();
    return RT_OK;
}

extern "C" int rt_attn_bwd(const rt_attn_bwd_desc* d, rt_stream_t stream) {
    if (!d || !d->q || !d->k || !d->v || !d->out || !d->dout || !d->lse || !d->delta || !d->dq || !d->dk || !d->dv)
        return RT_ERR_BADARG;
    if ((d->dh != 32 && d->dh != 64) || d->Sk <= 0 || d->Sk > 64 * TMAX || d->Sq <= 0 || d->Sq > 64 * TMAX)
        return RT_ERR_UNSUPPORTED;
    if ((d->ldq | d->ldk | d->ldv | d->ldo | d->lddq | d->lddk | d->lddv) & 7) return RT_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const size_t smem1 = 2 * (size_t)d->Sk * (d->dh * 2 + 16);
    const size_t smem2 = 2 * (size_t)d->Sq * (d->dh * 2 + 16) + 2 * sizeof(float) * (size_t)d->Sq;
    const dim3 g1((d->Sq + 63) / 64, d->B * d->H), g2((d->Sk + 63) / 64, d->B * d->H);
    int rc;
    if (d->dh == 32) {
        if ((rc = set_smem(attn_bwd_dq_kernel<32>, smem1)) != RT_OK) return rc;
        if ((rc = set_smem(attn_bwd_dkv_kernel<32>, smem2)) != RT_OK) return rc;
        hipLaunchKernelGGL(attn_bwd_dq_kernel<32>, g1, dim3(256), smem1, s, *d);
        hipLaunchKernelGGL(attn_bwd_dkv_kernel<32>, g2, dim3(256), smem2, s, *d);
    } else {
        if ((rc = set_smem(attn_bwd_dq_kernel<64>, smem1)) != RT_OK) return rc;
        if ((rc = set_smem(attn_bwd_dkv_kernel<64>, smem2)) != RT_OK) return rc;
        hipLaunchKernelGGL(attn_bwd_dq_kernel<64>, g1, dim3(256), smem1, s, *d);
        hipLaunchKernelGGL(attn_bwd_dkv_kernel<64>, g2, dim3(256), smem2, s, *d);
    }
    RT_CHECK_LAUNCH();
    return RT_OK;
}
