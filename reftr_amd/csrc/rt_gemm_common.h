// Shared pieces of the implicit-GEMM translation units (rt_gemm.hip, rt_gemm_ksplit.hip): the launch-side argument block and
// the fused epilogues.  Internal to the library (nothing here is part of the C ABI).
#pragma once
#include "rt_common.h"

struct GemmArgs {
    const bf16_t* src; const bf16_t* wgt;
    bf16_t* out_bf16; float* out_f32; bf16_t* out_preact; float* acc2_f32;
    const float* bias; const float* res_f32; const bf16_t* res_bf16; const bf16_t* gate; const bf16_t* preact; const bf16_t* dtanh;
    int B, SH, SW, SC, DH, DW, N, KH, KW, stride, pad, transposed, act, res_first;
    float gate_scale, drop_p; uint32_t drop_seed; int drop_shift;
    int M, K, sshift, xcd, early, epi_lds, abl, prefetch, mfast, dil;
    unsigned src_bytes, wgt_bytes;
    const uint32_t* seed_dev;
};

// epilogue of one lane's 4 consecutive output features of row m:
// +bias -> [+res] -> act -> dropout -> [+res] -> *gate -> *gelu'(preact) -> *(1 - dtanh^2) -> store
// The operands epilogue4 reads, fetched ahead of the reduction (skinny kernel: the M <= 16 launches are pure latency chains --
// launch -> operand loads -> MFMA -> LDS reduce -> EPILOGUE LOADS -> stores -- and the epilogue's loads depend on nothing the
// kernel computes, so they are requested first and cost no round trip of their own).
struct Epi4Pre { f32x4 bias, res; bf16x4 resb, gate, preact, dtanh; };
static __device__ __forceinline__ Epi4Pre epi4_prefetch(const GemmArgs& p, int m, int n) {
    Epi4Pre e;
    const size_t o = (size_t)m * p.N + n;
    const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
    e.bias = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + n) : z;
    e.res = p.res_f32 ? *reinterpret_cast<const f32x4*>(p.res_f32 + o) : z;
    e.resb = p.res_bf16 ? *reinterpret_cast<const bf16x4*>(p.res_bf16 + o) : bf16x4{};
    e.gate = p.gate ? *reinterpret_cast<const bf16x4*>(p.gate + o) : bf16x4{};
    e.preact = p.preact ? *reinterpret_cast<const bf16x4*>(p.preact + o) : bf16x4{};
    e.dtanh = p.dtanh ? *reinterpret_cast<const bf16x4*>(p.dtanh + o) : bf16x4{};
    return e;
}

template <bool PRE = false>
static __device__ __forceinline__ void epilogue4(const GemmArgs& p, int m, int n, f32x4 v, const Epi4Pre* e = nullptr) {
    if (p.bias) v += PRE ? e->bias : *reinterpret_cast<const f32x4*>(p.bias + n);
    const size_t o = (size_t)m * p.N + n;
    if (p.out_preact) {
        bf16x4 pv;
#pragma unroll
        for (int r = 0; r < 4; ++r) pv[r] = (bf16_t)v[r];
        *reinterpret_cast<bf16x4*>(p.out_preact + o) = pv;
    }
    if (p.res_first) {
        if (p.res_f32) v += PRE ? e->res : *reinterpret_cast<const f32x4*>(p.res_f32 + o);
        if (p.res_bf16) {
            const bf16x4 rr = PRE ? e->resb : *reinterpret_cast<const bf16x4*>(p.res_bf16 + o);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += (float)rr[r];
        }
    }
    if (p.act == RT_ACT_RELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
    } else if (p.act == RT_ACT_GELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = rt_gelu(v[r]);
    } else if (p.act == RT_ACT_TANH) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = tanhf(v[r]);
    }
    if (p.drop_p > 0.f) {
        const uint32_t thresh = rt_drop_thresh(p.drop_p);
        const float keep_scale = 1.0f / (1.0f - p.drop_p);
        const uint32_t seed = rt_site_seed(p.seed_dev, p.drop_seed);
#pragma unroll
        for (int r = 0; r < 4; ++r)
            v[r] = (rt_hash32(seed, (uint32_t)((o + r) >> p.drop_shift)) >= thresh) ? v[r] * keep_scale : 0.f;
    }
    if (!p.res_first) {
        if (p.res_f32) v += PRE ? e->res : *reinterpret_cast<const f32x4*>(p.res_f32 + o);
        if (p.res_bf16) {
            const bf16x4 rr = PRE ? e->resb : *reinterpret_cast<const bf16x4*>(p.res_bf16 + o);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += (float)rr[r];
        }
    }
    if (p.gate) {
        const bf16x4 gg = PRE ? e->gate : *reinterpret_cast<const bf16x4*>(p.gate + o);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = ((float)gg[r] > 0.f) ? v[r] * p.gate_scale : 0.f;
    }
    if (p.preact) {
        const bf16x4 uu = PRE ? e->preact : *reinterpret_cast<const bf16x4*>(p.preact + o);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] *= rt_gelu_grad((float)uu[r]);
    }
    if (p.dtanh) {
        const bf16x4 tt = PRE ? e->dtanh : *reinterpret_cast<const bf16x4*>(p.dtanh + o);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] *= (1.f - (float)tt[r] * (float)tt[r]);
    }
    if (p.out_f32) *reinterpret_cast<f32x4*>(p.out_f32 + o) = v;
    if (p.acc2_f32) *reinterpret_cast<f32x4*>(p.acc2_f32 + o) += v;          // a second, accumulating destination (one owner per element)
    if (p.out_bf16) {
        bf16x4 ov;
#pragma unroll
        for (int r = 0; r < 4; ++r) ov[r] = (bf16_t)v[r];
        *reinterpret_cast<bf16x4*>(p.out_bf16 + o) = ov;
    }
}

// Same epilogue on 8 consecutive output features of row m (the LDS-staged, row-coalesced path of the DMA kernel):
// every global access is a full 16-B (bf16) / 32-B (fp32) contiguous piece of one output row.
typedef __attribute__((ext_vector_type(8))) float f32x8;
// `pre` (compile-time): the bf16 residual / gate pieces of this row piece were fetched at the start of the workgroup (pres, pgate).
template <bool PRE = false>
static __device__ __forceinline__ void epilogue8(const GemmArgs& p, int m, int n, f32x8 v, const bf16x8 pres = bf16x8{}, const bf16x8 pgate = bf16x8{}) {
    const size_t o = (size_t)m * p.N + n;
    if (p.bias) {
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.bias + n), b1 = *reinterpret_cast<const f32x4*>(p.bias + n + 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) { v[r] += b0[r]; v[4 + r] += b1[r]; }
    }
    if (p.out_preact) {
        bf16x8 pv;
#pragma unroll
        for (int r = 0; r < 8; ++r) pv[r] = (bf16_t)v[r];
        *reinterpret_cast<bf16x8*>(p.out_preact + o) = pv;
    }
    auto add_res = [&]() __attribute__((always_inline)) {
        if (p.res_f32) {
            const f32x4 r0 = *reinterpret_cast<const f32x4*>(p.res_f32 + o), r1 = *reinterpret_cast<const f32x4*>(p.res_f32 + o + 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) { v[r] += r0[r]; v[4 + r] += r1[r]; }
        }
        if (p.res_bf16) {
            const bf16x8 rr = PRE ? pres : *reinterpret_cast<const bf16x8*>(p.res_bf16 + o);
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] += (float)rr[r];
        }
    };
    if (p.res_first) add_res();
    if (p.act == RT_ACT_RELU) {
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = fmaxf(v[r], 0.f);
    } else if (p.act == RT_ACT_GELU) {
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = rt_gelu(v[r]);
    } else if (p.act == RT_ACT_TANH) {
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = tanhf(v[r]);
    }
    if (p.drop_p > 0.f) {
        const uint32_t thresh = rt_drop_thresh(p.drop_p);
        const float keep_scale = 1.0f / (1.0f - p.drop_p);
        const uint32_t seed = rt_site_seed(p.seed_dev, p.drop_seed);
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = (rt_hash32(seed, (uint32_t)((o + r) >> p.drop_shift)) >= thresh) ? v[r] * keep_scale : 0.f;
    }
    if (!p.res_first) add_res();
    if (p.gate) {
        const bf16x8 gg = PRE ? pgate : *reinterpret_cast<const bf16x8*>(p.gate + o);
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = ((float)gg[r] > 0.f) ? v[r] * p.gate_scale : 0.f;
    }
    if (p.preact) {
        const bf16x8 uu = *reinterpret_cast<const bf16x8*>(p.preact + o);
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] *= rt_gelu_grad((float)uu[r]);
    }
    if (p.dtanh) {
        const bf16x8 tt = *reinterpret_cast<const bf16x8*>(p.dtanh + o);
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] *= (1.f - (float)tt[r] * (float)tt[r]);
    }
    if (p.out_f32) {
        *reinterpret_cast<f32x4*>(p.out_f32 + o) = f32x4{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(p.out_f32 + o + 4) = f32x4{v[4], v[5], v[6], v[7]};
    }
    if (p.acc2_f32) {
        *reinterpret_cast<f32x4*>(p.acc2_f32 + o) += f32x4{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(p.acc2_f32 + o + 4) += f32x4{v[4], v[5], v[6], v[7]};
    }
    if (p.out_bf16) {
        bf16x8 ov;
#pragma unroll
        for (int r = 0; r < 8; ++r) ov[r] = (bf16_t)v[r];
        *reinterpret_cast<bf16x8*>(p.out_bf16 + o) = ov;
    }
}


// rt_gemm_pipe.hip: software-pipelined LDS-DMA variants (hints 2xx)
int rt_launch_gemm_pipe(const GemmArgs& a, int hint, hipStream_t s);
// rt_gemm_astat.hip: activation-stationary form of the short-K / wide-N dense products (hint 501)
bool rt_gemm_astat_ok(const GemmArgs& a);
int rt_launch_gemm_astat(const GemmArgs& a, hipStream_t s);
// rt_gemm_pp.hip: K-parity ping-pong LDS-DMA variants (hints 3xx)
int rt_launch_gemm_pp(const GemmArgs& a, int hint, hipStream_t s);
