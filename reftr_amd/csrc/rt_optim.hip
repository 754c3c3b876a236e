// Optimizer side of the step (engine_vg.py:62-67, main_vg.py:234-268) over ONE flat fp32 parameter buffer:
//   rt_sqnorm      global L2 norm^2 of the flat gradient (clip_grad_norm_'s total norm)
//   rt_adamw_flat  gradient scale (1/world) + clip coefficient + decoupled-weight-decay AdamW, 28 B/param of
//                  HBM traffic in a single streaming pass, per-range learning rates (param groups)
#include "rt_common.h"
#include <stdlib.h>

namespace {

__global__ __launch_bounds__(256) void sqnorm_kernel(const float* __restrict__ g, size_t n, float* __restrict__ out) {
    __shared__ float sm[16];
    float s = 0.f;
    const size_t n4 = n >> 2;
    const float4* g4 = reinterpret_cast<const float4*>(g);
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 v = g4[i];
        s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    for (size_t i = (n4 << 2) + blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += g[i] * g[i];
    s = rt_block_sum(s, sm);
    if (threadIdx.x == 0) atomicAdd(out, s);
}

__global__ __launch_bounds__(256) void sqnorm_bf16_kernel(const bf16_t* __restrict__ g, size_t n, float* __restrict__ out) {
    __shared__ float sm[16];
    float s = 0.f;
    const size_t n8 = n >> 3;
    const bf16x8* g8 = reinterpret_cast<const bf16x8*>(g);
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
        const bf16x8 v = g8[i];
#pragma unroll
        for (int c = 0; c < 8; ++c) { const float f = (float)v[c]; s += f * f; }
    }
    for (size_t i = (n8 << 3) + blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const float f = (float)g[i]; s += f * f; }
    s = rt_block_sum(s, sm);
    if (threadIdx.x == 0) atomicAdd(out, s);
}

__global__ __launch_bounds__(256) void adamw_kernel(const rt_adamw_desc p, const int nontemporal) {
    if (p.active && p.active[0] == 0) return;
    const float total = sqrtf(p.gnorm_sq ? p.gnorm_sq[0] : 0.f) * p.grad_scale;
    float coef = 1.f;
    if (p.max_norm > 0.f) coef = fminf(1.f, p.max_norm / (total + 1e-6f));
    if (blockIdx.x == 0 && threadIdx.x == 0 && p.gnorm_out) p.gnorm_out[0] = total;
    const float gs = p.grad_scale * coef;
    const int step = p.step_dev ? p.step_dev[0] : p.step;
    const float bc1 = 1.f - powf(p.beta1, (float)step);
    const float bc2 = 1.f - powf(p.beta2, (float)step);
    const float inv_sqrt_bc2 = rsqrtf(bc2);
    const size_t i0 = (size_t)p.span_begin >> 2, n4 = (size_t)p.span_end >> 2;
    float4* P4 = reinterpret_cast<float4*>(p.p);
    const float4* G4 = reinterpret_cast<const float4*>(p.g);
    typedef bf16_t bf16x4_t __attribute__((ext_vector_type(4)));
    const bf16x4_t* G16 = reinterpret_cast<const bf16x4_t*>(p.g16);
    float4* M4 = reinterpret_cast<float4*>(p.m);
    float4* V4 = reinterpret_cast<float4*>(p.v);
    for (size_t i = i0 + blockIdx.x * (size_t)256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const size_t e = i << 2;
        float lr = 0.f, wd = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r)
            if (r < p.n_ranges && e >= (size_t)p.range_begin[r] && e < (size_t)p.range_end[r]) {
                lr = p.lr_dev ? p.lr_dev[r] : p.range_lr[r]; wd = p.range_wd[r];
            }
        float4 pv, gv, mv, vv;
        if (nontemporal) {         // streamed once per step: do not displace the activations / operands in L2 and the MALL
            auto ntl = [](const float4* q) __attribute__((always_inline)) {
                const f32x4 t4 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(q));
                return make_float4(t4[0], t4[1], t4[2], t4[3]);
            };
            pv = ntl(P4 + i); mv = ntl(M4 + i); vv = ntl(V4 + i);
            if (!G16) gv = ntl(G4 + i);
        } else { pv = P4[i]; mv = M4[i]; vv = V4[i]; if (!G16) gv = G4[i]; }
        if (G16) { const bf16x4_t h = __builtin_nontemporal_load(G16 + i); gv = make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]); }
        float* pp = &pv.x; const float* gp = &gv.x; float* mp = &mv.x; float* vp = &vv.x;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float g = gp[c] * gs;
            pp[c] *= (1.f - lr * wd);
            mp[c] = p.beta1 * mp[c] + (1.f - p.beta1) * g;
            vp[c] = p.beta2 * vp[c] + (1.f - p.beta2) * g * g;
            const float denom = sqrtf(vp[c]) * inv_sqrt_bc2 + p.eps;
            pp[c] -= (lr / bc1) * (mp[c] / denom);
        }
        if (nontemporal) { P4[i] = pv; __builtin_nontemporal_store(f32x4{mv.x, mv.y, mv.z, mv.w}, reinterpret_cast<f32x4*>(M4 + i)); __builtin_nontemporal_store(f32x4{vv.x, vv.y, vv.z, vv.w}, reinterpret_cast<f32x4*>(V4 + i)); }
        else { P4[i] = pv; M4[i] = mv; V4[i] = vv; }
    }
}

// torch.optim.SGD(momentum, weight_decay) as the reference builds it with --sgd (main_vg.py:263-265: momentum 0.9, dampening 0, no
// Nesterov): g' = clip * scale * g + wd * p;  buf = momentum * buf + g'  (a zero-initialised buffer makes step 1 "buf = g'");
// p -= lr * buf.  Same descriptor as AdamW: m = momentum buffer, beta1 = momentum, v / beta2 / eps unused.  20 B per parameter.
__global__ __launch_bounds__(256) void sgd_kernel(const rt_adamw_desc p) {
    if (p.active && p.active[0] == 0) return;
    const float total = sqrtf(p.gnorm_sq ? p.gnorm_sq[0] : 0.f) * p.grad_scale;
    float coef = 1.f;
    if (p.max_norm > 0.f) coef = fminf(1.f, p.max_norm / (total + 1e-6f));
    if (blockIdx.x == 0 && threadIdx.x == 0 && p.gnorm_out) p.gnorm_out[0] = total;
    const float gs = p.grad_scale * coef;
    const size_t i0 = (size_t)p.span_begin >> 2, n4 = (size_t)p.span_end >> 2;
    float4* P4 = reinterpret_cast<float4*>(p.p);
    const float4* G4 = reinterpret_cast<const float4*>(p.g);
    typedef bf16_t bf16x4_t __attribute__((ext_vector_type(4)));
    const bf16x4_t* G16 = reinterpret_cast<const bf16x4_t*>(p.g16);
    float4* M4 = reinterpret_cast<float4*>(p.m);
    for (size_t i = i0 + blockIdx.x * (size_t)256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const size_t e = i << 2;
        float lr = 0.f, wd = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r)
            if (r < p.n_ranges && e >= (size_t)p.range_begin[r] && e < (size_t)p.range_end[r]) {
                lr = p.lr_dev ? p.lr_dev[r] : p.range_lr[r]; wd = p.range_wd[r];
            }
        float4 pv = P4[i], mv = M4[i], gv;
        if (G16) { const bf16x4_t h = G16[i]; gv = make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]); }
        else gv = G4[i];
        float* pp = &pv.x; const float* gp = &gv.x; float* mp = &mv.x;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float g = gp[c] * gs + wd * pp[c];
            mp[c] = p.beta1 * mp[c] + g;
            pp[c] -= lr * mp[c];
        }
        P4[i] = pv; M4[i] = mv;
    }
}

__global__ void counter_add_kernel(int32_t* c, int32_t inc) { c[0] += inc; }

}  // namespace

extern "C" int rt_counter_add(int32_t* ctr, int32_t inc, rt_stream_t stream) {
    if (!ctr) return RT_ERR_BADARG;
    hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, ctr, inc);
    RT_CHECK_LAUNCH();
    return RT_OK;
}

extern "C" int rt_sqnorm(const float* g, int64_t n, float* out, rt_stream_t stream) {
    if (!g || !out || n <= 0) return RT_ERR_BADARG;
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = rt_zero_f32(out, 1, s);
    if (e != hipSuccess) return (int)e;
    int blocks = (int)(((size_t)n / 4 + 255) / 256); if (blocks > 2048) blocks = 2048; if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(sqnorm_kernel, dim3(blocks), dim3(256), 0, s, g, (size_t)n, out);
    RT_CHECK_LAUNCH();
    return RT_OK;
}

extern "C" int rt_sqnorm_bf16(const void* g16, int64_t n, float* out, rt_stream_t stream) {
    if (!g16 || !out || n <= 0) return RT_ERR_BADARG;
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = rt_zero_f32(out, 1, s);
    if (e != hipSuccess) return (int)e;
    int blocks = (int)(((size_t)n / 8 + 255) / 256); if (blocks > 2048) blocks = 2048; if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(sqnorm_bf16_kernel, dim3(blocks), dim3(256), 0, s, (const bf16_t*)g16, (size_t)n, out);
    RT_CHECK_LAUNCH();
    return RT_OK;
}

extern "C" int rt_adamw_flat(const rt_adamw_desc* d, rt_stream_t stream) {
    if (!d || !d->p || (!d->g && !d->g16) || !d->m || !d->v || d->n <= 0 || (d->n & 3) || d->n_ranges < 1 || d->n_ranges > 8 || (d->step < 1 && !d->step_dev))
        return RT_ERR_BADARG;
    for (int r = 0; r < d->n_ranges; ++r) if ((d->range_begin[r] & 3) || (d->range_end[r] & 3)) return RT_ERR_BADARG;
    rt_adamw_desc a = *d;
    if (a.span_begin == 0 && a.span_end == 0) a.span_end = a.n;
    if (a.span_begin < 0 || a.span_end > a.n || a.span_begin >= a.span_end || (a.span_begin & 3) || (a.span_end & 3)) return RT_ERR_BADARG;
    int blocks = (int)(((size_t)(a.span_end - a.span_begin) / 4 + 255) / 256); if (blocks > 4096) blocks = 4096;
    static const int nt_env = getenv("REFTR_ADAMW_NT") ? atoi(getenv("REFTR_ADAMW_NT")) : 1;
    hipLaunchKernelGGL(adamw_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, nt_env);
    RT_CHECK_LAUNCH();
    return RT_OK;
}

extern "C" int rt_sgd_flat(const rt_adamw_desc* d, rt_stream_t stream) {
    if (!d || !d->p || (!d->g && !d->g16) || !d->m || d->n <= 0 || (d->n & 3) || d->n_ranges < 1 || d->n_ranges > 8) return RT_ERR_BADARG;
    for (int r = 0; r < d->n_ranges; ++r) if ((d->range_begin[r] & 3) || (d->range_end[r] & 3)) return RT_ERR_BADARG;
    rt_adamw_desc a = *d;
    if (a.span_begin == 0 && a.span_end == 0) a.span_end = a.n;
    if (a.span_begin < 0 || a.span_end > a.n || a.span_begin >= a.span_end || (a.span_begin & 3) || (a.span_end & 3)) return RT_ERR_BADARG;
    int blocks = (int)(((size_t)(a.span_end - a.span_begin) / 4 + 255) / 256); if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(sgd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    RT_CHECK_LAUNCH();
    return RT_OK;
}
