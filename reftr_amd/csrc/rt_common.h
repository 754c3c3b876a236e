// Shared device helpers for the RefTR gfx950 kernels (CDNA4, wave64).
// All kernels in this directory are written for MI355X only: 64-lane waves,
// MFMA 16x16x32 bf16 fragments, 160 KiB LDS per CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/reftr_hip.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define RT_WAVE 64

// Tuning switches of the A/B sessions recorded in LAB_NOTES.md.  The product library fixes every one at its measured-best value
// (the second argument); the lab library (REFTR_LAB=1 at build and at import: -DRT_LAB, libreftr_hip_lab.so) reads them from the
// environment, once, at the first launch that consults them.
#ifdef RT_LAB
#include <stdlib.h>
#define RT_TUNE(name, dflt) (getenv(name) ? atoi(getenv(name)) : (dflt))
#define RT_TUNE_SET(name) (getenv(name) != nullptr)
#else
#define RT_TUNE(name, dflt) (dflt)
#define RT_TUNE_SET(name) false
#endif

__device__ __forceinline__ float rt_bf2f(bf16_t v) { return (float)v; }
__device__ __forceinline__ bf16_t rt_f2bf(float v) { return (bf16_t)v; }

// Counter-based dropout hash: keep element `idx` of site `seed` iff hash >= thresh
// (thresh = p * 2^32).  Forward and backward regenerate the same mask from
// (seed, idx); the oracle restates the same integer arithmetic in numpy.
__device__ __forceinline__ uint32_t rt_hash32(uint32_t seed, uint32_t idx) {
    uint32_t x = idx * 0x9E3779B1u ^ seed;
    x ^= x >> 16; x *= 0x85EBCA6Bu;
    x ^= x >> 13; x *= 0xC2B2AE35u;
    x ^= x >> 16;
    return x;
}
// effective seed of a dropout site when the step seed lives in device memory (captured graphs)
__device__ __forceinline__ uint32_t rt_site_seed(const uint32_t* seed_dev, uint32_t site) {
    return seed_dev ? rt_hash32(*seed_dev, site) : site;
}
__device__ __forceinline__ uint32_t rt_drop_thresh(float p) {
    return (uint32_t)((double)p * 4294967296.0);
}

__device__ __forceinline__ float rt_gelu(float x) {
    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
__device__ __forceinline__ float rt_gelu_grad(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}

__device__ __forceinline__ float rt_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float rt_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Block-wide sum for blockDim.x <= 1024 (multiple of 64); `sm` holds >= 16 floats.
__device__ __forceinline__ float rt_block_sum(float v, float* sm) {
    v = rt_wave_sum(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) sm[wid] = v;
    __syncthreads();
    float r = 0.f;
    for (int i = 0; i < nw; ++i) r += sm[i];
    return r;
}

// Small fp32 workspaces are cleared with a KERNEL, not hipMemsetAsync: inside a captured hipGraph (ROCm 7.2) memset
// nodes were observed to race with the kernel nodes that follow them (intermittent garbage statistics / losses),
// while kernel -> kernel ordering on the captured stream is reliable.
static __global__ void rt_zero_f32_kernel(float* p, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = 0.f;
}
static inline hipError_t rt_zero_f32(float* p, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(rt_zero_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, (int)n);
    return hipGetLastError();
}

// XCD-aware workgroup -> work-item map.  Workgroups are handed to the 8 XCDs round-robin by linear id, and each XCD has
// its own L2: with the identity map, neighbouring tiles (which share operand rows) land on 8 different L2s and every
// shared row is fetched over the fabric up to 8 times.  This map gives XCD x the contiguous range of items
// [x*n/8, (x+1)*n/8), so tiles that share rows share an L2.  `xcd_on` = 0 keeps the identity map (A/B switch).
__device__ __forceinline__ int rt_xcd_remap(int b, int n, int xcd_on) {
    if (!xcd_on || n < 16) return b;
    const int xcd = b & 7, idx = b >> 3, per = n >> 3, rem = n & 7;
    return xcd * per + (xcd < rem ? xcd : rem) + idx;
}

// Gradient-norm accumulator (round 4): the squared L2 norm of the weight gradients is collected WHERE THEY ARE PRODUCED -- every
// epilogue that assigns or accumulates a piece of a weight gradient adds (new^2 - old^2) of that piece -- instead of by a second
// pass over the 607 MB gradient buffer (engine_vg.py:62-63's clip_grad_norm_).  Contributions go to one of RT_SQ_SLOTS fp32 words
// (a cache line apart, picked by workgroup / wave id) with fire-and-forget atomics; rt_sqnorm_finish adds the slots up.
__device__ __forceinline__ void rt_sq_add(float* slots, unsigned who, float v) {
    atomicAdd(slots + (size_t)(who & (RT_SQ_SLOTS - 1)) * RT_SQ_STRIDE, v);
}
// sign * |buf|^2 of up to 32 fp32 buffers into the slots (the producers without an in-kernel contribution: a pass with sign -1
// in front of an accumulating launch, +1 behind every launch); rt_optim.hip
int rt_sq_pass(float* const* bufs, const long long* counts, const float* signs, int n, float* slots, hipStream_t s);
// twin[i] = bf16(buf[i]) for up to any number of fp32 buffers (the bf16 exchange twins of weight gradients whose producer has no
// in-kernel twin store); rt_optim.hip
int rt_round_pass(float* const* bufs, void* const* twins, const long long* counts, int n, hipStream_t s);

#define RT_CHECK_LAUNCH() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int)e_; } while (0)
