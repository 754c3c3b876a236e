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

// ------------------------------------------------------------------------------------------------
// Matrix-aware AdamW (round 4): the same update, walked per WEIGHT MATRIX in 64(n) x 64(c) tiles per tap, so that the kernel that
// produces the new fp32 master also emits the bf16 GEMM operands the next forward / backward read -- W [N][T][C] (x FrozenBN scale)
// and its transpose [C][T][N] -- while the new values are still in registers.  The separate operand refresh (rt_weight_prep_batched:
// re-reads 4 B per parameter, one more dependent launch at the head of the step) disappears; arithmetic and rounding points are
// those of adamw_kernel followed by weight_prep_batched_kernel, bit for bit.  Everything that is not a matrix job (biases, norm
// parameters, embeddings) is updated by adamw_chunks_kernel over a static chunk table (the complement of the jobs).
struct AdamCoef { float gs, bc1, inv_sqrt_bc2; };
__device__ __forceinline__ AdamCoef adam_coef(const rt_adamw_desc& p) {
    const float total = sqrtf(p.gnorm_sq ? p.gnorm_sq[0] : 0.f) * p.grad_scale;
    float coef = 1.f;
    if (p.max_norm > 0.f) coef = fminf(1.f, p.max_norm / (total + 1e-6f));
    if (blockIdx.x == 0 && threadIdx.x == 0 && p.gnorm_out) p.gnorm_out[0] = total;
    const int step = p.step_dev ? p.step_dev[0] : p.step;
    AdamCoef c;
    c.gs = p.grad_scale * coef;
    c.bc1 = 1.f - powf(p.beta1, (float)step);
    c.inv_sqrt_bc2 = rsqrtf(1.f - powf(p.beta2, (float)step));
    return c;
}
__device__ __forceinline__ void adam_range(const rt_adamw_desc& p, size_t e, float& lr, float& wd) {
    lr = 0.f; wd = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r)
        if (r < p.n_ranges && e >= (size_t)p.range_begin[r] && e < (size_t)p.range_end[r]) {
            lr = p.lr_dev ? p.lr_dev[r] : p.range_lr[r]; wd = p.range_wd[r];
        }
}
// the update of one element, written exactly as in adamw_kernel (same operation order: the two kernels must round alike)
__device__ __forceinline__ void adam_elem(const rt_adamw_desc& p, const AdamCoef& c, float lr, float wd, float graw, float& pv, float& mv, float& vv) {
    const float g = graw * c.gs;
    pv *= (1.f - lr * wd);
    mv = p.beta1 * mv + (1.f - p.beta1) * g;
    vv = p.beta2 * vv + (1.f - p.beta2) * g * g;
    const float denom = sqrtf(vv) * c.inv_sqrt_bc2 + p.eps;
    pv -= (lr / c.bc1) * (mv / denom);
}

// table: int64 [njobs][8] = {element offset of the matrix in p/g/m/v, scale ptr | 0, dst ptr | 0, dst_t ptr | 0, N, T, C, first tile}
// A matrix [N][T][C] is walked as the 2-D array [N][K'] (K' = T * C: a row is contiguous) in tiles of 32 rows x 256 columns: a wave
// reads / writes ONE KB-contiguous row piece per instruction in each of p, g, m, v (the first version used 64 x 64 tiles -- 256-byte
// pieces from four arrays -- and ran at 3.2-4.1 TB/s against the flat pass's 6: a DRAM page was opened for a quarter of its bytes).
// The bf16 copy W goes out in the same pass (512 contiguous bytes per wave); the transposed copy [C][T][N] is staged through LDS and
// written by one thread per column as the 64 contiguous bytes of its 32 rows.
constexpr int AM_TN = 32, AM_TK = 256, AM_LD = AM_TK + 4;
__global__ __launch_bounds__(256) void adamw_mat_kernel(const rt_adamw_desc p, const int64_t* __restrict__ table, int njobs) {
    if (p.active && p.active[0] == 0) return;
    __shared__ __attribute__((aligned(16))) bf16_t tile[AM_TN][AM_LD];
    const AdamCoef cf = adam_coef(p);
    int lo = 0, hi = njobs - 1;                       // last job whose first_tile <= blockIdx.x
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid * 8 + 7] <= (int64_t)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const int64_t* j = table + lo * 8;
    const size_t base = (size_t)j[0];
    bf16_t* dst = reinterpret_cast<bf16_t*>(j[2]);
    bf16_t* dst_t = reinterpret_cast<bf16_t*>(j[3]);
    // Sparse-state jobs (round 4): a matrix WITHOUT bf16 operands (the embedding tables: 24 M of BERT's parameters, of which a step
    // touches <= B * L rows) may carry, in the otherwise unused scale slot, one byte per KB row piece: 0 = "m and v of this piece
    // are all zero".  A piece whose gradient is all zero too is then left alone WITHOUT reading p / m / v -- bit-exact, because with
    // g = m = v = 0 the update is p *= (1 - lr * wd) and nothing else, and the skip is only taken when that factor rounds to 1.0f
    // (lr_bert * wd = 1e-9 in every reference config: models the reference's own fp32 no-op).  4 B instead of 32 B per element.
    uint8_t* flags = (!dst && !dst_t && j[1]) ? reinterpret_cast<uint8_t*>(j[1]) : nullptr;
    const float* scale = flags ? nullptr : reinterpret_cast<const float*>(j[1]);
    const int N = (int)j[4], T = (int)j[5], C = (int)j[6];
    const int K = T * C;
    const int local = blockIdx.x - (int)j[7];
    const int kt = (K + AM_TK - 1) / AM_TK;
    const int tk = local % kt, tn = local / kt;        // column tiles fastest: neighbouring workgroups stream neighbouring KBs of the same rows
    const int n0 = tn * AM_TN, k0 = tk * AM_TK;
    float lr, wd;
    adam_range(p, base, lr, wd);                      // a matrix lies inside one learning-rate range
    float* P = p.p + base; float* Mo = p.m + base; float* Vo = p.v + base;
    const float* G = p.g ? p.g + base : nullptr;
    const bf16_t* G16 = p.g16 ? reinterpret_cast<const bf16_t*>(p.g16) + base : nullptr;
    typedef bf16_t bf16x4_t __attribute__((ext_vector_type(4)));
    const int t = threadIdx.x;
    const bool vec = (K & 3) == 0 && (base & 3) == 0 && (!dst || ((uintptr_t)dst & 7) == 0);
    if (vec) {
        const int r = t >> 6, q = (t & 63) * 4, k = k0 + q;
        const bool may_skip = flags && (1.f - lr * wd) == 1.f;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            f32x4 pv[4], gv[4], mv[4], vv[4];
            bool ok[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int n = n0 + (half * 4 + jj) * 4 + r;
                ok[jj] = n < N && k < K;
                if (ok[jj]) {
                    const size_t o = (size_t)n * K + k;
                    if (G16) {
                        const bf16x4_t h = __builtin_nontemporal_load(reinterpret_cast<const bf16x4_t*>(G16 + o));
                        gv[jj] = f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
                    } else gv[jj] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(G + o));
                }
            }
            if (flags) {
                // a wave = one KB row piece (row n, column tile tk): the decisions below are wave-uniform
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int n = n0 + (half * 4 + jj) * 4 + r;
                    const bool nz = ok[jj] && (gv[jj][0] != 0.f || gv[jj][1] != 0.f || gv[jj][2] != 0.f || gv[jj][3] != 0.f);
                    const bool g_zero = __ballot(nz) == 0;
                    if (n < N && may_skip && g_zero && flags[(size_t)n * kt + tk] == 0) ok[jj] = false;      // nothing to do for this piece
                }
            }
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int n = n0 + (half * 4 + jj) * 4 + r;
                if (ok[jj]) {
                    const size_t o = (size_t)n * K + k;
                    pv[jj] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(P + o));
                    mv[jj] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(Mo + o));
                    vv[jj] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(Vo + o));
                }
            }
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int nl = (half * 4 + jj) * 4 + r, n = n0 + nl;
                bf16x4_t ov = bf16x4_t{(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
                if (ok[jj]) {
                    const size_t o = (size_t)n * K + k;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { float a = pv[jj][e], b = mv[jj][e], d = vv[jj][e]; adam_elem(p, cf, lr, wd, gv[jj][e], a, b, d); pv[jj][e] = a; mv[jj][e] = b; vv[jj][e] = d; }
                    *reinterpret_cast<f32x4*>(P + o) = pv[jj];
                    __builtin_nontemporal_store(mv[jj], reinterpret_cast<f32x4*>(Mo + o));
                    __builtin_nontemporal_store(vv[jj], reinterpret_cast<f32x4*>(Vo + o));
                    if (flags) {                       // the state of this piece after the update (wave-uniform: ok[jj] is, within a row piece, except past K)
                        bool live = false;
#pragma unroll
                        for (int e = 0; e < 4; ++e) live = live || mv[jj][e] != 0.f || vv[jj][e] != 0.f;
                        const bool any_live = __ballot(live) != 0;
                        if ((t & 63) == 0) flags[(size_t)n * kt + tk] = any_live ? 1 : 0;
                    }
                    const float sc = scale ? scale[n] : 1.f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) ov[e] = (bf16_t)(scale ? pv[jj][e] * sc : pv[jj][e]);
                    if (dst) *reinterpret_cast<bf16x4_t*>(dst + o) = ov;
                }
                if (dst_t) *reinterpret_cast<bf16x4_t*>(&tile[nl][q]) = ov;
            }
        }
    } else {                                          // ragged / misaligned matrices: element-wise over the same tile
        for (int idx = t; idx < AM_TN * AM_TK; idx += 256) {
            const int nl = idx / AM_TK, kk = idx - nl * AM_TK, n = n0 + nl, k = k0 + kk;
            float w = 0.f;
            if (n < N && k < K) {
                const size_t o = (size_t)n * K + k;
                float pv = P[o], mv = Mo[o], vv = Vo[o];
                const float gv = G16 ? (float)G16[o] : G[o];
                adam_elem(p, cf, lr, wd, gv, pv, mv, vv);
                P[o] = pv; Mo[o] = mv; Vo[o] = vv;
                w = scale ? pv * scale[n] : pv;
                if (dst) dst[o] = (bf16_t)w;
            }
            if (dst_t) tile[nl][kk] = (bf16_t)w;
        }
    }
    if (!dst_t) return;
    __syncthreads();
    const int k = k0 + t;                              // one column of the tile per thread: 32 rows = 64 contiguous bytes of dst_t
    if (k >= K) return;
    const int c = k % C, tap = k / C;
    bf16_t* out = dst_t + ((size_t)c * T + tap) * N + n0;
    const bool vec_n = (N & 7) == 0 && ((uintptr_t)dst_t & 15) == 0;
#pragma unroll
    for (int g8 = 0; g8 < AM_TN / 8; ++g8) {
        if (n0 + g8 * 8 >= N) break;
        if (vec_n) {                                   // N % 8 == 0: the 8 rows exist
            bf16x8 ov;
#pragma unroll
            for (int e = 0; e < 8; ++e) ov[e] = tile[g8 * 8 + e][t];
            *reinterpret_cast<bf16x8*>(out + g8 * 8) = ov;
        } else {
            for (int e = 0; e < 8 && n0 + g8 * 8 + e < N; ++e) out[g8 * 8 + e] = tile[g8 * 8 + e][t];
        }
    }
}

// everything that is not a matrix job: table = nchunks x {int64 element offset (multiple of 4), int64 count (multiple of 4, <= 16384)}
__global__ __launch_bounds__(256) void adamw_chunks_kernel(const rt_adamw_desc p, const int64_t* __restrict__ table) {
    if (p.active && p.active[0] == 0) return;
    const AdamCoef cf = adam_coef(p);
    const size_t off = (size_t)table[2 * blockIdx.x], cnt = (size_t)table[2 * blockIdx.x + 1];
    typedef bf16_t bf16x4_t __attribute__((ext_vector_type(4)));
    for (size_t i = threadIdx.x * 4; i < cnt; i += 1024) {
        const size_t e = off + i;
        float lr, wd;
        adam_range(p, e, lr, wd);
        f32x4 pv = *reinterpret_cast<const f32x4*>(p.p + e), mv = *reinterpret_cast<const f32x4*>(p.m + e), vv = *reinterpret_cast<const f32x4*>(p.v + e), gv;
        if (p.g16) { const bf16x4_t h = *reinterpret_cast<const bf16x4_t*>(reinterpret_cast<const bf16_t*>(p.g16) + e); gv = f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]}; }
        else gv = *reinterpret_cast<const f32x4*>(p.g + e);
#pragma unroll
        for (int c = 0; c < 4; ++c) { float a = pv[c], b = mv[c], d = vv[c]; adam_elem(p, cf, lr, wd, gv[c], a, b, d); pv[c] = a; mv[c] = b; vv[c] = d; }
        *reinterpret_cast<f32x4*>(p.p + e) = pv; *reinterpret_cast<f32x4*>(p.m + e) = mv; *reinterpret_cast<f32x4*>(p.v + e) = vv;
    }
}

// ---- gradient-norm accumulator (rt_common.h): passes over whole buffers / chunk tables, and the final sum of the slots
struct SqList { const float* buf[32]; long long cnt[32]; float sign[32]; int first[33]; int n; };
__global__ __launch_bounds__(256) void sq_list_kernel(const SqList l, float* __restrict__ slots) {
    __shared__ float sm[16];
    int lo = 0, hi = l.n - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (l.first[mid] <= (int)blockIdx.x) lo = mid; else hi = mid - 1; }
    const float* g = l.buf[lo];
    const size_t n = (size_t)l.cnt[lo];
    const size_t b0 = (size_t)((int)blockIdx.x - l.first[lo]) * 4096;              // 4096 elements per workgroup
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const size_t i = b0 + (size_t)j * 1024 + threadIdx.x * 4;
        if (i + 4 <= n) { const f32x4 v = *reinterpret_cast<const f32x4*>(g + i); s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]; }
        else for (size_t k = i; k < n; ++k) s += g[k] * g[k];
    }
    s = rt_block_sum(s, sm);
    if (threadIdx.x == 0) rt_sq_add(slots, blockIdx.x, l.sign[lo] * s);
}
// the tensors that are accumulated with atomics (the complement of the weight matrices): table = n x {element offset, count <= 16384}
__global__ __launch_bounds__(256) void sq_chunks_kernel(const float* __restrict__ base, const int64_t* __restrict__ table, float* __restrict__ slots) {
    __shared__ float sm[16];
    const size_t off = (size_t)table[2 * blockIdx.x], cnt = (size_t)table[2 * blockIdx.x + 1];
    float s = 0.f;
    for (size_t i = threadIdx.x * 4; i < cnt; i += 1024) {
        if (i + 4 <= cnt && ((off + i) & 3) == 0) { const f32x4 v = *reinterpret_cast<const f32x4*>(base + off + i); s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]; }
        else for (size_t k = i; k < cnt && k < i + 4; ++k) s += base[off + k] * base[off + k];
    }
    s = rt_block_sum(s, sm);
    if (threadIdx.x == 0) rt_sq_add(slots, blockIdx.x, s);
}
__global__ __launch_bounds__(256) void sq_sum_kernel(const float* __restrict__ slots, float* __restrict__ out, const float* __restrict__ extra) {
    __shared__ float sm[16];
    float s = slots[(size_t)threadIdx.x * RT_SQ_STRIDE];
    s = rt_block_sum(s, sm);
    if (threadIdx.x == 0) out[0] = fmaxf(s, 0.f) + (extra ? extra[0] : 0.f);      // (new^2 - old^2 terms may leave -1 ulp when everything is zero)
}

struct RoundList { const float* buf[32]; bf16_t* twin[32]; long long cnt[32]; int first[33]; int n; };
__global__ __launch_bounds__(256) void round_list_kernel(const RoundList l) {
    int lo = 0, hi = l.n - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (l.first[mid] <= (int)blockIdx.x) lo = mid; else hi = mid - 1; }
    const float* g = l.buf[lo]; bf16_t* tw = l.twin[lo];
    const size_t n = (size_t)l.cnt[lo];
    const size_t b0 = (size_t)((int)blockIdx.x - l.first[lo]) * 4096;
    typedef bf16_t bf16x4_t __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const size_t i = b0 + (size_t)j * 1024 + threadIdx.x * 4;
        if (i + 4 <= n) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(g + i);
            *reinterpret_cast<bf16x4_t*>(tw + i) = bf16x4_t{(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
        } else for (size_t k = i; k < n; ++k) tw[k] = (bf16_t)g[k];
    }
}
__global__ __launch_bounds__(256) void round_chunks_kernel(const float* __restrict__ base, bf16_t* __restrict__ twin, const int64_t* __restrict__ table) {
    const size_t off = (size_t)table[2 * blockIdx.x], cnt = (size_t)table[2 * blockIdx.x + 1];
    typedef bf16_t bf16x4_t __attribute__((ext_vector_type(4)));
    for (size_t i = threadIdx.x * 4; i < cnt; i += 1024) {
        if (i + 4 <= cnt && ((off + i) & 3) == 0) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(base + off + i);
            *reinterpret_cast<bf16x4_t*>(twin + off + i) = bf16x4_t{(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
        } else for (size_t k = i; k < cnt && k < i + 4; ++k) twin[off + k] = (bf16_t)base[off + k];
    }
}

__global__ void counter_add_if_zero_kernel(int32_t* c, int32_t inc, const uint32_t* cond, int reset_else) {
    if (cond[0] == 0u) c[0] += inc; else if (reset_else) c[0] = 0;
}
__global__ void stamp_kernel(uint64_t* buf, int idx) { buf[idx] = wall_clock64(); }
// the end of an iteration in deferred mode, decided on the device: the update is armed (and the step counter advanced) only when no
// cooperative launch failed AND the iteration's total loss is finite
__global__ void finish_step_kernel(int32_t* step, int32_t* active, const uint32_t* cond, const float* loss) {
    const bool bad = (cond && cond[0] != 0u) || (loss && !isfinite(loss[0]));
    if (!bad) { step[0] += 1; active[0] += 1; } else active[0] = 0;
}

__global__ void finish_stats_kernel(const rt_finish_desc d) {
    const uint32_t cw = d.cond ? d.cond[0] : 0u;
    const bool bad = cw != 0u || (d.loss && !isfinite(d.loss[0]));
    if (!bad) { d.step[0] += 1; d.active[0] += 1; } else d.active[0] = 0;
    float gn = d.grad_norm ? d.grad_norm[0] : 0.f;
    if (d.sq) { gn = sqrtf(d.sq[0]) * d.norm_scale; if (d.grad_norm) d.grad_norm[0] = gn; }
    if (d.stats) {
        int o = 0;
        for (int i = 0; i < d.n_src; ++i) d.stats[o++] = d.src[i][0];
        if (d.cond_in_stats) d.stats[o++] = (float)cw;
        d.stats[o] = gn;
    }
}

}  // namespace

extern "C" int rt_counter_add(int32_t* ctr, int32_t inc, rt_stream_t stream) {
    if (!ctr) return RT_ERR_BADARG;
    hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, ctr, inc);
    RT_CHECK_LAUNCH();
    return RT_OK;
}


extern "C" int rt_counter_add_if_zero(int32_t* ctr, int32_t inc, const uint32_t* cond, int reset_else, rt_stream_t stream) {
    if (!ctr || !cond) return RT_ERR_BADARG;
    hipLaunchKernelGGL(counter_add_if_zero_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, ctr, inc, cond, reset_else);
    RT_CHECK_LAUNCH();
    return RT_OK;
}

extern "C" int rt_finish_step(int32_t* step, int32_t* active, const uint32_t* cond, const float* loss, rt_stream_t stream) {
    if (!step || !active) return RT_ERR_BADARG;
    hipLaunchKernelGGL(finish_step_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step, active, cond, loss);
    RT_CHECK_LAUNCH();
    return RT_OK;
}

extern "C" int rt_finish_stats(const rt_finish_desc* d, rt_stream_t stream) {
    if (!d || !d->step || !d->active || d->n_src < 0 || d->n_src > RT_STATS_MAX) return RT_ERR_BADARG;
    for (int i = 0; i < d->n_src; ++i) if (!d->src[i]) return RT_ERR_BADARG;
    hipLaunchKernelGGL(finish_stats_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, *d);
    RT_CHECK_LAUNCH();
    return RT_OK;
}

extern "C" int rt_stamp(uint64_t* buf, int idx, rt_stream_t stream) {
    if (!buf || idx < 0) return RT_ERR_BADARG;
    hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, buf, idx);
    RT_CHECK_LAUNCH();
    return RT_OK;
}

static int adamw_desc_ok(const rt_adamw_desc* d) {
    if (!d || !d->p || (!d->g && !d->g16) || !d->m || !d->v || d->n <= 0 || (d->n & 3) || d->n_ranges < 1 || d->n_ranges > 8 || (d->step < 1 && !d->step_dev))
        return RT_ERR_BADARG;
    for (int r = 0; r < d->n_ranges; ++r) if ((d->range_begin[r] & 3) || (d->range_end[r] & 3)) return RT_ERR_BADARG;
    return RT_OK;
}

extern "C" int rt_adamw_mat(const rt_adamw_desc* d, const int64_t* table, int njobs, int total_tiles, rt_stream_t stream) {
    const int rc = adamw_desc_ok(d);
    if (rc != RT_OK) return rc;
    if (!table || njobs <= 0 || total_tiles <= 0) return RT_ERR_BADARG;
    hipLaunchKernelGGL(adamw_mat_kernel, dim3((unsigned)total_tiles), dim3(256), 0, (hipStream_t)stream, *d, table, njobs);
    RT_CHECK_LAUNCH();
    return RT_OK;
}

extern "C" int rt_adamw_chunks(const rt_adamw_desc* d, const int64_t* table, int nchunks, rt_stream_t stream) {
    const int rc = adamw_desc_ok(d);
    if (rc != RT_OK) return rc;
    if (!table || nchunks <= 0) return RT_ERR_BADARG;
    hipLaunchKernelGGL(adamw_chunks_kernel, dim3((unsigned)nchunks), dim3(256), 0, (hipStream_t)stream, *d, table);
    RT_CHECK_LAUNCH();
    return RT_OK;
}

int rt_sq_pass(float* const* bufs, const long long* counts, const float* signs, int n, float* slots, hipStream_t s) {
    for (int base = 0; base < n; base += 32) {
        SqList l; l.n = 0; int blocks = 0;
        for (int i = base; i < n && i < base + 32; ++i) {
            if (!bufs[i] || counts[i] <= 0) continue;
            l.buf[l.n] = bufs[i]; l.cnt[l.n] = counts[i]; l.sign[l.n] = signs[i]; l.first[l.n] = blocks; ++l.n;
            blocks += (int)((counts[i] + 4095) / 4096);
        }
        if (!l.n) continue;
        l.first[l.n] = blocks;
        hipLaunchKernelGGL(sq_list_kernel, dim3((unsigned)blocks), dim3(256), 0, s, l, slots);
        RT_CHECK_LAUNCH();
    }
    return RT_OK;
}

int rt_round_pass(float* const* bufs, void* const* twins, const long long* counts, int n, hipStream_t s) {
    for (int base = 0; base < n; base += 32) {
        RoundList l; l.n = 0; int blocks = 0;
        for (int i = base; i < n && i < base + 32; ++i) {
            if (!bufs[i] || !twins[i] || counts[i] <= 0) continue;
            l.buf[l.n] = bufs[i]; l.twin[l.n] = (bf16_t*)twins[i]; l.cnt[l.n] = counts[i]; l.first[l.n] = blocks; ++l.n;
            blocks += (int)((counts[i] + 4095) / 4096);
        }
        if (!l.n) continue;
        l.first[l.n] = blocks;
        hipLaunchKernelGGL(round_list_kernel, dim3((unsigned)blocks), dim3(256), 0, s, l);
        RT_CHECK_LAUNCH();
    }
    return RT_OK;
}

extern "C" int rt_round_chunks(const float* base, void* twin, const int64_t* table, int n, rt_stream_t stream) {
    if (!base || !twin || !table || n <= 0) return RT_ERR_BADARG;
    hipLaunchKernelGGL(round_chunks_kernel, dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream, base, (bf16_t*)twin, table);
    RT_CHECK_LAUNCH();
    return RT_OK;
}

extern "C" int rt_sqnorm_finish(const float* base, const int64_t* table, int nchunks, float* slots, const float* extra, float* out,
                                rt_stream_t stream) {
    if (!slots || !out || (nchunks > 0 && (!base || !table))) return RT_ERR_BADARG;
    hipStream_t s = (hipStream_t)stream;
    if (nchunks > 0) {
        hipLaunchKernelGGL(sq_chunks_kernel, dim3((unsigned)nchunks), dim3(256), 0, s, base, table, slots);
        RT_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(sq_sum_kernel, dim3(1), dim3(256), 0, s, slots, out, extra);
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
    static const int nt_env = RT_TUNE("REFTR_ADAMW_NT", 1);
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
