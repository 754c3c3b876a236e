// Small memory-bound kernels around the GEMMs: bias gradients, row gather / scatter-add, fused adds,
// BERT embedding lookup (+ backward scatter), QueryEncoder's CLS-key attention, context-mask construction.
#include "rt_common.h"

namespace {

// grp_rows > 0: (r / g) * stride + off + r % g;  grp_rows < 0: broadcast (r / -g) * stride + off;  0: identity
__device__ __forceinline__ int map_row(int r, int grp_rows, int grp_stride, int grp_off) {
    if (grp_rows > 0) return (r / grp_rows) * grp_stride + grp_off + (r % grp_rows);
    if (grp_rows < 0) return (r / (-grp_rows)) * grp_stride + grp_off;
    return r;
}

// ---------------------------------------------------------------- column sums (bias gradients)
// db[n] += sum_m dy[m, n]; block = 64 columns x 4 row-lanes, grid.y splits the rows.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ dy, float* __restrict__ db, int M, int N,
                                                     int rows_per_block) {
    __shared__ float sm[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rl = threadIdx.x >> 6;
    const int r0 = blockIdx.y * rows_per_block;
    const int r1 = min(r0 + rows_per_block, M);
    float s = 0.f;
    if (c < N)
        for (int r = r0 + rl; r < r1; r += 4) s += (float)dy[(size_t)r * N + c];
    sm[rl][threadIdx.x & 63] = s;
    __syncthreads();
    if (rl == 0 && c < N) atomicAdd(db + c, sm[0][threadIdx.x] + sm[1][threadIdx.x] + sm[2][threadIdx.x] + sm[3][threadIdx.x]);
}

// ---------------------------------------------------------------- fused elementwise add / cast with row maps
// out[map_o(r)] = a[map_a(r)] + b[map_b(r)] (b optional), written as fp32 and/or bf16; accumulate: out_f32 +=
__global__ __launch_bounds__(256) void rows_add_kernel(const rt_rows_add_desc p) {
    const size_t total = (size_t)p.rows * p.D;
    bf16_t* ob = (bf16_t*)p.out_bf16;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % p.D);
        const int r = (int)(i / p.D);
        float v = 0.f;
        if (p.a_f32) v += p.a_f32[(size_t)map_row(r, p.a_grp_rows, p.a_grp_stride, p.a_grp_off) * p.D + c];
        if (p.a_bf16) v += (float)((const bf16_t*)p.a_bf16)[(size_t)map_row(r, p.a_grp_rows, p.a_grp_stride, p.a_grp_off) * p.D + c];
        if (p.b_f32) v += p.b_f32[(size_t)map_row(r, p.b_grp_rows, p.b_grp_stride, p.b_grp_off) * p.D + c];
        v *= p.alpha;
        const size_t o = (size_t)map_row(r, p.o_grp_rows, p.o_grp_stride, p.o_grp_off) * p.D + c;
        if (p.out_f32) {
            if (p.accumulate == 2) atomicAdd(p.out_f32 + o, v);
            else if (p.accumulate) p.out_f32[o] += v;
            else p.out_f32[o] = v;
        }
        if (ob) ob[o] = (bf16_t)v;
    }
}

// ---------------------------------------------------------------- BERT embeddings
// e[r, :] = word[ids[r]] + pos[r % L] + type[0]   (HF BertEmbeddings)
// RoBERTa (HF create_position_ids_from_input_ids): pos = cumsum(ids != pad) * (ids != pad) + pad -- integer, exact
__global__ void roberta_pos_ids_kernel(const int64_t* __restrict__ ids, int32_t* __restrict__ pos_ids, int B, int L, int pad) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    int run = 0;
    for (int l = 0; l < L; ++l) {
        const bool tok = ids[(size_t)b * L + l] != pad;
        run += tok ? 1 : 0;
        pos_ids[(size_t)b * L + l] = tok ? run + pad : pad;
    }
}
__global__ __launch_bounds__(256) void bert_embed_kernel(const int64_t* __restrict__ ids, const float* __restrict__ word,
                                                         const float* __restrict__ pos, const float* __restrict__ type0,
                                                         float* __restrict__ out, int rows, int L, int D,
                                                         const int32_t* __restrict__ pos_ids) {
    const size_t total = (size_t)rows * D;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % D);
        const int r = (int)(i / D);
        const int pi = pos_ids ? pos_ids[r] : r % L;
        out[i] = word[(size_t)ids[r] * D + c] + pos[(size_t)pi * D + c] + type0[c];
    }
}
__global__ __launch_bounds__(256) void bert_embed_bwd_kernel(const int64_t* __restrict__ ids, const float* __restrict__ de,
                                                             float* __restrict__ dword, float* __restrict__ dpos,
                                                             float* __restrict__ dtype0, int rows, int L, int D,
                                                             const int32_t* __restrict__ pos_ids) {
    const size_t total = (size_t)rows * D;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % D);
        const int r = (int)(i / D);
        const float g = de[i];
        atomicAdd(dword + (size_t)ids[r] * D + c, g);
        atomicAdd(dpos + (size_t)(pos_ids ? pos_ids[r] : r % L) * D + c, g);
        atomicAdd(dtype0 + c, g);
    }
}

// ---------------------------------------------------------------- context masks (integer / bool only)
// single phrase (models/reftr_transformer.py:239-248): ctx[b,0,l] = !smask[b,l] | l==0 | l==len_b-1
// multi phrase  (:224-238): ctx[b,j,l] = !(pos_l[b,j] <= l < pos_r[b,j]); qmask[b,j] = !phrase_mask[b,j,2]
__global__ void context_mask_kernel(const uint8_t* smask, const uint8_t* phrase_mask, const int64_t* pos_l,
                                    const int64_t* pos_r, uint8_t* ctx, uint8_t* qmask, int B, int L, int P, int Lp) {
    const int b = blockIdx.x;
    if (!phrase_mask) {
        int len = 0;
        for (int l = 0; l < L; ++l) len += smask[b * L + l] ? 1 : 0;
        for (int l = threadIdx.x; l < L; l += blockDim.x)
            ctx[b * L + l] = (!smask[b * L + l] || l == 0 || l == len - 1) ? 1 : 0;
        if (threadIdx.x == 0) qmask[b] = 0;
    } else {
        for (int i = threadIdx.x; i < P * L; i += blockDim.x) {
            const int j = i / L, l = i % L;
            ctx[((size_t)b * P + j) * L + l] = (l >= pos_l[b * P + j] && l < pos_r[b * P + j]) ? 0 : 1;
        }
        for (int j = threadIdx.x; j < P; j += blockDim.x)
            qmask[b * P + j] = phrase_mask[((size_t)b * P + j) * Lp + 2] ? 0 : 1;
    }
}

// ---------------------------------------------------------------- QueryEncoder attention
// models/reftr_transformer.py:48-55: w[b,j,:] = softmax_l(k[b] . qs[b,l] masked by ctx[b,j,l]) (NO 1/sqrt(d)),
// c[b,j,:] = sum_l w[b,j,l] vs[b,l,:].   One block per (b, j); E <= 256, L <= 128.
__global__ __launch_bounds__(1024) void qenc_attn_fwd_kernel(const float* __restrict__ k, const float* __restrict__ qs,
                                                             const float* __restrict__ vs, const uint8_t* __restrict__ ctx,
                                                             float* __restrict__ wout, float* __restrict__ cout,
                                                             int P, int L, int E) {
    // 16 waves: the L score rows go round-robin over the waves (one 1-KB row load + a wave reduction each), the weighted sum
    // over tokens is split four ways over the waves' quarters and met in LDS -- the 8-workgroup launch is pure latency
    __shared__ float sw[128];
    __shared__ float red[32];
    __shared__ float part[4][256];
    const int b = blockIdx.x / P, j = blockIdx.x % P;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int l = wave; l < L; l += 16) {
        float s = 0.f;
        for (int d = lane; d < E; d += 64) s += k[(size_t)b * E + d] * qs[((size_t)b * L + l) * E + d];
        s = rt_wave_sum(s);
        if (lane == 0) sw[l] = ctx[((size_t)b * P + j) * L + l] ? -INFINITY : s;
    }
    __syncthreads();
    float m = -INFINITY;
    for (int l = threadIdx.x; l < L; l += 1024) m = fmaxf(m, sw[l]);
    m = rt_wave_max(m);
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = red[0];
#pragma unroll
    for (int w = 1; w < 16; ++w) m = fmaxf(m, red[w]);
    __syncthreads();
    float e = 0.f;
    for (int l = threadIdx.x; l < L; l += 1024) { const float x = __expf(sw[l] - m); sw[l] = x; e += x; }
    e = rt_wave_sum(e);
    if (lane == 0) red[16 + wave] = e;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) tot += red[16 + w];
    const float inv = 1.f / tot;
    __syncthreads();
    for (int l = threadIdx.x; l < L; l += 1024) { sw[l] *= inv; wout[((size_t)b * P + j) * L + l] = sw[l]; }
    __syncthreads();
    // c[d] = sum_l w[l] vs[l][d]: thread (quarter, d) sums its quarter of the tokens, the quarters meet in a fixed order
    const int qt = threadIdx.x >> 8, d = threadIdx.x & 255;
    const int per = (L + 3) >> 2, l0 = qt * per, l1 = min(L, l0 + per);
    float a = 0.f;
    if (d < E)
        for (int l = l0; l < l1; ++l) a += sw[l] * vs[((size_t)b * L + l) * E + d];
    part[qt][d] = a;
    __syncthreads();
    if (qt == 0 && d < E) cout[((size_t)b * P + j) * E + d] = (part[0][d] + part[1][d]) + (part[2][d] + part[3][d]);
}

// backward: dvs[b,l,:] += w[l] dc[:]; dw[l] = dc . vs[l]; ds = w (dw - sum w dw); dk[b,:] += ds[l] qs[l]; dqs[b,l,:] += ds[l] k
__global__ __launch_bounds__(1024) void qenc_attn_bwd_kernel(const float* __restrict__ k, const float* __restrict__ qs,
                                                             const float* __restrict__ vs, const float* __restrict__ w,
                                                             const float* __restrict__ dc, float* __restrict__ dk,
                                                             float* __restrict__ dqs, float* __restrict__ dvs,
                                                             int P, int L, int E) {
    __shared__ float sds[128];
    __shared__ float red[16];
    __shared__ float gks[16][64];
    const int b = blockIdx.x / P, j = blockIdx.x % P;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* wr = w + ((size_t)b * P + j) * L;
    const float* dcr = dc + ((size_t)b * P + j) * E;
    for (int l = wave; l < L; l += 16) {
        float s = 0.f;
        for (int d = lane; d < E; d += 64) s += dcr[d] * vs[((size_t)b * L + l) * E + d];
        s = rt_wave_sum(s);
        if (lane == 0) sds[l] = s;      // dw[l]
    }
    __syncthreads();
    float acc = 0.f;
    for (int l = threadIdx.x; l < L; l += 1024) acc += wr[l] * sds[l];
    acc = rt_wave_sum(acc);
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) dot += red[i];
    __syncthreads();
    for (int l = threadIdx.x; l < L; l += 1024) sds[l] = wr[l] * (sds[l] - dot);
    __syncthreads();
    // Last (atomic-issue bound) loop: blockIdx.y owns 64 features, the sixteen waves split the token range; the partial dk
    // sums meet in LDS and are added in a fixed order, so dk gets ONE add per (phrase, feature) -- with one phrase per image
    // every output of this kernel is bit-reproducible (a four-way atomic sum here used to flip bf16 roundings downstream).
    const int d = (int)blockIdx.y * 64 + lane;
    const int per = (L + 15) >> 4;
    const int l0 = wave * per, l1 = min(L, l0 + per);
    float gk = 0.f;
    if (d < E) {
        const float kd = k[(size_t)b * E + d], dcd = dcr[d];
        for (int l = l0; l < l1; ++l) {
            const size_t o = ((size_t)b * L + l) * E + d;
            gk += sds[l] * qs[o];
            atomicAdd(dqs + o, sds[l] * kd);
            atomicAdd(dvs + o, wr[l] * dcd);
        }
    }
    gks[wave][lane] = gk;
    __syncthreads();
    if (wave == 0 && d < E) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += gks[i][lane];
        atomicAdd(dk + (size_t)b * E + d, t);
    }
}


// ---------------------------------------------------------------- tiny-N backward-data (the 4-wide box head)
// dx[m, k] = gate(sum_{n<N} dy[m, n] * w[n, k]),  N <= 8, fp32 dy / w, bf16 out
__global__ __launch_bounds__(256) void small_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                          const bf16_t* __restrict__ gate, bf16_t* __restrict__ dx,
                                                          int M, int N, int K) {
    const size_t total = (size_t)M * K;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int k = (int)(i % K);
        const int m = (int)(i / K);
        float a = 0.f;
        for (int n = 0; n < N; ++n) a += dy[(size_t)m * N + n] * w[(size_t)n * K + k];
        if (gate && !((float)gate[i] > 0.f)) a = 0.f;
        dx[i] = (bf16_t)a;
    }
}

// ---------------------------------------------------------------- gradients of the positional embeddings
// dpos fp32 [B*S, E] (gradient w.r.t. the `pos` sequence, models/reftr.py:51-120):
//   d lang_pos_embeddings[l] += sum_b dpos[b, l];  d token_type[0] += sum over language rows;
//   d level_embed[0], d token_type[1] += sum over image rows.
__global__ __launch_bounds__(256) void pos_grad_lang_kernel(const float* __restrict__ dpos, float* __restrict__ d_lang_pos,
                                                            float* __restrict__ d_type, int B, int S, int L, int E) {
    const int l = blockIdx.x;
    for (int c = threadIdx.x; c < E; c += 256) {
        float a = 0.f;
        for (int b = 0; b < B; ++b) a += dpos[((size_t)b * S + l) * E + c];
        d_lang_pos[(size_t)l * E + c] += a;
        atomicAdd(d_type + c, a);
    }
}
__global__ __launch_bounds__(256) void pos_grad_img_kernel(const float* __restrict__ dpos, float* __restrict__ d_type,
                                                           float* __restrict__ d_level, int B, int S, int L, int E) {
    const int b = blockIdx.y;
    const int r0 = L + blockIdx.x * 32, r1 = min(r0 + 32, S);
    for (int c = threadIdx.x; c < E; c += 256) {
        float a = 0.f;
        for (int r = r0; r < r1; ++r) a += dpos[((size_t)b * S + r) * E + c];
        atomicAdd(d_type + E + c, a);
        atomicAdd(d_level + c, a);
    }
}

}  // namespace

extern "C" int rt_colsum(const void* dy, int is_bf16, float* db, int M, int N, rt_stream_t stream) {
    if (!dy || !db || M <= 0 || N <= 0) return RT_ERR_BADARG;
    int ysplit = (M + 255) / 256; if (ysplit > 64) ysplit = 64;
    const int rpb = (M + ysplit - 1) / ysplit;
    const dim3 grid((N + 63) / 64, (M + rpb - 1) / rpb);
    if (is_bf16) hipLaunchKernelGGL(colsum_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, db, M, N, rpb);
    else hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)dy, db, M, N, rpb);
    RT_CHECK_LAUNCH();
    return RT_OK;
}

extern "C" int rt_rows_add(const rt_rows_add_desc* d, rt_stream_t stream) {
    if (!d || (!d->out_f32 && !d->out_bf16) || d->rows <= 0 || d->D <= 0) return RT_ERR_BADARG;
    const size_t total = (size_t)d->rows * d->D;
    int blocks = (int)((total + 255) / 256); if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(rows_add_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, *d);
    RT_CHECK_LAUNCH();
    return RT_OK;
}

extern "C" int rt_roberta_pos_ids(const int64_t* ids, int32_t* pos_ids, int B, int L, int pad_idx, rt_stream_t stream) {
    if (!ids || !pos_ids || B <= 0 || L <= 0) return RT_ERR_BADARG;
    hipLaunchKernelGGL(roberta_pos_ids_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, ids, pos_ids, B, L, pad_idx);
    RT_CHECK_LAUNCH();
    return RT_OK;
}

extern "C" int rt_bert_embed_fwd(const int64_t* ids, const float* word, const float* pos, const float* type0,
                                 float* out, int rows, int L, int D, const int32_t* pos_ids, rt_stream_t stream) {
    if (!ids || !word || !pos || !type0 || !out) return RT_ERR_BADARG;
    const size_t total = (size_t)rows * D;
    int blocks = (int)((total + 255) / 256); if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(bert_embed_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, ids, word, pos, type0, out, rows, L, D, pos_ids);
    RT_CHECK_LAUNCH();
    return RT_OK;
}

extern "C" int rt_bert_embed_bwd(const int64_t* ids, const float* de, float* dword, float* dpos, float* dtype0,
                                 int rows, int L, int D, const int32_t* pos_ids, rt_stream_t stream) {
    if (!ids || !de || !dword || !dpos || !dtype0) return RT_ERR_BADARG;
    const size_t total = (size_t)rows * D;
    int blocks = (int)((total + 255) / 256); if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(bert_embed_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, ids, de, dword, dpos, dtype0, rows, L, D, pos_ids);
    RT_CHECK_LAUNCH();
    return RT_OK;
}

extern "C" int rt_context_mask(const uint8_t* smask, const uint8_t* phrase_mask, const int64_t* pos_l, const int64_t* pos_r,
                               uint8_t* ctx, uint8_t* qmask, int B, int L, int P, int Lp, rt_stream_t stream) {
    if (!smask || !ctx || !qmask) return RT_ERR_BADARG;
    if (phrase_mask && (!pos_l || !pos_r || Lp < 3)) return RT_ERR_BADARG;
    hipLaunchKernelGGL(context_mask_kernel, dim3(B), dim3(128), 0, (hipStream_t)stream, smask, phrase_mask, pos_l, pos_r, ctx, qmask, B, L, P, Lp);
    RT_CHECK_LAUNCH();
    return RT_OK;
}

extern "C" int rt_qenc_attn_fwd(const float* k, const float* qs, const float* vs, const uint8_t* ctx, float* w, float* c,
                                int B, int P, int L, int E, rt_stream_t stream) {
    if (!k || !qs || !vs || !ctx || !w || !c) return RT_ERR_BADARG;
    if (L > 128 || L <= 0) return RT_ERR_UNSUPPORTED;
    if (E > 256) return RT_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(qenc_attn_fwd_kernel, dim3(B * P), dim3(1024), 0, (hipStream_t)stream, k, qs, vs, ctx, w, c, P, L, E);
    RT_CHECK_LAUNCH();
    return RT_OK;
}

extern "C" int rt_qenc_attn_bwd(const float* k, const float* qs, const float* vs, const float* w, const float* dc,
                                float* dk, float* dqs, float* dvs, int B, int P, int L, int E, rt_stream_t stream) {
    if (!k || !qs || !vs || !w || !dc || !dk || !dqs || !dvs) return RT_ERR_BADARG;
    if (L > 128 || L <= 0) return RT_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(qenc_attn_bwd_kernel, dim3(B * P, (E + 63) / 64), dim3(1024), 0, (hipStream_t)stream, k, qs, vs, w, dc, dk, dqs, dvs, P, L, E);
    RT_CHECK_LAUNCH();
    return RT_OK;
}

extern "C" int rt_small_dgrad(const float* dy, const float* w, const void* gate, void* dx, int M, int N, int K, rt_stream_t stream) {
    if (!dy || !w || !dx || N <= 0 || N > 8) return RT_ERR_BADARG;
    const size_t total = (size_t)M * K;
    int blocks = (int)((total + 255) / 256); if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(small_dgrad_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dy, w, (const bf16_t*)gate, (bf16_t*)dx, M, N, K);
    RT_CHECK_LAUNCH();
    return RT_OK;
}

extern "C" int rt_pos_grad(const float* dpos, float* d_lang_pos, float* d_type, float* d_level, int B, int S, int L, int E,
                           rt_stream_t stream) {
    if (!dpos || !d_lang_pos || !d_type || !d_level) return RT_ERR_BADARG;
    if (L > 0) hipLaunchKernelGGL(pos_grad_lang_kernel, dim3(L), dim3(256), 0, (hipStream_t)stream, dpos, d_lang_pos, d_type, B, S, L, E);
    if (S > L) hipLaunchKernelGGL(pos_grad_img_kernel, dim3((S - L + 31) / 32, B), dim3(256), 0, (hipStream_t)stream, dpos, d_type, d_level, B, S, L, E);
    RT_CHECK_LAUNCH();
    return RT_OK;
}


// rt_zero_chunks: see include/reftr_hip.h.  One workgroup per chunk (<= 16384 floats); 16-B stores where the chunk is aligned.
namespace {
__global__ __launch_bounds__(256) void zero_chunks_kernel(float* __restrict__ base, const int64_t* __restrict__ table) {
    const int64_t off = table[2 * blockIdx.x], cnt = table[2 * blockIdx.x + 1];
    float* p = base + off;
    int64_t i = threadIdx.x;
    const int64_t head = (4 - (off & 3)) & 3;                    // elements in front of the first 16-B boundary
    if (i < head && i < cnt) p[i] = 0.f;
    const int64_t n4 = cnt > head ? (cnt - head) >> 2 : 0;
    f32x4* p4 = reinterpret_cast<f32x4*>(p + head);
    for (int64_t j = threadIdx.x; j < n4; j += 256) p4[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int64_t tail0 = head + (n4 << 2);
    if (tail0 + i < cnt) p[tail0 + i] = 0.f;
}
}  // namespace

extern "C" int rt_zero_chunks(float* base, const int64_t* table, int n, rt_stream_t stream) {
    if (!base || !table || n <= 0) return RT_ERR_BADARG;
    hipLaunchKernelGGL(zero_chunks_kernel, dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream, base, table);
    RT_CHECK_LAUNCH();
    return RT_OK;
}
