// rt_conv_wgrad: weight-gradient GEMM, dw[n][tap][c] += scale[n] * sum_m dy[m,n] * xg[m,(tap,c)].
//
// The reduction axis (pixels / tokens, m) is the SLOW axis of both operands, so the natural 16-B global
// loads give LDS tiles [32 m-rows][BN or BC channels] with channels contiguous.  The MFMA wants 8
// consecutive reduction elements per lane for one fixed channel; that transpose is done by the gfx950 LDS
// transpose read: each 16-lane group hands ds_read_b64_tr_b16 a 4(m) x 16(channel) block and lane i gets
// channel i's 4 m-values.  Two such reads (m-rows 4g..4g+3 and 16+4g..16+4g+3) make one bf16x8 fragment;
// both operands use the same m -> k-slot assignment so the contraction is consistent.
// Row stride of the LDS tiles is padded by 32 B: the 8 rows a 32-lane half touches land on 8 distinct
// 32-B bank segments (conflict-free for the 2x32-lane servicing of the transpose read).
//
// The m axis is split across blockIdx.y; partial tiles are accumulated with fp32 global atomics into the
// (pre-zeroed) flat gradient buffer.
#include "rt_common.h"

namespace {

struct WgradArgs {
    const bf16_t* dy; const bf16_t* x; float* dw; const float* scale; float* dbias;
    int B, SH, SW, SC, DH, DW, N, KH, KW, stride, pad;
    int M, chunks_per_block, c_tiles;
    unsigned dy_bytes, x_bytes;
};

__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* base, int off0, int off1) {
    typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_ptr;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(base + off0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(base + off1));
    union { struct { s16x4 a, b; } s; bf16x8 v; } u;
    u.s.a = lo; u.s.b = hi;
    return u.v;
}

// SIMPLE: 1x1 / stride 1 / pad 0 (x row = m).  NVEC: N % 8 == 0 (16-B dy loads).
template <int BN, int BC, bool SIMPLE, bool NVEC>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(const bf16_t* __restrict__ dyp,
                                                         const bf16_t* __restrict__ xp, const WgradArgs p) {
    constexpr int TN = BN / 32, TC = BC / 32;       // MFMA tiles per wave (waves 2(n) x 2(c))
    constexpr int SA = BN * 2 + 32, SB = BC * 2 + 32;   // LDS row strides (bytes)
    constexpr int A_BYTES = 32 * SA, B_BYTES = 32 * SB, BUF_BYTES = A_BYTES + B_BYTES;
    constexpr int ACH = BN / 8, BCH = BC / 8;       // 16-B chunks per row
    constexpr int AJ = (32 * ACH) / 256, BJ = (32 * BCH) / 256;   // chunks per thread (>=1)
    constexpr int A_RSTEP = 256 / ACH, B_RSTEP = 256 / BCH;
    static_assert(AJ >= 1 && BJ >= 1, "tile too small");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wn = wave & 1, wc = wave >> 1;
    const int li = lane & 15, lg = lane >> 4;

    const int n_tiles = (p.N + BN - 1) / BN;
    const int tile_n = blockIdx.x % n_tiles;
    const int rest = blockIdx.x / n_tiles;
    const int tile_c = rest % p.c_tiles;
    const int tap = rest / p.c_tiles;
    const int kh = tap / p.KW, kw = tap - kh * p.KW;
    const int n0 = tile_n * BN, c0 = tile_c * BC;

    const int chunk_begin = blockIdx.y * p.chunks_per_block;
    const int total_chunks = (p.M + 31) >> 5;
    int chunk_end = chunk_begin + p.chunks_per_block;
    if (chunk_end > total_chunks) chunk_end = total_chunks;
    if (chunk_begin >= chunk_end) return;

    const int a_chunk = t % ACH, a_row = t / ACH;          // rows a_row + A_RSTEP*j
    const int b_chunk = t % BCH, b_row = t / BCH;
    const int a_n = n0 + a_chunk * 8;
    const bool a_nok = a_n < p.N;
    const int b_c = c0 + b_chunk * 8;
    const bool b_cok = b_c < p.SC;

    // running (batch, y, x) of each staged x-row for the gather modes (advanced by 32 rows per chunk)
    int gb[BJ], gy[BJ], gx[BJ];
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
        const int m = (chunk_begin << 5) + b_row + B_RSTEP * j;
        gx[j] = m % p.DW; const int tmp = m / p.DW; gy[j] = tmp % p.DH; gb[j] = tmp / p.DH;
    }

    f32x4 acc[TN][TC];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TC; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    const bool do_bias = p.dbias && tile_c == 0 && tap == 0;
    float bsum = 0.f;

    // Two register stages + two LDS buffers, operands fetched with bounds-checked buffer loads (out-of-range offset
    // -> zeros): chunk ch+2 is requested while chunk ch is multiplied and chunk ch+1 is still in flight.
    uint4 ra0[AJ], rb0[BJ], ra1[AJ], rb1[BJ];
    const __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(dyp), 0, p.dy_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(xp), 0, p.x_bytes, 0x00020000);
    constexpr int OOB = 0x7fffffff;
    int lc = chunk_begin;             // next chunk to load (the gather state gb/gy/gx belongs to it)

    auto load_tiles = [&](uint4 (&ra)[AJ], uint4 (&rb)[BJ]) __attribute__((always_inline)) {
        const int mbase = lc << 5;
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const int m = mbase + a_row + A_RSTEP * j;
            const bool ok = a_nok && m < p.M;
            if (NVEC) {
                const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs_dy, ok ? (m * p.N + a_n) * 2 : OOB, 0, 0);
                ra[j] = make_uint4(v[0], v[1], v[2], v[3]);
            } else {   // ragged N (e.g. the 4-wide box head): element loads, zero fill
                union { uint4 q; unsigned short e[8]; } u; u.q = make_uint4(0, 0, 0, 0);
                const unsigned short* d16 = reinterpret_cast<const unsigned short*>(dyp);
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (ok && a_n + e < p.N) u.e[e] = d16[m * p.N + a_n + e];
                ra[j] = u.q;
            }
        }
        const bool last = (lc + 1 >= chunk_end);
#pragma unroll
        for (int j = 0; j < BJ; ++j) {
            const int m = mbase + b_row + B_RSTEP * j;
            bool ok = b_cok && m < p.M;
            int pix;
            if (SIMPLE) pix = m;
            else {
                const int sy = gy[j] * p.stride - p.pad + kh, sx = gx[j] * p.stride - p.pad + kw;
                ok = ok && (unsigned)sy < (unsigned)p.SH && (unsigned)sx < (unsigned)p.SW;
                pix = (gb[j] * p.SH + sy) * p.SW + sx;
                if (!last) {
                    gx[j] += 32;
                    while (gx[j] >= p.DW) { gx[j] -= p.DW; if (++gy[j] >= p.DH) { gy[j] = 0; ++gb[j]; } }
                }
            }
            const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs_x, ok ? (pix * p.SC + b_c) * 2 : OOB, 0, 0);
            rb[j] = make_uint4(v[0], v[1], v[2], v[3]);
        }
        if (!last) ++lc;              // past the end the last chunk is re-loaded (never consumed)
    };
    auto store_tiles = [&](int buf, const uint4 (&ra)[AJ], const uint4 (&rb)[BJ]) __attribute__((always_inline)) {
        unsigned char* bA = smem + buf * BUF_BYTES;
        unsigned char* bB = bA + A_BYTES;
#pragma unroll
        for (int j = 0; j < AJ; ++j)
            *reinterpret_cast<uint4*>(bA + (a_row + A_RSTEP * j) * SA + a_chunk * 16) = ra[j];
#pragma unroll
        for (int j = 0; j < BJ; ++j)
            *reinterpret_cast<uint4*>(bB + (b_row + B_RSTEP * j) * SB + b_chunk * 16) = rb[j];
    };
    // per-lane byte offsets of the two transpose reads inside a 16-channel column block
    const int tr_row0 = 4 * lg + (li >> 2);
    const int tr_col = (li & 3) * 8;
    auto compute = [&](int buf) __attribute__((always_inline)) {
        const unsigned char* bA = smem + buf * BUF_BYTES;
        const unsigned char* bB = bA + A_BYTES;
        bf16x8 af[TN], bfr[TC];
#pragma unroll
        for (int a = 0; a < TN; ++a) {
            const int colb = (wn * (BN / 2) + a * 16) * 2 + tr_col;
            af[a] = tr_frag(bA, tr_row0 * SA + colb, (tr_row0 + 16) * SA + colb);
        }
#pragma unroll
        for (int b = 0; b < TC; ++b) {
            const int colb = (wc * (BC / 2) + b * 16) * 2 + tr_col;
            bfr[b] = tr_frag(bB, tr_row0 * SB + colb, (tr_row0 + 16) * SB + colb);
        }
#pragma unroll
        for (int a = 0; a < TN; ++a)
#pragma unroll
            for (int b = 0; b < TC; ++b)
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a], bfr[b], acc[a][b], 0, 0, 0);
        if (do_bias) {      // fused bias gradient: column sums of the dy tile that is already in LDS
            constexpr int TPC = 256 / BN;            // threads per column
            constexpr int RPT = 32 / TPC;            // rows per thread
            const int col = t % BN, r0 = (t / BN) * RPT;
#pragma unroll
            for (int r = 0; r < RPT; ++r)
                bsum += (float)*reinterpret_cast<const bf16_t*>(bA + (r0 + r) * SA + col * 2);
        }
    };

    const int nch = chunk_end - chunk_begin;
    load_tiles(ra0, rb0);
    load_tiles(ra1, rb1);
    store_tiles(0, ra0, rb0);
    __syncthreads();
    for (int c = 0; c < nch; c += 2) {
        load_tiles(ra0, rb0);             // chunk c+2
        compute(0);
        store_tiles(1, ra1, rb1);
        __syncthreads();
        if (c + 1 >= nch) break;
        load_tiles(ra1, rb1);             // chunk c+3
        compute(1);
        store_tiles(0, ra0, rb0);
        __syncthreads();
    }

    if (do_bias) {
        const int n = n0 + t % BN;
        if (n < p.N) atomicAdd(p.dbias + n, bsum);
    }
    const int taps = p.KH * p.KW;
#pragma unroll
    for (int a = 0; a < TN; ++a) {
        const int nb = n0 + wn * (BN / 2) + a * 16 + lg * 4;
#pragma unroll
        for (int b = 0; b < TC; ++b) {
            const int c = c0 + wc * (BC / 2) + b * 16 + li;
            if (c >= p.SC) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = nb + r;
                if (n >= p.N) continue;
                float v = acc[a][b][r];
                if (p.scale) v *= p.scale[n];
                atomicAdd(p.dw + ((size_t)n * taps + tap) * p.SC + c, v);
            }
        }
    }
}

// M <= 16 rows (decoder-side Linears): plain outer-product accumulation, one thread per (n, 4 k's)
__global__ __launch_bounds__(256) void small_m_wgrad_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                            float* __restrict__ dw, const float* __restrict__ scale,
                                                            float* __restrict__ dbias, int M, int N, int K) {
    const int k4 = K >> 2;
    const size_t total = (size_t)N * k4;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int kk = (int)(i % k4) * 4;
        const int n = (int)(i / k4);
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        float gs = 0.f;
        for (int m = 0; m < M; ++m) {
            const float g = (float)dy[(size_t)m * N + n];
            gs += g;
            const bf16x4 xv = *reinterpret_cast<const bf16x4*>(x + (size_t)m * K + kk);
#pragma unroll
            for (int r = 0; r < 4; ++r) a[r] += g * (float)xv[r];
        }
        if (scale) a *= scale[n];
        f32x4* o = reinterpret_cast<f32x4*>(dw + (size_t)n * K + kk);
        *o = *o + a;
        if (dbias && kk == 0) dbias[n] += gs;
    }
}

template <int BN, int BC>
int launch_wgrad(WgradArgs a, int msplit, hipStream_t s) {
    const int nt = (a.N + BN - 1) / BN;
    a.c_tiles = (a.SC + BC - 1) / BC;
    const int taps = a.KH * a.KW;
    const int total_chunks = (a.M + 31) / 32;
    const long long base_blocks = (long long)nt * a.c_tiles * taps;
    if (msplit <= 0) {
        long long want = (512 + base_blocks - 1) / base_blocks;    // aim for ~512 workgroups ...
        long long maxs = total_chunks / 16;                        // ... of >= 16 chunks (512 rows): each split costs a
                                                                   // full tile of fp32 atomics
        if (maxs < 1) maxs = 1;
        if (want > maxs) want = maxs;
        if (want < 1) want = 1;
        msplit = (int)want;
    }
    if (msplit > total_chunks) msplit = total_chunks;
    if (msplit < 1) msplit = 1;
    a.chunks_per_block = (total_chunks + msplit - 1) / msplit;
    const int gy = (total_chunks + a.chunks_per_block - 1) / a.chunks_per_block;
    constexpr size_t smem = 2 * (size_t)(32 * (BN * 2 + 32) + 32 * (BC * 2 + 32));
    const dim3 grid((unsigned)base_blocks, (unsigned)gy), block(256);
    const bool simple = (a.KH == 1 && a.KW == 1 && a.stride == 1 && a.pad == 0);
    const bool nvec = (a.N & 7) == 0;
    if (simple && nvec)       hipLaunchKernelGGL((conv_wgrad_kernel<BN, BC, true, true>), grid, block, smem, s, a.dy, a.x, a);
    else if (simple)          hipLaunchKernelGGL((conv_wgrad_kernel<BN, BC, true, false>), grid, block, smem, s, a.dy, a.x, a);
    else if (nvec)            hipLaunchKernelGGL((conv_wgrad_kernel<BN, BC, false, true>), grid, block, smem, s, a.dy, a.x, a);
    else                      hipLaunchKernelGGL((conv_wgrad_kernel<BN, BC, false, false>), grid, block, smem, s, a.dy, a.x, a);
    RT_CHECK_LAUNCH();
    return RT_OK;
}

}  // namespace

extern "C" int rt_conv_wgrad(const rt_conv_wgrad_desc* d, rt_stream_t stream) {
    if (!d || !d->dy || !d->x || !d->dw) return RT_ERR_BADARG;
    if (d->SC <= 0 || (d->SC & 15) || (d->N & 3) || d->N <= 0) return RT_ERR_UNSUPPORTED;
    if (d->KH <= 0 || d->KW <= 0 || d->B <= 0 || d->DH <= 0 || d->DW <= 0 || d->stride <= 0) return RT_ERR_BADARG;
    WgradArgs a;
    a.dy = (const bf16_t*)d->dy; a.x = (const bf16_t*)d->x; a.dw = d->dw; a.scale = d->scale; a.dbias = d->dbias;
    a.B = d->B; a.SH = d->SH; a.SW = d->SW; a.SC = d->SC; a.DH = d->DH; a.DW = d->DW; a.N = d->N;
    a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad = d->pad;
    const long long M = (long long)d->B * d->DH * d->DW;
    if (M > 0x7fffffffLL / 4) return RT_ERR_UNSUPPORTED;
    if (M * d->N >= 0x3fffffffLL || (long long)d->B * d->SH * d->SW * d->SC >= 0x3fffffffLL) return RT_ERR_UNSUPPORTED;
    a.M = (int)M; a.chunks_per_block = 0; a.c_tiles = 0;
    a.dy_bytes = (unsigned)(M * d->N * 2);
    a.x_bytes = (unsigned)((long long)d->B * d->SH * d->SW * d->SC * 2);
    hipStream_t s = (hipStream_t)stream;
    if (a.M <= 16 && a.KH == 1 && a.KW == 1 && a.stride == 1 && a.pad == 0 && (a.SC & 3) == 0) {
        const size_t total = (size_t)a.N * (a.SC >> 2);
        int blocks = (int)((total + 255) / 256); if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(small_m_wgrad_kernel, dim3(blocks), dim3(256), 0, s, a.dy, a.x, a.dw, a.scale, a.dbias, a.M, a.N, a.SC);
        RT_CHECK_LAUNCH();
        return RT_OK;
    }
    if (a.N >= 128 && a.SC >= 128) return launch_wgrad<128, 128>(a, d->msplit, s);
    if (a.N >= 128) return launch_wgrad<128, 64>(a, d->msplit, s);
    if (a.SC >= 128) return launch_wgrad<64, 128>(a, d->msplit, s);
    return launch_wgrad<64, 64>(a, d->msplit, s);
}
