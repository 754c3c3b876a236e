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
    const bf16_t* dy; const bf16_t* x; float* dw; const float* scale;
    int B, SH, SW, SC, DH, DW, N, KH, KW, stride, pad;
    int M, chunks_per_block, c_tiles;
};

__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* base, int off0, int off1) {
    typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_ptr;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(base + off0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(base + off1));
    union { struct { s16x4 a, b; } s; bf16x8 v; } u;
    u.s.a = lo; u.s.b = hi;
    return u.v;
}

template <int BN, int BC>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradArgs p) {
    constexpr int TN = BN / 32, TC = BC / 32;       // MFMA tiles per wave (waves 2(n) x 2(c))
    constexpr int SA = BN * 2 + 32, SB = BC * 2 + 32;   // LDS row strides (bytes)
    constexpr int A_BYTES = 32 * SA, B_BYTES = 32 * SB, BUF_BYTES = A_BYTES + B_BYTES;
    constexpr int ACH = BN / 8, BCH = BC / 8;       // 16-B chunks per row
    constexpr int AJ = (32 * ACH) / 256, BJ = (32 * BCH) / 256;   // chunks per thread (>=1)
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

    const int a_chunk = t % ACH, a_row = t / ACH;          // rows a_row + (256/ACH)*j
    const int b_chunk = t % BCH, b_row = t / BCH;
    constexpr int A_RSTEP = 256 / ACH, B_RSTEP = 256 / BCH;
    const bool simple = (p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad == 0);
    const bool n_vec = (p.N & 7) == 0;

    f32x4 acc[TN][TC];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TC; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    uint4 ra[AJ], rb[BJ];
    const uint4 zero4 = make_uint4(0, 0, 0, 0);

#define RT_WG_LOAD(ch_)                                                                             \
    {                                                                                               \
        const int mbase = (ch_) << 5;                                                               \
        _Pragma("unroll") for (int j = 0; j < AJ; ++j) {                                            \
            const int m = mbase + a_row + A_RSTEP * j;                                              \
            const int n = n0 + a_chunk * 8;                                                         \
            if (n_vec) {                                                                            \
                ra[j] = (m < p.M && n < p.N) ? *reinterpret_cast<const uint4*>(p.dy + (size_t)m * p.N + n) : zero4; \
            } else { /* ragged N (e.g. the 4-wide box head): element loads, zero fill */             \
                union { uint4 q; bf16_t e[8]; } u; u.q = zero4;                                     \
                if (m < p.M) { _Pragma("unroll") for (int e = 0; e < 8; ++e)                         \
                    if (n + e < p.N) u.e[e] = p.dy[(size_t)m * p.N + n + e]; }                       \
                ra[j] = u.q;                                                                        \
            }                                                                                       \
        }                                                                                         \
        _Pragma("unroll") for (int j = 0; j < BJ; ++j) {                                            \
            const int m = mbase + b_row + B_RSTEP * j;                                              \
            const int c = c0 + b_chunk * 8;                                                         \
            bool ok = (m < p.M) && (c < p.SC);                                                      \
            size_t pix;                                                                             \
            if (simple) { pix = (size_t)m; }                                                        \
            else {                                                                                  \
                const int mm = ok ? m : 0;                                                          \
                const int dx = mm % p.DW; const int tmp = mm / p.DW;                                \
                const int dy_ = tmp % p.DH; const int bb = tmp / p.DH;                              \
                const int sy = dy_ * p.stride - p.pad + kh, sx = dx * p.stride - p.pad + kw;        \
                ok = ok && (unsigned)sy < (unsigned)p.SH && (unsigned)sx < (unsigned)p.SW;          \
                pix = (size_t)((bb * p.SH + sy) * p.SW + sx);                                       \
            }                                                                                       \
            rb[j] = ok ? *reinterpret_cast<const uint4*>(p.x + pix * p.SC + c) : zero4;             \
        }                                                                                           \
    }

#define RT_WG_STORE(buf_)                                                                           \
    {                                                                                               \
        unsigned char* bA = smem + (buf_) * BUF_BYTES;                                              \
        unsigned char* bB = bA + A_BYTES;                                                           \
        _Pragma("unroll") for (int j = 0; j < AJ; ++j)                                              \
            *reinterpret_cast<uint4*>(bA + (a_row + A_RSTEP * j) * SA + a_chunk * 16) = ra[j];      \
        _Pragma("unroll") for (int j = 0; j < BJ; ++j)                                              \
            *reinterpret_cast<uint4*>(bB + (b_row + B_RSTEP * j) * SB + b_chunk * 16) = rb[j];      \
    }

    RT_WG_LOAD(chunk_begin);
    RT_WG_STORE(0);
    __syncthreads();

    // per-lane byte offsets of the two transpose reads inside a 16-channel column block
    const int tr_row0 = 4 * lg + (li >> 2);
    const int tr_col = (li & 3) * 8;

    int cur = 0;
    for (int ch = chunk_begin; ch < chunk_end; ++ch) {
        const bool has_next = (ch + 1) < chunk_end;
        if (has_next) RT_WG_LOAD(ch + 1);
        const unsigned char* bA = smem + cur * BUF_BYTES;
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
        if (has_next) RT_WG_STORE(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
#undef RT_WG_LOAD
#undef RT_WG_STORE

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

template <int BN, int BC>
int launch_wgrad(WgradArgs a, int msplit, hipStream_t s) {
    const int nt = (a.N + BN - 1) / BN;
    a.c_tiles = (a.SC + BC - 1) / BC;
    const int taps = a.KH * a.KW;
    const int total_chunks = (a.M + 31) / 32;
    const long long base_blocks = (long long)nt * a.c_tiles * taps;
    if (msplit <= 0) {
        long long want = (1024 + base_blocks - 1) / base_blocks;   // aim for ~1024 workgroups
        long long maxs = (total_chunks + 3) / 4;                   // at least 4 chunks per block
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
    hipLaunchKernelGGL((conv_wgrad_kernel<BN, BC>), dim3((unsigned)base_blocks, (unsigned)gy), dim3(256), smem, s, a);
    RT_CHECK_LAUNCH();
    return RT_OK;
}

}  // namespace

extern "C" int rt_conv_wgrad(const rt_conv_wgrad_desc* d, rt_stream_t stream) {
    if (!d || !d->dy || !d->x || !d->dw) return RT_ERR_BADARG;
    if (d->SC <= 0 || (d->SC & 15) || (d->N & 3) || d->N <= 0) return RT_ERR_UNSUPPORTED;
    if (d->KH <= 0 || d->KW <= 0 || d->B <= 0 || d->DH <= 0 || d->DW <= 0 || d->stride <= 0) return RT_ERR_BADARG;
    WgradArgs a;
    a.dy = (const bf16_t*)d->dy; a.x = (const bf16_t*)d->x; a.dw = d->dw; a.scale = d->scale;
    a.B = d->B; a.SH = d->SH; a.SW = d->SW; a.SC = d->SC; a.DH = d->DH; a.DW = d->DW; a.N = d->N;
    a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad = d->pad;
    const long long M = (long long)d->B * d->DH * d->DW;
    if (M > 0x7fffffffLL / 4) return RT_ERR_UNSUPPORTED;
    a.M = (int)M; a.chunks_per_block = 0; a.c_tiles = 0;
    hipStream_t s = (hipStream_t)stream;
    if (a.N >= 128 && a.SC >= 128) return launch_wgrad<128, 128>(a, d->msplit, s);
    if (a.N >= 128) return launch_wgrad<128, 64>(a, d->msplit, s);
    if (a.SC >= 128) return launch_wgrad<64, 128>(a, d->msplit, s);
    return launch_wgrad<64, 64>(a, d->msplit, s);
}
