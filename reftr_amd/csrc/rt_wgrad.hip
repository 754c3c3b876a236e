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
#include <stdlib.h>

namespace {

struct WgradArgs {
    const bf16_t* dy; const bf16_t* x; float* dw; const float* scale; float* dbias;
    int B, SH, SW, SC, DH, DW, N, KH, KW, stride, pad, dil;
    int M, chunks_per_block, c_tiles;
    float* part; int out_elems; int xcd, early, epi_lds;      // split partials workspace ([split][N*taps*SC]) or NULL -> atomics
    int overwrite;            // single-writer launches (one split, no workspace) store instead of accumulating
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
                const int sy = gy[j] * p.stride - p.pad + kh * p.dil, sx = gx[j] * p.stride - p.pad + kw * p.dil;
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

// ------------------------------------------------------------------------------------------------
// LDS-DMA variant (N % 8 == 0): both operand tiles go global -> LDS with buffer_load_dwordx4 ... lds, CR reduction
// rows per stage, NS stages.  The DMA lands lane l of a wave instruction at LDS base + 16*l, so rows are unpadded
// (row bytes = 2*BN / 2*BC) and the bank spreading the transpose read needs comes from a source-side XOR of the
// 16-B slot index with a function of the row: 8 consecutive rows put their 32-B fragments on 8 distinct bank groups.
// SIMPLE operands are addressed through a buffer descriptor that is re-based per chunk with scalar arithmetic (base +=
// chunk bytes, num_records -= chunk bytes): the per-lane offsets are loop-invariant and the ragged tail rows fall off
// the end of the descriptor, i.e. read as zeros.
template <int N> __device__ __forceinline__ void wg_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
typedef __attribute__((ext_vector_type(4))) int wg_i32x4;
__device__ __forceinline__ wg_i32x4 wg_make_rsrc(const void* ptr, unsigned bytes) {
    const uint64_t a = (uint64_t)ptr;
    return wg_i32x4{(int)(uint32_t)a, (int)(uint32_t)(a >> 32), (int)bytes, 0x00020000};
}
__device__ __forceinline__ void wg_dma16(const wg_i32x4 rsrc, unsigned lds_base, int voff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                 ::"s"(lds_base), "v"(voff), "s"(rsrc)
                 : "memory", "m0");
}
template <int RB> __device__ __forceinline__ int wg_swz(int row) {      // XOR applied to the 16-B slot index
    return RB >= 256 ? ((row & 7) << 1) : (((row >> 1) & 3) << 1);
}

// `lin_raw` = linear workgroup id inside this problem's (gx x gy) grid: the kernel is launched either alone
// (conv_wgrad_dma_kernel) or as one of up to 24 problems of a grouped launch (conv_wgrad_dma_grouped_kernel).
template <int BN, int BC, bool SIMPLE, int CR, int NS>
__device__ __forceinline__ void wgrad_dma_body(const bf16_t* __restrict__ dyp, const bf16_t* __restrict__ xp, const WgradArgs& p,
                                               const int lin_raw, const int grid_x, const int grid_y) {
    constexpr int TN = BN / 32, TC = BC / 32;
    constexpr int RBA = BN * 2, RBB = BC * 2;            // LDS row bytes
    constexpr int SPRA = BN / 8, SPRB = BC / 8;          // 16-B slots per row
    constexpr int A_BYTES = CR * RBA, B_BYTES = CR * RBB, BUF_BYTES = A_BYTES + B_BYTES;
    constexpr int AJ = CR * SPRA / 256, BJ = CR * SPRB / 256;
    constexpr int RPIA = 64 / SPRA, RPIB = 64 / SPRB;    // rows covered by one wave instruction
    constexpr int KS = CR / 32;
    constexpr int LPT = AJ + BJ;
    constexpr int OOB = 0x7fffffff;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef __attribute__((address_space(3))) void* lds_ptr;

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wn = wave & 1, wc = wave >> 1;
    const int li = lane & 15, lg = lane >> 4;

    const int n_tiles = (p.N + BN - 1) / BN;
    // linear workgroup id -> (tile, split) with every split's tiles on one XCD (they read the same dy / x rows)
    const int lin = rt_xcd_remap(lin_raw, grid_x * grid_y, p.xcd);
    const int by = __builtin_amdgcn_readfirstlane(lin / grid_x);      // (integer division goes through VALU)
    const int bx = __builtin_amdgcn_readfirstlane(lin - by * grid_x);
    const int tile_n = bx % n_tiles;
    const int rest = bx / n_tiles;
    const int tile_c = rest % p.c_tiles;
    const int tap = rest / p.c_tiles;
    const int kh = tap / p.KW, kw = tap - kh * p.KW;
    const int n0 = tile_n * BN, c0 = tile_c * BC;

    const int total_chunks = (p.M + CR - 1) / CR;
    const int chunk_begin = by * p.chunks_per_block;
    int chunk_end = chunk_begin + p.chunks_per_block;
    if (chunk_end > total_chunks) chunk_end = total_chunks;
    if (chunk_begin >= chunk_end) return;

    // loop-invariant per-lane source offsets (relative to the chunk's first row)
    int voff_a[AJ];
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
        const int r = (j * 4 + wave) * RPIA + lane / SPRA;
        const int sl = (lane % SPRA) ^ wg_swz<RBA>(r);
        const int n = n0 + sl * 8;
        voff_a[j] = n < p.N ? (r * p.N + n) * 2 : OOB;
    }
    int voff_b[BJ], b_r[BJ], gb[BJ], gy[BJ], gx[BJ];
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
        const int r = (j * 4 + wave) * RPIB + lane / SPRB;
        const int sl = (lane % SPRB) ^ wg_swz<RBB>(r);
        const int c = c0 + sl * 8;
        b_r[j] = r;
        if (SIMPLE) { voff_b[j] = c < p.SC ? (r * p.SC + c) * 2 : OOB; gb[j] = gy[j] = gx[j] = 0; }
        else {
            voff_b[j] = c < p.SC ? c * 2 : OOB;
            const int m = chunk_begin * CR + r;
            gx[j] = m % p.DW; const int tmp = m / p.DW; gy[j] = tmp % p.DH; gb[j] = tmp / p.DH;
        }
    }

    f32x4 acc[TN][TC];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TC; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    const bool do_bias = p.dbias && tile_c == 0 && tap == 0;
    float bsum = 0.f;

    const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr)smem + (unsigned)__builtin_amdgcn_readfirstlane(wave) * 1024u;
    const wg_i32x4 rs_x_abs = wg_make_rsrc(xp, p.x_bytes);
    int lc = chunk_begin;

    auto issue = [&](int stage) __attribute__((always_inline)) {
        const unsigned bA = lds0 + stage * BUF_BYTES, bB = bA + A_BYTES;
        const unsigned aoff = (unsigned)lc * (unsigned)(CR * 2) * (unsigned)p.N;
        const wg_i32x4 rs_dy = wg_make_rsrc(reinterpret_cast<const unsigned char*>(dyp) + aoff, p.dy_bytes - aoff);
#pragma unroll
        for (int j = 0; j < AJ; ++j) wg_dma16(rs_dy, bA + j * 4096, voff_a[j]);
        const bool last = (lc + 1 >= chunk_end);
        if (SIMPLE) {
            const unsigned xoff = (unsigned)lc * (unsigned)(CR * 2) * (unsigned)p.SC;
            const wg_i32x4 rs_x = wg_make_rsrc(reinterpret_cast<const unsigned char*>(xp) + xoff, p.x_bytes - xoff);
#pragma unroll
            for (int j = 0; j < BJ; ++j) wg_dma16(rs_x, bB + j * 4096, voff_b[j]);
        } else {
#pragma unroll
            for (int j = 0; j < BJ; ++j) {
                const int m = lc * CR + b_r[j];
                const int sy = gy[j] * p.stride - p.pad + kh * p.dil, sx = gx[j] * p.stride - p.pad + kw * p.dil;
                const bool ok = m < p.M && (unsigned)sy < (unsigned)p.SH && (unsigned)sx < (unsigned)p.SW && voff_b[j] != OOB;
                const int pix = (gb[j] * p.SH + sy) * p.SW + sx;
                wg_dma16(rs_x_abs, bB + j * 4096, ok ? pix * p.SC * 2 + voff_b[j] : OOB);
                if (!last) {
                    gx[j] += CR;
                    while (gx[j] >= p.DW) { gx[j] -= p.DW; if (++gy[j] >= p.DH) { gy[j] = 0; ++gb[j]; } }
                }
            }
        }
        if (!last) ++lc;
    };

    // transpose-read addresses: lane reads row tr_row0 (+16) of a 16-channel column block, 8 B at (li&3)*8
    const int tr_row0 = 4 * lg + (li >> 2);
    const int tr_col = (li & 3) * 8;
    int addr_a[TN], addr_b[TC];
#pragma unroll
    for (int a = 0; a < TN; ++a) {
        const int colb = (wn * (BN / 2) + a * 16) * 2 + tr_col;
        addr_a[a] = tr_row0 * RBA + ((((colb >> 4) ^ wg_swz<RBA>(tr_row0)) << 4) | (colb & 15));
    }
#pragma unroll
    for (int b = 0; b < TC; ++b) {
        const int colb = (wc * (BC / 2) + b * 16) * 2 + tr_col;
        addr_b[b] = tr_row0 * RBB + ((((colb >> 4) ^ wg_swz<RBB>(tr_row0)) << 4) | (colb & 15));
    }
    auto compute = [&](int stage) __attribute__((always_inline)) {
        const unsigned char* bA = smem + stage * BUF_BYTES;
        const unsigned char* bB = bA + A_BYTES;
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            bf16x8 af[TN], bfr[TC];
#pragma unroll
            for (int a = 0; a < TN; ++a) af[a] = tr_frag(bA, addr_a[a] + kk * 32 * RBA, addr_a[a] + (kk * 32 + 16) * RBA);
#pragma unroll
            for (int b = 0; b < TC; ++b) bfr[b] = tr_frag(bB, addr_b[b] + kk * 32 * RBB, addr_b[b] + (kk * 32 + 16) * RBB);
#pragma unroll
            for (int a = 0; a < TN; ++a)
#pragma unroll
                for (int b = 0; b < TC; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a], bfr[b], acc[a][b], 0, 0, 0);
        }
        if (do_bias) {      // fused bias gradient: column sums of the dy tile that is already in LDS
            constexpr int TPC = 256 / BN, RPT = CR / TPC;
            const int col = t % BN, r0 = (t / BN) * RPT;
#pragma unroll
            for (int r = 0; r < RPT; ++r) {
                const int row = r0 + r;
                bsum += (float)*reinterpret_cast<const bf16_t*>(bA + row * RBA + ((((col >> 3) ^ wg_swz<RBA>(row)) << 4) | ((col & 7) * 2)));
            }
        }
    };

    const int nch = chunk_end - chunk_begin;
#pragma unroll
    for (int s0 = 0; s0 < NS - 1; ++s0) issue(s0);
    int cbuf = 0, lbuf = NS - 1;
    if (p.early) {          // issue-before-wait (see rt_gemm.hip): NS-1 chunks in flight while parked, two barriers per chunk
        for (int c = 0; c < nch; ++c) {
            issue(lbuf);
            wg_wait_vmcnt<(NS - 1) * LPT>();
            __syncthreads();
            compute(cbuf);
            __syncthreads();
            cbuf = cbuf + 1 == NS ? 0 : cbuf + 1;
            lbuf = lbuf + 1 == NS ? 0 : lbuf + 1;
        }
    } else {
        for (int c = 0; c < nch; ++c) {
            wg_wait_vmcnt<(NS - 2) * LPT>();
            __syncthreads();
            issue(lbuf);
            compute(cbuf);
            cbuf = cbuf + 1 == NS ? 0 : cbuf + 1;
            lbuf = lbuf + 1 == NS ? 0 : lbuf + 1;
        }
    }
    wg_wait_vmcnt<0>();

    if (do_bias) {
        const int n = n0 + t % BN;
        if (n < p.N) atomicAdd(p.dbias + n, bsum);
    }
    const int taps = p.KH * p.KW;
    // Partial tile -> (a) split workspace with plain stores (reduced by wgrad_reduce_kernel) or (b) fp32 atomics.
    // fp32 atomics run at ~0.3 T elements/s chip-wide (one element per L2 channel clock,
    // benchmarks/probes/atomic_probe.hip), 4-5x below plain stores, so they are kept for the unsplit case only (where a
    // dependent load-add-store chain per lane would be slower than fire-and-forget atomics).
    const int mode = p.part ? 0 : 2;
    float* const dst = p.part ? p.part + (size_t)by * p.out_elems : p.dw;
    if (mode == 0 && p.epi_lds) {
        // Row-coalesced partial stores: the C/D layout gives 64-B pieces (16 lanes x 4 B) on 4 rows per store instruction;
        // staged through LDS (the operand stages are dead; everyone drained its DMA tail above) a wave writes 1 KB = two
        // full 512-B rows of the partial tile per instruction.  Done in n-halves when the fp32 tile exceeds the LDS.
        constexpr int EP_LD = BC + 4;
        constexpr int HALVES = ((size_t)BN * EP_LD * 4 > (size_t)NS * BUF_BYTES) ? 2 : 1;
        constexpr int ROWS = BN / HALVES;
        static_assert((size_t)ROWS * EP_LD * 4 <= (size_t)NS * BUF_BYTES, "partial tile does not fit the LDS stages");
        float* tile = reinterpret_cast<float*>(smem);
#pragma unroll
        for (int h = 0; h < HALVES; ++h) {
            __syncthreads();
            if (HALVES == 1 || wn == h) {
#pragma unroll
                for (int a = 0; a < TN; ++a)
#pragma unroll
                    for (int b = 0; b < TC; ++b)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            tile[((HALVES == 1 ? wn * (BN / 2) : 0) + a * 16 + lg * 4 + r) * EP_LD + wc * (BC / 2) + b * 16 + li] = acc[a][b][r];
            }
            __syncthreads();
            constexpr int PPR = BC / 4;                    // f32x4 pieces per row
#pragma unroll
            for (int i = 0; i < ROWS * PPR / 256; ++i) {
                const int idx = i * 256 + t;
                const int rl = idx / PPR, cl = (idx - rl * PPR) * 4;
                const int n = n0 + h * ROWS + rl, c = c0 + cl;
                if (n >= p.N || c >= p.SC) continue;
                *reinterpret_cast<f32x4*>(dst + ((size_t)n * taps + tap) * p.SC + c) = *reinterpret_cast<const f32x4*>(tile + rl * EP_LD + cl);
            }
        }
        return;
    }
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
                float* o = dst + ((size_t)n * taps + tap) * p.SC + c;
                if (mode == 0) { *o = v; continue; }
                if (p.scale) v *= p.scale[n];
                if (p.overwrite) *o = v; else atomicAdd(o, v);
            }
        }
    }
}

template <int BN, int BC, bool SIMPLE, int CR, int NS, int MINB>
__global__ __launch_bounds__(256, MINB) void conv_wgrad_dma_kernel(const bf16_t* __restrict__ dyp,
                                                                   const bf16_t* __restrict__ xp, const WgradArgs p) {
    wgrad_dma_body<BN, BC, SIMPLE, CR, NS>(dyp, xp, p, blockIdx.y * gridDim.x + blockIdx.x, gridDim.x, gridDim.y);
}

// Grouped launch: up to 24 independent weight-gradient problems (descriptors BY VALUE in the kernel arguments, so a captured
// hipGraph needs no device table).  A transformer layer's Linear weight gradients are ~50-250 workgroups each -- a fraction of
// the 256 CUs -- and none is on the backward-data dependency chain: queued and launched together they fill the chip.
// first[] is padded to multiples of 8 so the XCD map of a problem stays aligned with the hardware's round-robin.
struct WgradGroup { WgradArgs j[24]; int first[25]; int gx[24]; int gy[24]; int n; };
template <int BN, int BC, int CR, int NS, int MINB>
__global__ __launch_bounds__(256, MINB) void conv_wgrad_dma_grouped_kernel(const WgradGroup g) {
    int lo = 0, hi = g.n - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (g.first[mid] <= (int)blockIdx.x) lo = mid; else hi = mid - 1; }
    lo = __builtin_amdgcn_readfirstlane(lo);
    const int lin = (int)blockIdx.x - g.first[lo];
    if (lin >= g.gx[lo] * g.gy[lo]) return;
    wgrad_dma_body<BN, BC, true, CR, NS>(g.j[lo].dy, g.j[lo].x, g.j[lo], lin, g.gx[lo], g.gy[lo]);
}

struct ReduceGroup { const float* part[24]; float* dw[24]; const float* scale[24]; int out_elems[24], row_elems[24], nsplit[24], overwrite[24], first[25]; int n; };

// dw[i] += scale[i / row_elems] * sum_s part[s][i].  A workgroup owns 64 float4 columns; its 4 thread rows take the
// splits round-robin (4 independent loads in flight each) and meet in LDS.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                           const float* __restrict__ scale, int out_elems, int row_elems, int nsplit, int overwrite) {
    __shared__ f32x4 red[3][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int i = (blockIdx.x * 64 + tx) * 4;
    const bool ok = i < out_elems;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    if (ok) {
        int sidx = ty;
        for (; sidx + 12 < nsplit; sidx += 16) {
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(part + (size_t)sidx * out_elems + i);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(part + (size_t)(sidx + 4) * out_elems + i);
            const f32x4 v2 = *reinterpret_cast<const f32x4*>(part + (size_t)(sidx + 8) * out_elems + i);
            const f32x4 v3 = *reinterpret_cast<const f32x4*>(part + (size_t)(sidx + 12) * out_elems + i);
            a += (v0 + v1) + (v2 + v3);
        }
        for (; sidx < nsplit; sidx += 4) a += *reinterpret_cast<const f32x4*>(part + (size_t)sidx * out_elems + i);
    }
    if (ty) red[ty - 1][tx] = a;
    __syncthreads();
    if (ty || !ok) return;
    a += red[0][tx] + red[1][tx] + red[2][tx];
    if (scale) a *= scale[i / row_elems];
    f32x4* o = reinterpret_cast<f32x4*>(dw + i);
    *o = overwrite ? a : *o + a;
}

__global__ __launch_bounds__(256) void wgrad_reduce_grouped_kernel(const ReduceGroup g) {
    __shared__ f32x4 red[3][64];
    int lo = 0, hi = g.n - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (g.first[mid] <= (int)blockIdx.x) lo = mid; else hi = mid - 1; }
    lo = __builtin_amdgcn_readfirstlane(lo);
    const float* part = g.part[lo]; float* dw = g.dw[lo]; const float* scale = g.scale[lo];
    const int out_elems = g.out_elems[lo], row_elems = g.row_elems[lo], nsplit = g.nsplit[lo];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int i = (((int)blockIdx.x - g.first[lo]) * 64 + tx) * 4;
    const bool ok = i < out_elems;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    if (ok)
        for (int sidx = ty; sidx < nsplit; sidx += 4) a += *reinterpret_cast<const f32x4*>(part + (size_t)sidx * out_elems + i);
    if (ty) red[ty - 1][tx] = a;
    __syncthreads();
    if (ty || !ok) return;
    a += red[0][tx] + red[1][tx] + red[2][tx];
    if (scale) a *= scale[i / row_elems];
    f32x4* o = reinterpret_cast<f32x4*>(dw + i);
    *o = g.overwrite[lo] ? a : *o + a;
}

// M <= 16 rows (decoder-side Linears): plain outer-product accumulation, one thread per (n, 4 k's)
__global__ __launch_bounds__(256) void small_m_wgrad_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                            float* __restrict__ dw, const float* __restrict__ scale,
                                                            float* __restrict__ dbias, int M, int N, int K, int overwrite) {
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
        *o = overwrite ? a : *o + a;
        if (dbias && kk == 0) dbias[n] += gs;
    }
}

// Grouped form of small_m_wgrad_kernel: up to 64 independent (dy, x, dw) problems per launch, descriptors passed BY VALUE in
// the kernel arguments (3 KB) so a captured hipGraph needs no device-side table.  The decoder / query-encoder Linears over
// B*n_q token rows produce ~50 of these per step; none of them is on the backward-data dependency chain.
struct SmallJobs { rt_small_wgrad_job j[64]; int first[65]; int n; };
__global__ __launch_bounds__(256) void small_m_wgrad_grouped_kernel(const SmallJobs p) {
    __shared__ float sm[16];
    int lo = 0, hi = p.n - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (p.first[mid] <= (int)blockIdx.x) lo = mid; else hi = mid - 1; }
    const rt_small_wgrad_job& q = p.j[lo];
    const bf16_t* dy = (const bf16_t*)q.dy; const bf16_t* x = (const bf16_t*)q.x;
    const int k4 = q.K >> 2;
    const size_t total = (size_t)q.N * k4;
    const size_t i = (size_t)(blockIdx.x - p.first[lo]) * 256 + threadIdx.x;
    float ss = 0.f;                          // |dw after|^2 - |dw before|^2 of this thread's four elements (the clip norm, rt_sqnorm_finish)
    if (i < total) {
        const int kk = (int)(i % k4) * 4, n = (int)(i / k4);
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        float gs = 0.f;
        // all <= 16 rows' loads are requested before the first use (rows past M re-read the last row: cache hits, masked out below):
        // the launch is a few thousand workgroups of one load round trip each, not M of them
        float gv[16]; bf16x4 xv[16];
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            const int mm = m < q.M ? m : q.M - 1;
            gv[m] = (float)dy[(size_t)mm * q.N + n];
            xv[m] = *reinterpret_cast<const bf16x4*>(x + (size_t)mm * q.K + kk);
        }
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            if (m >= q.M) continue;
            gs += gv[m];
#pragma unroll
            for (int r = 0; r < 4; ++r) a[r] += gv[m] * (float)xv[m][r];
        }
        f32x4* o = reinterpret_cast<f32x4*>(q.dw + (size_t)n * q.K + kk);
        f32x4 fin;
        if (q.overwrite) {
            const f32x4 d = a * a; ss = (d[0] + d[1]) + (d[2] + d[3]);
            fin = a;
        } else {
            const f32x4 old = *o;
            const f32x4 d = a * (old + old + a); ss = (d[0] + d[1]) + (d[2] + d[3]);
            fin = old + a;
        }
        *o = fin;
        if (q.g16) *reinterpret_cast<bf16x4*>((bf16_t*)q.g16 + (size_t)n * q.K + kk) = bf16x4{(bf16_t)fin[0], (bf16_t)fin[1], (bf16_t)fin[2], (bf16_t)fin[3]};
        if (q.dbias && kk == 0) q.dbias[n] += gs;
    }
    if (q.sqacc) {                           // uniform per workgroup: a workgroup belongs to one job
        ss = rt_block_sum(ss, sm);
        if (threadIdx.x == 0) rt_sq_add(q.sqacc, blockIdx.x, ss);
    }
}

// overwrite semantics for launches whose tiles are accumulated with atomics by several workgroups: clear dw, then accumulate
static int wg_clear_for_atomics(WgradArgs& a, hipStream_t s) {
    if (!a.overwrite) return RT_OK;
    const size_t n = (size_t)a.N * a.KH * a.KW * a.SC;
    const hipError_t e = rt_zero_f32(a.dw, n, s);
    a.overwrite = 0;
    return e == hipSuccess ? RT_OK : (int)e;
}

template <int BN, int BC>
int launch_wgrad(WgradArgs a, int msplit, hipStream_t s) {
    { const int zrc = wg_clear_for_atomics(a, s); if (zrc != RT_OK) return zrc; }
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

template <int BN, int BC, int CR, int NS, int MINB>
int launch_wgrad_dma(WgradArgs a, int msplit, float* ws, long long ws_bytes, hipStream_t s) {
    const bool auto_split = msplit <= 0;
    const int nt = (a.N + BN - 1) / BN;
    a.c_tiles = (a.SC + BC - 1) / BC;
    const int taps = a.KH * a.KW;
    const int total_chunks = (a.M + CR - 1) / CR;
    const long long base_blocks = (long long)nt * a.c_tiles * taps;
    if (msplit <= 0) {
        static const int target = RT_TUNE("REFTR_WG_TARGET", 512);
        // each split costs one more partial tile through the workspace: small outputs take >= 256-row splits, larger
        // ones >= 512 (benchmarks/wgrad_probe.py)
        static const int minrows_env = RT_TUNE("REFTR_WG_MINROWS", 0);
        const int minrows = minrows_env ? minrows_env : ((long long)a.N * taps * a.SC * 4 <= (512 << 10) ? 256 : 512);
        long long want = (target + base_blocks - 1) / base_blocks;
        long long maxs = (long long)total_chunks * CR / minrows;
        if (maxs < 1) maxs = 1;
        if (want > maxs) want = maxs;
        if (want < 1) want = 1;
        msplit = (int)want;
    }
    if (msplit > total_chunks) msplit = total_chunks;
    if (msplit < 1) msplit = 1;
    const long long out_elems = (long long)a.N * taps * a.SC;
    if (auto_split && ws && msplit > 1) {           // keep the partials inside the caller's workspace
        const long long fit = ws_bytes / (out_elems * 4);
        if (fit < msplit) msplit = fit >= 2 ? (int)fit : msplit;
    }
    a.chunks_per_block = (total_chunks + msplit - 1) / msplit;
    const int gy = (total_chunks + a.chunks_per_block - 1) / a.chunks_per_block;
    const bool use_ws = ws && gy > 1 && (long long)gy * out_elems * 4 <= ws_bytes && (out_elems & 3) == 0;
    a.part = use_ws ? ws : nullptr; a.out_elems = (int)out_elems;
    if (!use_ws && gy > 1) { const int zrc = wg_clear_for_atomics(a, s); if (zrc != RT_OK) return zrc; }
    constexpr size_t smem = (size_t)NS * CR * (BN * 2 + BC * 2);
    const dim3 grid((unsigned)base_blocks, (unsigned)gy), block(256);
    const bool simple = (a.KH == 1 && a.KW == 1 && a.stride == 1 && a.pad == 0);
    if (simple) {
        if (smem > 65536) (void)hipFuncSetAttribute((const void*)conv_wgrad_dma_kernel<BN, BC, true, CR, NS, MINB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL((conv_wgrad_dma_kernel<BN, BC, true, CR, NS, MINB>), grid, block, smem, s, a.dy, a.x, a);
    } else {
        if (smem > 65536) (void)hipFuncSetAttribute((const void*)conv_wgrad_dma_kernel<BN, BC, false, CR, NS, MINB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL((conv_wgrad_dma_kernel<BN, BC, false, CR, NS, MINB>), grid, block, smem, s, a.dy, a.x, a);
    }
    RT_CHECK_LAUNCH();
    if (use_ws) {
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((out_elems / 4 + 63) / 64)), dim3(256), 0, s, ws, a.dw, a.scale,
                           (int)out_elems, taps * a.SC, gy, a.overwrite);
        RT_CHECK_LAUNCH();
    }
    return RT_OK;
}

}  // namespace

static int fill_wgrad_args(const rt_conv_wgrad_desc* d, WgradArgs& a);

// second-generation kernels (rt_wgrad2.hip)
bool rt_w2_eligible(const rt_conv_wgrad_desc& d);
int rt_w2_run(const rt_conv_wgrad_desc* descs, const int* idx, int n, float* workspace, long long workspace_bytes, hipStream_t s);
static int w2_enabled() {
    static const int e = RT_TUNE("REFTR_WG2", 1);
    return e;
}

// The Linear weight gradients a grouped launch accepts: plain [M,N]^T [M,K] products on the 128x128 DMA kernel.
static bool groupable(const rt_conv_wgrad_desc& d) {
    const long long M = (long long)d.B * d.DH * d.DW;
    return d.KH == 1 && d.KW == 1 && d.stride == 1 && d.pad == 0 && d.N >= 128 && d.SC >= 128 && (d.N & 7) == 0 && M > 16 &&
           d.variant == 0 && d.msplit <= 0;
}

static int wgrad_grouped_v1(const rt_conv_wgrad_desc* descs, int n, float* workspace, int64_t workspace_bytes, rt_stream_t stream);
static int conv_wgrad_impl(const rt_conv_wgrad_desc* d, rt_stream_t stream);

// Gradient-norm accounting for the launches WITHOUT an in-kernel contribution (first-generation kernels, the M <= 16 kernels): a pass
// with sign -1 over every dw that is accumulated onto (|before|^2) in front of the launches, +1 over every dw behind them.  Those
// matrices are the small ones (decoder / query-encoder / head Linears, ragged channel counts) and were just written: L2 hits.
static int sq_account(const rt_conv_wgrad_desc* descs, int n, bool after, hipStream_t s) {
    float* bufs[32]; long long cnts[32]; float signs[32];
    int m = 0;
    float* slots = nullptr;
    for (int i = 0; i <= n; ++i) {
        const bool last = i == n;
        if (!last) {
            const rt_conv_wgrad_desc& d = descs[i];
            if (!d.sqacc) continue;
            // a dw that appears more than once in the group (shared weights, a second accumulate queued before the flush) is
            // accounted ONCE, by its first occurrence: -|before|^2 unless that first launch overwrites, +|after|^2 behind them all
            int first = i;
            for (int j = 0; j < i; ++j) if (descs[j].sqacc && descs[j].dw == d.dw) { first = j; break; }
            if (first != i || (!after && d.overwrite)) continue;
            if (slots && d.sqacc != slots) return RT_ERR_BADARG;     // one accumulator per call
            slots = d.sqacc;
            bufs[m] = d.dw; cnts[m] = (long long)d.N * d.KH * d.KW * d.SC; signs[m] = after ? 1.f : -1.f; ++m;
        }
        if (m == 32 || (last && m > 0)) {
            const int rc = rt_sq_pass(bufs, cnts, signs, m, slots, s);
            if (rc != RT_OK) return rc;
            m = 0;
        }
    }
    return RT_OK;
}
// the bf16 exchange twins of the same launches (behind them): one rounding pass over what they wrote
static int twin_account(const rt_conv_wgrad_desc* descs, int n, hipStream_t s) {
    float* bufs[32]; void* twins[32]; long long cnts[32];
    int m = 0;
    for (int i = 0; i <= n; ++i) {
        const bool last = i == n;
        if (!last && descs[i].g16) { bufs[m] = descs[i].dw; twins[m] = descs[i].g16; cnts[m] = (long long)descs[i].N * descs[i].KH * descs[i].KW * descs[i].SC; ++m; }
        if (m == 32 || (last && m > 0)) {
            const int rc = rt_round_pass(bufs, twins, cnts, m, s);
            if (rc != RT_OK) return rc;
            m = 0;
        }
    }
    return RT_OK;
}

extern "C" int rt_conv_wgrad_grouped(const rt_conv_wgrad_desc* descs, int n, float* workspace, int64_t workspace_bytes,
                                     rt_stream_t stream) {
    if (!descs || n <= 0) return RT_ERR_BADARG;
    hipStream_t s = (hipStream_t)stream;
    if (w2_enabled()) {
        // second generation: every eligible problem of the group (1x1 / 3x3 convolutions and Linears alike) shares the grouped
        // v2 launches and their group-level split policy; the rest (ragged channel counts, pinned variants) goes on below
        static int idx[4096];
        static rt_conv_wgrad_desc rest[4096];
        int m = 0, nr = 0;
        for (int i = 0; i < n && i < 4096; ++i) {
            if (!descs[i].dy || !descs[i].x || !descs[i].dw) return RT_ERR_BADARG;
            if (rt_w2_eligible(descs[i])) idx[m++] = i; else rest[nr++] = descs[i];
        }
        if (n > 4096) return RT_ERR_UNSUPPORTED;
        if (m > 0) {
            const int rc = rt_w2_run(descs, idx, m, workspace, workspace ? (long long)workspace_bytes : 0, s);
            if (rc != RT_OK) return rc;
        }
        // (the v2 launches above have consumed the workspace; the first-generation group below re-uses it -- same stream, in order)
        if (nr == 0) return RT_OK;
        int rc = sq_account(rest, nr, false, s);
        if (rc == RT_OK) rc = wgrad_grouped_v1(rest, nr, workspace, workspace_bytes, stream);
        if (rc == RT_OK) rc = sq_account(rest, nr, true, s);
        if (rc == RT_OK) rc = twin_account(rest, nr, s);
        return rc;
    }
    int rc = sq_account(descs, n, false, s);
    if (rc == RT_OK) rc = wgrad_grouped_v1(descs, n, workspace, workspace_bytes, stream);
    if (rc == RT_OK) rc = sq_account(descs, n, true, s);
    if (rc == RT_OK) rc = twin_account(descs, n, s);
    return rc;
}

static int wgrad_grouped_v1(const rt_conv_wgrad_desc* descs, int n, float* workspace, int64_t workspace_bytes, rt_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    constexpr int BN = 128, BC = 128, CR = 32, NS = 3, MINB = 2;
    static WgradGroup g; static ReduceGroup r;            // host-side staging (single caller thread per device, see header)
    g.n = 0; r.n = 0;
    int blocks = 0, rblocks = 0;
    long long ws_off = 0;
    auto flush = [&]() -> int {
        if (g.n > 0) {
            g.first[g.n] = blocks;
            constexpr size_t smem = (size_t)NS * CR * (BN * 2 + BC * 2);
            hipLaunchKernelGGL((conv_wgrad_dma_grouped_kernel<BN, BC, CR, NS, MINB>), dim3(blocks), dim3(256), smem, s, g);
            RT_CHECK_LAUNCH();
        }
        if (r.n > 0) {
            r.first[r.n] = rblocks;
            hipLaunchKernelGGL(wgrad_reduce_grouped_kernel, dim3(rblocks), dim3(256), 0, s, r);
            RT_CHECK_LAUNCH();
        }
        g.n = 0; r.n = 0; blocks = 0; rblocks = 0; ws_off = 0;
        return RT_OK;
    };
    for (int i = 0; i < n; ++i) {
        const rt_conv_wgrad_desc& d = descs[i];
        if (!groupable(d)) {                               // anything else keeps its own launch
            const int rc = conv_wgrad_impl(&d, stream);
            if (rc != RT_OK) return rc;
            continue;
        }
        WgradArgs a;
        const int rc = fill_wgrad_args(&d, a);
        if (rc != RT_OK) return rc;
        // same split rule as the single launch (launch_wgrad_dma)
        const int nt = (a.N + BN - 1) / BN;
        a.c_tiles = (a.SC + BC - 1) / BC;
        const int total_chunks = (a.M + CR - 1) / CR;
        const long long base_blocks = (long long)nt * a.c_tiles;
        const long long out_elems = (long long)a.N * a.SC;
        const int minrows = out_elems * 4 <= (512 << 10) ? 256 : 512;
        static const int gtarget = RT_TUNE("REFTR_WG_TARGET", 512);
        long long want = (gtarget + base_blocks - 1) / base_blocks, maxs = (long long)total_chunks * CR / minrows;
        if (maxs < 1) maxs = 1;
        if (want > maxs) want = maxs;
        int msplit = (int)(want < 1 ? 1 : want);
        if (msplit > total_chunks) msplit = total_chunks;
        a.chunks_per_block = (total_chunks + msplit - 1) / msplit;
        const int gy = (total_chunks + a.chunks_per_block - 1) / a.chunks_per_block;
        const long long need = gy > 1 ? (long long)gy * out_elems * 4 : 0;
        if (g.n == 24 || (need > 0 && workspace && ws_off + need > workspace_bytes && g.n > 0)) {
            const int frc = flush();
            if (frc != RT_OK) return frc;
        }
        const bool use_ws = gy > 1 && workspace && ws_off + need <= workspace_bytes;
        a.part = use_ws ? workspace + ws_off / 4 : nullptr; a.out_elems = (int)out_elems;
        if (!use_ws && gy > 1) { const int zrc = wg_clear_for_atomics(a, s); if (zrc != RT_OK) return zrc; }
        g.j[g.n] = a; g.first[g.n] = blocks; g.gx[g.n] = (int)base_blocks; g.gy[g.n] = gy; ++g.n;
        blocks += (int)((base_blocks * gy + 7) / 8 * 8);
        if (use_ws) {
            r.part[r.n] = a.part; r.dw[r.n] = a.dw; r.scale[r.n] = a.scale; r.out_elems[r.n] = (int)out_elems; r.row_elems[r.n] = a.SC;
            r.nsplit[r.n] = gy; r.overwrite[r.n] = a.overwrite; r.first[r.n] = rblocks; ++r.n;
            rblocks += (int)((out_elems / 4 + 63) / 64);
            ws_off += (need + 255) / 256 * 256;
        }
    }
    return flush();
}

static int fill_wgrad_args(const rt_conv_wgrad_desc* d, WgradArgs& a) {
    if (!d || !d->dy || !d->x || !d->dw) return RT_ERR_BADARG;
    if (d->SC <= 0 || (d->SC & 15) || (d->N & 3) || d->N <= 0) return RT_ERR_UNSUPPORTED;
    if (d->KH <= 0 || d->KW <= 0 || d->B <= 0 || d->DH <= 0 || d->DW <= 0 || d->stride <= 0) return RT_ERR_BADARG;
    if (d->dil > 1 && d->stride != 1) return RT_ERR_UNSUPPORTED;
    a.dy = (const bf16_t*)d->dy; a.x = (const bf16_t*)d->x; a.dw = d->dw; a.scale = d->scale; a.dbias = d->dbias;
    a.B = d->B; a.SH = d->SH; a.SW = d->SW; a.SC = d->SC; a.DH = d->DH; a.DW = d->DW; a.N = d->N;
    a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad = d->pad; a.dil = d->dil > 1 ? d->dil : 1;
    const long long M = (long long)d->B * d->DH * d->DW;
    if (M > 0x7fffffffLL / 4) return RT_ERR_UNSUPPORTED;
    if (M * d->N >= 0x3fffffffLL || (long long)d->B * d->SH * d->SW * d->SC >= 0x3fffffffLL) return RT_ERR_UNSUPPORTED;
    a.M = (int)M; a.chunks_per_block = 0; a.c_tiles = 0; a.part = nullptr; a.out_elems = 0; a.overwrite = d->overwrite ? 1 : 0;
    static const int xcd_env = RT_TUNE("REFTR_XCD", 1);
    a.xcd = xcd_env;
    static const int early_env = RT_TUNE("REFTR_EARLY", 3);
    a.early = early_env & 2 ? 1 : 0;
    static const int epi_env = RT_TUNE("REFTR_EPI", 3);
    a.epi_lds = epi_env & 2 ? 1 : 0;
    a.dy_bytes = (unsigned)(M * d->N * 2);
    a.x_bytes = (unsigned)((long long)d->B * d->SH * d->SW * d->SC * 2);
    return RT_OK;
}

extern "C" int rt_conv_wgrad(const rt_conv_wgrad_desc* d, rt_stream_t stream) {
    if (!d) return RT_ERR_BADARG;
    if (!d->sqacc && !d->g16) return conv_wgrad_impl(d, stream);
    const long long M = (long long)d->B * d->DH * d->DW;
    const bool v2 = w2_enabled() && rt_w2_eligible(*d) && !RT_TUNE_SET("REFTR_WGV") &&
                    !(M <= 16 && d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad == 0 && (d->SC & 3) == 0);
    if (v2) return conv_wgrad_impl(d, stream);           // the second-generation kernels account in their epilogues
    int rc = sq_account(d, 1, false, (hipStream_t)stream);
    if (rc == RT_OK) rc = conv_wgrad_impl(d, stream);
    if (rc == RT_OK) rc = sq_account(d, 1, true, (hipStream_t)stream);
    if (rc == RT_OK) rc = twin_account(d, 1, (hipStream_t)stream);
    return rc;
}

static int conv_wgrad_impl(const rt_conv_wgrad_desc* d, rt_stream_t stream) {
    WgradArgs a;
    const int frc = fill_wgrad_args(d, a);
    if (frc != RT_OK) return frc;
    hipStream_t s = (hipStream_t)stream;
    if (a.M <= 16 && a.KH == 1 && a.KW == 1 && a.stride == 1 && a.pad == 0 && (a.SC & 3) == 0) {
        const size_t total = (size_t)a.N * (a.SC >> 2);
        int blocks = (int)((total + 255) / 256); if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(small_m_wgrad_kernel, dim3(blocks), dim3(256), 0, s, a.dy, a.x, a.dw, a.scale, a.dbias, a.M, a.N, a.SC, a.overwrite);
        RT_CHECK_LAUNCH();
        return RT_OK;
    }
    if (w2_enabled() && rt_w2_eligible(*d) && !RT_TUNE_SET("REFTR_WGV")) {       // a group of one
        const int zero = 0;
        return rt_w2_run(d, &zero, 1, d->workspace, d->workspace ? (long long)d->workspace_bytes : 0, s);
    }
    // variant: 0 = LDS-DMA kernels (default), 9 = register-staged kernel, 1..5 = pinned 128x128 DMA staging shapes
    static const int wgv_env = RT_TUNE("REFTR_WGV", 0);
    const int wgv = d->variant > 0 ? d->variant : wgv_env;
    float* ws = d->workspace; long long wsb = d->workspace ? d->workspace_bytes : 0;
    static const int no_ws = RT_TUNE("REFTR_WG_NOWS", 0);
    if (no_ws) { ws = nullptr; wsb = 0; }
    if (wgv != 9 && (a.N & 7) == 0) {
        if (a.N >= 128 && a.SC >= 128) {
            switch (wgv) {
                case 1: return launch_wgrad_dma<128, 128, 64, 2, 2>(a, d->msplit, ws, wsb, s);
                case 2: return launch_wgrad_dma<128, 128, 32, 4, 2>(a, d->msplit, ws, wsb, s);
                case 3: return launch_wgrad_dma<128, 128, 64, 3, 2>(a, d->msplit, ws, wsb, s);
                case 5: return launch_wgrad_dma<128, 128, 64, 4, 2>(a, d->msplit, ws, wsb, s);
                default: return launch_wgrad_dma<128, 128, 32, 3, 2>(a, d->msplit, ws, wsb, s);
            }
        }
        if (a.N >= 128) return launch_wgrad_dma<128, 64, 64, 3, 2>(a, d->msplit, ws, wsb, s);
        if (a.SC >= 128) return launch_wgrad_dma<64, 128, 64, 3, 2>(a, d->msplit, ws, wsb, s);
        return launch_wgrad_dma<64, 64, 64, 4, 2>(a, d->msplit, ws, wsb, s);
    }
    if (a.N >= 128 && a.SC >= 128) return launch_wgrad<128, 128>(a, d->msplit, s);
    if (a.N >= 128) return launch_wgrad<128, 64>(a, d->msplit, s);
    if (a.SC >= 128) return launch_wgrad<64, 128>(a, d->msplit, s);
    return launch_wgrad<64, 64>(a, d->msplit, s);
}

extern "C" int rt_small_wgrad_grouped(const rt_small_wgrad_job* jobs, int njobs, rt_stream_t stream) {
    if (!jobs || njobs <= 0) return RT_ERR_BADARG;
    for (int base = 0; base < njobs; base += 64) {
        SmallJobs p;
        p.n = njobs - base < 64 ? njobs - base : 64;
        int blocks = 0;
        for (int i = 0; i < p.n; ++i) {
            const rt_small_wgrad_job& q = jobs[base + i];
            if (!q.dy || !q.x || !q.dw || q.M <= 0 || q.M > 16 || q.N <= 0 || q.K <= 0 || (q.K & 3)) return RT_ERR_BADARG;
            p.j[i] = q; p.first[i] = blocks;
            blocks += (int)(((size_t)q.N * (q.K >> 2) + 255) / 256);
        }
        p.first[p.n] = blocks;
        hipLaunchKernelGGL(small_m_wgrad_grouped_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
        RT_CHECK_LAUNCH();
    }
    return RT_OK;                           // (the gradient-norm contribution is taken inside the kernel)
}
