// Weight-gradient GEMM, second generation: dw[n][tap][c] (+)= scale[n] * sum_m dy[m, n] * xg[m, (tap, c)].
//
// The contraction runs over pixels / tokens (m = thousands .. hundreds of thousands) and the output is tiny (N x taps x C),
// so the questions that decide the speed are (1) how many bytes a workgroup pulls through L2 per flop, (2) how the m axis
// is cut and (3) where a workgroup's operand rows come from.  Differences to the first generation (rt_wgrad.hip, kept for
// ragged channel counts, short row counts and as the A/B baseline):
//   * 8 waves per workgroup on tiles up to 256 x 256 (2x the flop per staged byte of 128 x 128), v_mfma_f32_32x32x16_bf16
//     (half the LDS fragment traffic per flop of 16x16x32), accumulators 128 VGPRs per lane;
//   * a whole GROUP of problems (a ResNet stage's 1x1 and 3x3 convolutions, a transformer section's Linears) shares one
//     launch and ONE split policy: the m axis is only cut when the group has fewer tile tasks than the chip has workgroup
//     slots.  Unsplit tiles accumulate straight into dw (one owner per element: plain read-modify-write, no atomics, no
//     partial tiles through memory, no reduction launch); split tiles go through the workspace and one grouped reduction;
//   * XCD placement: the linear task order (problem, row split, tile / tap -- the tasks that read the same dy / x rows are
//     adjacent) is dealt to the 8 XCDs in contiguous runs of gridDim.x / 8 tasks (block id -> rt_xcd_remap) instead of the
//     hardware's round-robin: the same number of tasks per XCD as the plain map (an XCD is 32 CUs: one task more than its
//     share is a whole extra round), but a split's tiles and taps share one XCD's L2 -- fabric fetches 3.39 -> 1.89 GB over
//     the ResNet groups at equal time, the transformer groups 15 % faster.  (A table that put whole units on XCDs round-robin,
//     REFTR_W2_XCD=1, cut the fetches as much and lost 30 % to that imbalance.)
//   * software pipeline: one barrier per 32-row chunk, NS-1 chunks of LDS-DMA in flight, and the fragment reads of the next
//     16-row step (including the first step of the NEXT chunk) are issued before the MFMAs of the current one;
//   * wave-specialised L2 prefetch: waves 0-3 issue the stage DMAs; waves 4-7 touch every 128-B line of the chunk `pf`
//     iterations ahead of its DMA with a 4-byte LDS-DMA into a scratch word and NEVER wait for those (vmcnt completes in
//     order, so a wave that also had to wait for stage DMAs could not run its touches ahead), so the stage requests find their
//     lines in L2: a miss costs 2-3 us under load and only NS-1 stages = 96 KB per CU can be in flight, a touch needs no LDS.
// Operand staging, as before: both operands keep their natural [pixel][channel] layout, go global -> LDS by DMA
// (buffer_load_dwordx4 ... lds, 16 B per lane, lane-linear), and become MFMA fragments through the LDS transpose read
// (ds_read_b64_tr_b16).  The DMA cannot scatter, so the bank swizzle (16-B slot ^= (row & 3) << 2: the four rows x 64 B a
// 32-lane half of the transpose read touches land on four different 64-B bank groups) is applied to the SOURCE address.
#include "rt_common.h"
#include <stdlib.h>
#include <stdio.h>

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) int w2_i32x4;

struct W2Prob {
    const bf16_t* dy; const bf16_t* x; float* dw; const float* scale; float* dbias; float* part; float* sqacc; bf16_t* g16;
    int SH, SW, SC, DH, DW, N, KH, KW, stride, pad, M, dil;
    int n_tiles, c_tiles, tiles, splits, chunks_per_split, out_elems;
    unsigned dy_bytes, x_bytes;
    int simple, accumulate, xrot, cfg;
};
constexpr int W2_MAXP = 20;
// cum[x][i]: workgroups of problems 0 .. i-1 that run on XCD x (block id b -> XCD b & 7, position b >> 3 in its order)
struct W2Group { W2Prob p[W2_MAXP]; int cum[8][W2_MAXP + 1]; int n; int xcd; int abl; int pf; int remap; };

template <int N> __device__ __forceinline__ void w2_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ w2_i32x4 w2_rsrc(const void* ptr, unsigned bytes) {
    const uint64_t a = (uint64_t)ptr;
    return w2_i32x4{(int)(uint32_t)a, (int)(uint32_t)(a >> 32), (int)bytes, 0x00020000};
}
__device__ __forceinline__ w2_i32x4 w2_rsrc_uniform(const void* ptr, unsigned bytes) {     // forces the descriptor into SGPRs
    const uint64_t a = (uint64_t)ptr;
    return w2_i32x4{__builtin_amdgcn_readfirstlane((int)(uint32_t)a), __builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32)),
                    __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000};
}
__device__ __forceinline__ void w2_dma16(const w2_i32x4 rsrc, unsigned lds_base, int voff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                 ::"s"(lds_base), "v"(voff), "s"(rsrc)
                 : "memory", "m0");
}
__device__ __forceinline__ void w2_dma4(const w2_i32x4 rsrc, unsigned lds_base, int voff) {      // 4 B per lane: the L2 prefetch touch
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds"
                 ::"s"(lds_base), "v"(voff), "s"(rsrc)
                 : "memory", "m0");
}
__device__ __forceinline__ int w2_swz(int row) { return (row & 3) << 2; }
__device__ __forceinline__ bf16x8 w2_frag2(const unsigned char* base, int off0, int off1) {
    typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_ptr;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(base + off0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(base + off1));
    union { struct { s16x4 a, b; } s; bf16x8 v; } u;
    u.s.a = lo; u.s.b = hi;
    return u.v;
}

// BN x BC output tile, WN x WC waves (WN * WC == 8), CR contraction rows per stage, NS stages.
// TAPS = 3 (3x3, stride 1, pad 1 only): a workgroup accumulates the three kw taps of ONE kernel row kh of its output tile.  For a
// chunk of 32 output pixels m0 .. m0+31 the source pixel of (m, kw) is the flat pixel index m + (kh-1) W + (kw-1) whenever it lies
// inside the image row / image, so the x operand of all three taps is ONE contiguous slab of 34 rows (staged once, as plainly as a
// 1x1 problem's rows) and tap kw reads it shifted by kw rows; the dy fragments are shared by the three taps.  Where the source
// would fall outside (first / last column, first / last image row) the flat index names some other pixel: those (m, kw) terms are
// removed by AND-ing the x fragments with a row mask (ballot of the validity of the chunk's 32 rows per tap, expanded through a
// 256-entry LDS table: byte of 8 row bits -> 8 x 16-bit lanes).  3x the flop per staged byte and per dy fragment read.
template <int BN, int BC, int WN, int WC, int CR, int NS, bool SIMPLE, int TAPS = 1>
__device__ __forceinline__ void w2_body(const W2Prob& p, const int split, const int tile, const int abl, const int pf) {
    static_assert(WN * WC == 8, "8 waves");
    constexpr bool FUSED = TAPS == 3;
    static_assert(TAPS == 1 || (TAPS == 3 && !SIMPLE), "taps");
    constexpr int BROWS = FUSED ? 48 : CR;               // staged x rows: the 34-row window rounded up to whole 4-KB DMA rounds
    constexpr int WTN = BN / WN, WTC = BC / WC;          // wave tile
    constexpr int TN = WTN / 32, TC = WTC / 32;          // 32x32 MFMA blocks per wave
    static_assert(TN >= 1 && TC >= 1, "wave tile too small");
    constexpr int RBA = BN * 2, RBB = BC * 2;            // LDS row bytes
    constexpr int A_BYTES = CR * RBA, B_BYTES = BROWS * RBB, BUF_BYTES = A_BYTES + B_BYTES;
    constexpr int AJ = A_BYTES / 4096, BJ = B_BYTES / 4096;      // DMA instructions per thread of waves 0-3 per stage (4 waves x 1 KB)
    static_assert(AJ >= 1 && BJ >= 1 && A_BYTES % 4096 == 0 && B_BYTES % 4096 == 0, "stage must be whole 4-KB rounds");
    constexpr int LPT = AJ + BJ;
    constexpr int KS = CR / 16;
    static_assert(KS == 2, "the software pipeline below is written for two 16-row steps per chunk");
    constexpr int OOB = 0x7fffffff;
    constexpr int PFA = CR * RBA / 128, PFB = CR * RBB / 128;    // 128-B lines of a chunk (one per thread of waves 4-7)
    static_assert(PFA + PFB <= 256, "prefetch lines");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef __attribute__((address_space(3))) void* lds_ptr;

    const int t = threadIdx.x;
    const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wn = wave % WN, wc = wave / WN;

    const int tile_n = tile % p.n_tiles;
    const int rest = tile / p.n_tiles;
    const int tile_c = rest % p.c_tiles;
    const int tap = rest / p.c_tiles;
    const int kh = tap / p.KW, kw = tap - kh * p.KW;
    const int n0 = tile_n * BN, c0 = tile_c * BC;

    const int total_chunks = (p.M + CR - 1) / CR;
    const int chunk_begin = split * p.chunks_per_split;
    int chunk_end = chunk_begin + p.chunks_per_split;
    if (chunk_end > total_chunks) chunk_end = total_chunks;
    const int nch = chunk_end - chunk_begin;

    // ---- DMA source offsets (loop-invariant, relative to the chunk's first row); LDS destination is lane-linear
    int voff_a[AJ];
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
        const int byte = (j * 4 + (wave & 3)) * 1024 + lane * 16;
        const int r = byte / RBA, sl = ((byte % RBA) >> 4) ^ w2_swz(r);
        const int n = n0 + sl * 8;
        voff_a[j] = n < p.N ? (r * p.N + n) * 2 : OOB;
    }
    int voff_b[BJ], b_r[BJ], gb[BJ], gy[BJ], gx[BJ];
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
        const int byte = (j * 4 + (wave & 3)) * 1024 + lane * 16;
        const int r = byte / RBB, sl = ((byte % RBB) >> 4) ^ w2_swz(r);
        const int c = c0 + sl * 8;
        b_r[j] = r;
        if (SIMPLE) { voff_b[j] = c < p.SC ? (r * p.SC + c) * 2 : OOB; gb[j] = gy[j] = gx[j] = 0; }
        else if (FUSED) { voff_b[j] = (c < p.SC && r < CR + 4) ? (r * p.SC + c) * 2 : OOB; gb[j] = gy[j] = gx[j] = 0; }
        else {
            voff_b[j] = c < p.SC ? c * 2 : OOB;
            const int m = chunk_begin * CR + r;
            gx[j] = m % p.DW; const int tmp = m / p.DW; gy[j] = tmp % p.DH; gb[j] = tmp / p.DH;
        }
    }
    // ---- L2 prefetch touch (waves 4-7): thread u = t - 256 owns the u-th 128-B line of a chunk (dy lines first, then x lines).
    // For the gather path the x rows of a stride-1 tap are the chunk's rows shifted by (kh - pad) * SW + (kw - pad) pixels
    // (image borders ignored: a few lines too many); stride-2 gathers are not prefetched.
    int voff_pf = OOB;
    const bool pf_is_a = (wave - 4) * 64 < PFA;            // wave-uniform: PFA and PFB are multiples of 64
    if (t >= 256) {
        const int u = t - 256;
        if (u < PFA) {
            const int r = u / (RBA / 128), col = n0 * 2 + (u % (RBA / 128)) * 128;
            if (col < p.N * 2) voff_pf = r * p.N * 2 + col;
        } else if (u < PFA + PFB) {
            const int v = u - PFA;
            const int r = v / (RBB / 128), col = c0 * 2 + (v % (RBB / 128)) * 128;
            if (col < p.SC * 2 && (SIMPLE || p.stride == 1)) voff_pf = r * p.SC * 2 + col;
        }
    }
    const int pf_shift = SIMPLE ? 0 : ((kh * p.dil - p.pad) * p.SW + (kw * p.dil - p.pad)) * p.SC * 2;       // bytes

    f32x16 acc[TAPS][TN][TC];
#pragma unroll
    for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
        for (int a = 0; a < TN; ++a)
#pragma unroll
            for (int b = 0; b < TC; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[tp][a][b][r] = 0.f;

    const bool do_bias = p.dbias && tile_c == 0 && tap == 0;      // fused: tap = kh, kernel row 0 holds tap 0

    // ---- fused taps: row-mask table (byte of 8 row bits -> eight 16-bit lanes) and the (h, w) of this lane's row of the chunk
    // whose fragments are read next
    uint4* const mtab = reinterpret_cast<uint4*>(smem + NS * BUF_BYTES);
    int mw = 0, mh = 0;
    unsigned mbits[3] = {0u, 0u, 0u};
    if (FUSED) {
        if (t < 256) {
            uint4 e;
            e.x = ((t & 1) ? 0xFFFFu : 0u) | ((t & 2) ? 0xFFFF0000u : 0u);
            e.y = ((t & 4) ? 0xFFFFu : 0u) | ((t & 8) ? 0xFFFF0000u : 0u);
            e.z = ((t & 16) ? 0xFFFFu : 0u) | ((t & 32) ? 0xFFFF0000u : 0u);
            e.w = ((t & 64) ? 0xFFFFu : 0u) | ((t & 128) ? 0xFFFF0000u : 0u);
            mtab[t] = e;
        }
        const int m = (split * p.chunks_per_split) * CR + (lane & 31);
        mw = m % p.DW; mh = (m / p.DW) % p.DH;
    }
    auto row_masks = [&]() __attribute__((always_inline)) {       // validity of (row, kw) for the 32 rows of the chunk at (mh, mw)
        const bool vrow = (unsigned)(mh + tap - 1) < (unsigned)p.DH;
#pragma unroll
        for (int q = 0; q < 3; ++q)
            mbits[q] = (unsigned)__ballot(vrow && (unsigned)(mw + q - 1) < (unsigned)p.DW);
    };
    auto next_rows = [&]() __attribute__((always_inline)) {
        mw += CR;
        while (mw >= p.DW) { mw -= p.DW; if (++mh >= p.DH) mh = 0; }
    };
    float bsum = 0.f;

    const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr)smem + (unsigned)(wave & 3) * 1024u;
    const unsigned lds_pf = (unsigned)(uintptr_t)(lds_ptr)smem + (unsigned)(NS * BUF_BYTES) + (unsigned)(wave & 3) * 256u;
    const w2_i32x4 rs_x_abs = w2_rsrc(p.x, p.x_bytes);
    int lc = chunk_begin;

    auto issue_dma = [&](int stage) __attribute__((always_inline)) {             // waves 0-3
        const unsigned bA = lds0 + stage * BUF_BYTES, bB = bA + A_BYTES;
        const unsigned aoff = (unsigned)lc * (unsigned)(CR * 2) * (unsigned)p.N;
        const w2_i32x4 rs_dy = w2_rsrc(reinterpret_cast<const unsigned char*>(p.dy) + aoff, p.dy_bytes - aoff);
#pragma unroll
        for (int j = 0; j < AJ; ++j) w2_dma16(rs_dy, bA + j * 4096, voff_a[j]);
        const bool last = (lc + 1 >= chunk_end);
        if (FUSED) {
            // rows g0 .. g0 + 33 of x (flat pixel index, may start before the tensor / end behind it: those rows read as zero)
            const int base = (lc * CR + (tap - 1) * p.DW - 1) * p.SC * 2;
#pragma unroll
            for (int j = 0; j < BJ; ++j) {
                int off = voff_b[j] == OOB ? OOB : base + voff_b[j];
                if (off < 0) off = OOB;
                w2_dma16(rs_x_abs, bB + j * 4096, off);
            }
        } else if (SIMPLE) {
            const unsigned xoff = (unsigned)lc * (unsigned)(CR * 2) * (unsigned)p.SC;
            const w2_i32x4 rs_x = w2_rsrc(reinterpret_cast<const unsigned char*>(p.x) + xoff, p.x_bytes - xoff);
#pragma unroll
            for (int j = 0; j < BJ; ++j) w2_dma16(rs_x, bB + j * 4096, voff_b[j]);
        } else {
#pragma unroll
            for (int j = 0; j < BJ; ++j) {
                const int m = lc * CR + b_r[j];
                const int sy = gy[j] * p.stride - p.pad + kh * p.dil, sx = gx[j] * p.stride - p.pad + kw * p.dil;
                const bool ok = m < p.M && (unsigned)sy < (unsigned)p.SH && (unsigned)sx < (unsigned)p.SW && voff_b[j] != OOB;
                const int pix = (gb[j] * p.SH + sy) * p.SW + sx;
                w2_dma16(rs_x_abs, bB + j * 4096, ok ? pix * p.SC * 2 + voff_b[j] : OOB);
                if (!last) {
                    gx[j] += CR;
                    while (gx[j] >= p.DW) { gx[j] -= p.DW; if (++gy[j] >= p.DH) { gy[j] = 0; ++gb[j]; } }
                }
            }
        }
        if (!last) ++lc;
    };
    auto issue_touch = [&]() __attribute__((always_inline)) {                     // waves 4-7: chunk lc + pf, never waited for
        const int pc = lc + pf;
        const bool on = pc < chunk_end;                     // wave-uniform: the descriptor below must live in SGPRs
        if (pf_is_a) {
            const unsigned off = (unsigned)pc * (unsigned)(CR * 2) * (unsigned)p.N;
            const w2_i32x4 rs = w2_rsrc_uniform(reinterpret_cast<const unsigned char*>(p.dy) + (on ? off : 0u), on ? p.dy_bytes - off : 0u);
            w2_dma4(rs, lds_pf, voff_pf);
        } else {
            const long long off = (long long)pc * (CR * 2) * p.SC + pf_shift;
            const bool in = on && off >= 0 && off < (long long)p.x_bytes;
            const w2_i32x4 rs = w2_rsrc_uniform(reinterpret_cast<const unsigned char*>(p.x) + (in ? off : 0), in ? p.x_bytes - (unsigned)off : 0u);
            w2_dma4(rs, lds_pf, voff_pf);
        }
        if (lc + 1 < chunk_end) ++lc;
    };
    const bool dma_wave = wave < 4;
    auto issue = [&](int stage) __attribute__((always_inline)) {
        if (dma_wave) issue_dma(stage);
        else if (pf > 0) issue_touch();
    };

    // ---- transpose-read addresses.  16-lane group g = lane >> 4: channel block (g & 1) * 16, k block g >> 1 (8 rows);
    // lane i of a group supplies row (i >> 2) of a 4-row block and the 8 bytes of channels 4 * (i & 3) .. + 3.
    const int g = lane >> 4, li = lane & 15;
    const int row0 = (g >> 1) * 8 + (li >> 2);
    const int colb = (g & 1) * 32 + (li & 3) * 8;
    int addr_a[TN][2], addr_b[TC][2];
#pragma unroll
    for (int a = 0; a < TN; ++a) {
        const int cb = (wn * WTN + a * 32) * 2 + colb;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int r = row0 + 4 * h;
            addr_a[a][h] = r * RBA + ((((cb >> 4) ^ w2_swz(r)) << 4) | (cb & 15));
        }
    }
#pragma unroll
    for (int b = 0; b < TC; ++b) {
        const int cb = (wc * WTC + b * 32) * 2 + colb;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int r = row0 + 4 * h;
            addr_b[b][h] = r * RBB + ((((cb >> 4) ^ w2_swz(r)) << 4) | (cb & 15));
        }
    }
    int addr_b3[FUSED ? 3 : 1][TC][2];               // fused: tap kw reads the window kw rows further down (the swizzle follows the row)
    if (FUSED) {
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int b = 0; b < TC; ++b) {
                const int cb = (wc * WTC + b * 32) * 2 + colb;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int r = row0 + 4 * h + q;
                    addr_b3[q][b][h] = r * RBB + ((((cb >> 4) ^ w2_swz(r)) << 4) | (cb & 15));
                }
            }
    }
    auto load_frags = [&](int stage, int kk, bf16x8 (&af)[TN], bf16x8 (&bfr)[TAPS][TC]) __attribute__((always_inline)) {
        const unsigned char* bA = smem + stage * BUF_BYTES;
        const unsigned char* bB = bA + A_BYTES;
#pragma unroll
        for (int a = 0; a < TN; ++a) af[a] = w2_frag2(bA, addr_a[a][0] + kk * 16 * RBA, addr_a[a][1] + kk * 16 * RBA);
        if (FUSED) {
#pragma unroll
            for (int q = 0; q < TAPS; ++q) {
                const unsigned byte = (mbits[q] >> (kk * 16 + (lane >> 5) * 8)) & 0xFFu;
                const uint4 mk = mtab[byte];
#pragma unroll
                for (int b = 0; b < TC; ++b) {
                    union { bf16x8 v; uint4 u; } f;
                    f.v = w2_frag2(bB, addr_b3[q][b][0] + kk * 16 * RBB, addr_b3[q][b][1] + kk * 16 * RBB);
                    f.u.x &= mk.x; f.u.y &= mk.y; f.u.z &= mk.z; f.u.w &= mk.w;
                    bfr[q][b] = f.v;
                }
            }
        } else {
#pragma unroll
            for (int b = 0; b < TC; ++b) bfr[0][b] = w2_frag2(bB, addr_b[b][0] + kk * 16 * RBB, addr_b[b][1] + kk * 16 * RBB);
        }
    };
    auto mma = [&](const bf16x8 (&af)[TN], const bf16x8 (&bfr)[TAPS][TC]) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < TAPS; ++q)
#pragma unroll
            for (int a = 0; a < TN; ++a)
#pragma unroll
                for (int b = 0; b < TC; ++b)
                    acc[q][a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bfr[q][b], acc[q][a][b], 0, 0, 0);
    };
    auto bias_rows = [&](int stage) __attribute__((always_inline)) {
        // fused bias gradient: column sums of the dy tile that is in LDS (512 threads: BN columns x 512 / BN row groups)
        const unsigned char* bA = smem + stage * BUF_BYTES;
        constexpr int TPC = 512 / BN, RPT = CR / TPC;
        const int col = t % BN, r0 = (t / BN) * RPT;
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const int row = r0 + r;
            bsum += (float)*reinterpret_cast<const bf16_t*>(bA + row * RBA + ((((col >> 3) ^ w2_swz(row)) << 4) | ((col & 7) * 2)));
        }
    };

    // ---- main loop.  Stages hold chunks c .. c+NS-1 at the top of iteration c.  Per iteration:
    //   reads(c, step 1) | MFMA(c, step 0) | wait chunk c+1, BARRIER (chunk c+1 visible; nobody reads chunk c any more)
    //   | DMA chunk c+NS into chunk c's stage | reads(c+1, step 0) | MFMA(c, step 1)
    // so every fragment read is in flight under the 8 MFMAs (256 cycles) of the previous step, with one barrier per chunk.
    // (Rows past M read as zero: the dy descriptor ends at the last row; the gather path tests m < M per row.  Chunks past
    // the split's end re-fetch its last chunk and are never multiplied.)
#pragma unroll
    for (int s0 = 0; s0 < NS; ++s0) issue(s0);
    bf16x8 fa0[TN], fb0[TAPS][TC], fa1[TN], fb1[TAPS][TC];
    if (dma_wave) w2_wait_vmcnt<(NS - 1) * LPT>();
    __syncthreads();
    if (FUSED) row_masks();
    load_frags(0, 0, fa0, fb0);
    int cbuf = 0;
    for (int c = 0; c < nch; ++c) {
        const int nbuf = cbuf + 1 == NS ? 0 : cbuf + 1;
        if (!(abl & 2)) {
            load_frags(cbuf, 1, fa1, fb1);
            if (do_bias) bias_rows(cbuf);
            mma(fa0, fb0);
        }
        if (dma_wave && !(abl & 1)) w2_wait_vmcnt<(NS - 2) * LPT>();      // this thread's pieces of chunk c+1 have landed
        __syncthreads();
        if (!(abl & 1)) issue(cbuf);                           // chunk c + NS
        if (FUSED) { next_rows(); row_masks(); }               // masks of chunk c + 1
        if (!(abl & 2)) {
            load_frags(nbuf, 0, fa0, fb0);
            mma(fa1, fb1);
        }
        cbuf = nbuf;
    }
    w2_wait_vmcnt<0>();

    if (do_bias) {
        const int n = n0 + t % BN;
        if (n < p.N) atomicAdd(p.dbias + n, bsum);
    }
    // ---- epilogue.  C/D layout of the 32x32 MFMA: lane l holds column (l & 31) = input channel c and rows
    // (r & 3) + 8 * (r >> 2) + 4 * (l >> 5) = output channel n.  Each wave turns its 32-row blocks around through its own slice
    // of the (dead) stage memory, so that every global access is a 16-B piece of ONE output row per lane -- a wave instruction
    // covers whole rows of the wave tile -- and the read-modify-write of the unsplit case has all its loads in flight at once.
    const int taps = p.KH * p.KW;
    float* const dst = p.part ? p.part + (size_t)split * p.out_elems : p.dw;
    const bool direct = p.part == nullptr;
    constexpr int LPR = WTC / 4, RPI = 64 / LPR, NI = 32 / RPI;      // lanes per row, rows per instruction, instructions per block
    static_assert((size_t)8 * 32 * WTC * 4 <= (size_t)NS * BUF_BYTES, "epilogue slices do not fit the stages");
    float* ep = reinterpret_cast<float*>(smem) + wave * (32 * WTC);
    __syncthreads();                 // every wave is done reading the stages; the DMA tail is drained (vmcnt 0 above)
    const int er = lane / LPR, ec = (lane % LPR) * 4;
    float ss = 0.f;                  // this thread's share of |dw after|^2 - |dw before|^2 (direct tiles only: the gradient norm)
#pragma unroll
    for (int tp = 0; tp < TAPS; ++tp) {
    const int otap = FUSED ? tap * 3 + tp : tap;
#pragma unroll
    for (int a = 0; a < TN; ++a) {
#pragma unroll
        for (int b = 0; b < TC; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                ep[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * WTC + b * 32 + (lane & 31)] = acc[tp][a][b][r];
        constexpr int NB = NI > 8 ? 8 : NI;                          // 16-B pieces per lane in flight at a time
#pragma unroll
        for (int i0 = 0; i0 < NI; i0 += NB) {
            f32x4 v[NB], old[NB];
            float* o[NB]; bool ok[NB];
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int row = (i0 + i) * RPI + er;
                v[i] = *reinterpret_cast<const f32x4*>(ep + row * WTC + ec);
                const int n = n0 + wn * WTN + a * 32 + row, c = c0 + wc * WTC + ec;
                ok[i] = n < p.N && c < p.SC;
                o[i] = dst + ((size_t)(ok[i] ? n : 0) * taps + otap) * p.SC + (ok[i] ? c : 0);
                if (direct && p.scale) v[i] *= p.scale[ok[i] ? n : 0];
            }
            if (direct && p.accumulate) {
#pragma unroll
                for (int i = 0; i < NB; ++i) old[i] = *reinterpret_cast<const f32x4*>(o[i]);
                if (p.sqacc) {               // (a + b)^2 - a^2 = b (2a + b)
#pragma unroll
                    for (int i = 0; i < NB; ++i) if (ok[i]) { const f32x4 d = v[i] * (old[i] + old[i] + v[i]); ss += (d[0] + d[1]) + (d[2] + d[3]); }
                }
#pragma unroll
                for (int i = 0; i < NB; ++i) v[i] += old[i];
            } else if (direct && p.sqacc) {
#pragma unroll
                for (int i = 0; i < NB; ++i) if (ok[i]) { const f32x4 d = v[i] * v[i]; ss += (d[0] + d[1]) + (d[2] + d[3]); }
            }
#pragma unroll
            for (int i = 0; i < NB; ++i) if (ok[i]) *reinterpret_cast<f32x4*>(o[i]) = v[i];
            if (direct && p.g16) {           // the bf16 exchange twin of the same elements
                typedef bf16_t w2_bf16x4 __attribute__((ext_vector_type(4)));
#pragma unroll
                for (int i = 0; i < NB; ++i)
                    if (ok[i]) *reinterpret_cast<w2_bf16x4*>(p.g16 + (o[i] - p.dw)) = w2_bf16x4{(bf16_t)v[i][0], (bf16_t)v[i][1], (bf16_t)v[i][2], (bf16_t)v[i][3]};
            }
        }
    }
    }
    if (direct && p.sqacc) {
        ss = rt_wave_sum(ss);
        if (lane == 0) rt_sq_add(p.sqacc, (unsigned)blockIdx.x * 8u + (unsigned)wave, ss);
    }
}

// One launch for a whole group: every problem carries its tile configuration (0: 256x256, 1: 128x256, 2: 256x128, 3: 128x128),
// so that the small-output problems of a stage (layer2: 128-channel sides) do not serialise into four chip-wide launches
// with four tails and four reductions.
template <int CR, int NS>
__global__ __launch_bounds__(512, 1) void w2_grouped_kernel(const W2Group g) {
    // block id -> XCD x = id & 7 (hardware round-robin), position q = id >> 3 in that XCD's order -> (problem, split, tile):
    // problem i owns positions [cum[x][i], cum[x][i+1]); inside, its row splits s with (s + xrot) % 8 == x follow each other,
    // each with all of its tiles adjacent (they read the same dy / x rows: one trip over the fabric per XCD-resident split).
    const int x = (int)blockIdx.x & 7, q = (int)blockIdx.x >> 3;
    // remap (REFTR_W2_XCD=2): the linear task order (problem, row split, tile -- the tasks that read the same rows are adjacent) is
    // dealt to the XCDs in CONTIGUOUS runs of gridDim.x / 8 instead of round-robin: same task count per XCD as the linear map
    // (no balance cost), but a split's tiles / taps share one XCD's L2
    const int b = g.remap ? rt_xcd_remap((int)blockIdx.x, (int)gridDim.x, 1) : (int)blockIdx.x;
    int lo = 0, hi = g.n;
    if (g.xcd) {
        if (q >= g.cum[x][g.n]) return;
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (g.cum[x][mid] <= q) lo = mid; else hi = mid; }
    } else {
        if (b >= g.cum[0][g.n]) return;
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (g.cum[0][mid] <= b) lo = mid; else hi = mid; }
    }
    lo = __builtin_amdgcn_readfirstlane(lo);
    const W2Prob& p = g.p[lo];
    int split, tile;
    if (g.xcd) {
        const int local = q - g.cum[x][lo];
        tile = local % p.tiles;
        split = ((x - p.xrot) & 7) + 8 * (local / p.tiles);
    } else {
        const int local = b - g.cum[0][lo];
        split = local / p.tiles; tile = local - split * p.tiles;
    }
    split = __builtin_amdgcn_readfirstlane(split); tile = __builtin_amdgcn_readfirstlane(tile);
    const int sel = __builtin_amdgcn_readfirstlane(p.cfg * 2 + (p.simple ? 1 : 0));
    switch (sel) {
        case 0: w2_body<256, 256, 4, 2, CR, NS, false>(p, split, tile, g.abl, g.pf); break;
        case 1: w2_body<256, 256, 4, 2, CR, NS, true>(p, split, tile, g.abl, g.pf); break;
        case 2: w2_body<128, 256, 2, 4, CR, NS, false>(p, split, tile, g.abl, g.pf); break;
        case 3: w2_body<128, 256, 2, 4, CR, NS, true>(p, split, tile, g.abl, g.pf); break;
        case 4: w2_body<256, 128, 4, 2, CR, NS, false>(p, split, tile, g.abl, g.pf); break;
        case 5: w2_body<256, 128, 4, 2, CR, NS, true>(p, split, tile, g.abl, g.pf); break;
        case 6: w2_body<128, 128, 2, 4, CR, NS, false>(p, split, tile, g.abl, g.pf); break;
        case 7: w2_body<128, 128, 2, 4, CR, NS, true>(p, split, tile, g.abl, g.pf); break;
        default: w2_body<128, 128, 2, 4, CR, NS, false, 3>(p, split, tile, g.abl, 0); break;      // cfg 4: three taps per workgroup
    }
}

// dw[i] (+)= scale[i / row_elems] * sum_s part[s][i] for every split problem of a group (one launch)
struct W2Reduce { const float* part[W2_MAXP]; float* dw[W2_MAXP]; const float* scale[W2_MAXP]; float* sqacc[W2_MAXP]; bf16_t* g16[W2_MAXP];
                  int out_elems[W2_MAXP], row_elems[W2_MAXP], nsplit[W2_MAXP], accumulate[W2_MAXP], first[W2_MAXP + 1]; int n; };
__global__ __launch_bounds__(256) void w2_reduce_kernel(const W2Reduce g) {
    int lo = 0, hi = g.n - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (g.first[mid] <= (int)blockIdx.x) lo = mid; else hi = mid - 1; }
    lo = __builtin_amdgcn_readfirstlane(lo);
    const float* part = g.part[lo]; float* dw = g.dw[lo]; const float* scale = g.scale[lo];
    const int out_elems = g.out_elems[lo], nsplit = g.nsplit[lo];
    const int i = (((int)blockIdx.x - g.first[lo]) * 256 + (int)threadIdx.x) * 4;
    float* const sq = g.sqacc[lo];
    __shared__ float sm[16];
    float ss = 0.f;
    if (i < out_elems) {
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    int s = 0;
    for (; s + 4 <= nsplit; s += 4) {
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(part + (size_t)s * out_elems + i);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(part + (size_t)(s + 1) * out_elems + i);
        const f32x4 v2 = *reinterpret_cast<const f32x4*>(part + (size_t)(s + 2) * out_elems + i);
        const f32x4 v3 = *reinterpret_cast<const f32x4*>(part + (size_t)(s + 3) * out_elems + i);
        a += (v0 + v1) + (v2 + v3);
    }
    for (; s < nsplit; ++s) a += *reinterpret_cast<const f32x4*>(part + (size_t)s * out_elems + i);
    if (scale) a *= scale[i / g.row_elems[lo]];
    f32x4* o = reinterpret_cast<f32x4*>(dw + i);
    f32x4 fin;
    if (g.accumulate[lo]) {
        const f32x4 old = *o;
        const f32x4 d = a * (old + old + a);              // |old + a|^2 - |old|^2
        ss = (d[0] + d[1]) + (d[2] + d[3]);
        fin = old + a;
    } else {
        const f32x4 d = a * a;
        ss = (d[0] + d[1]) + (d[2] + d[3]);
        fin = a;
    }
    *o = fin;
    if (g.g16[lo]) {
        typedef bf16_t w2_bf16x4 __attribute__((ext_vector_type(4)));
        *reinterpret_cast<w2_bf16x4*>(g.g16[lo] + i) = w2_bf16x4{(bf16_t)fin[0], (bf16_t)fin[1], (bf16_t)fin[2], (bf16_t)fin[3]};
    }
    }
    if (sq) {                                             // uniform per workgroup (one problem per workgroup)
        ss = rt_block_sum(ss, sm);
        if (threadIdx.x == 0) rt_sq_add(sq, blockIdx.x, ss);
    }
}

template <int CR, int NS>
int w2_launch(const W2Group& g, int blocks, hipStream_t s) {
    constexpr size_t smem = (size_t)NS * CR * (256 + 256) * 2 + 4 * 256;      // the largest configuration's stages + the prefetch scratch words
    static_assert((size_t)NS * (CR * 256 + 48 * 256) + 4096 <= smem, "fused-tap stages + mask table");
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)w2_grouped_kernel<CR, NS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        attr_done = true;
    }
    hipLaunchKernelGGL((w2_grouped_kernel<CR, NS>), dim3((unsigned)blocks), dim3(512), smem, s, g);
    RT_CHECK_LAUNCH();
    return RT_OK;
}

}  // namespace

// Eligibility: channel counts the DMA path can address in 16-B pieces, enough rows for the pipeline to matter.
bool rt_w2_eligible(const rt_conv_wgrad_desc& d) {
    static const int minm_env = RT_TUNE("REFTR_W2_MINM", 256);
    const long long M = (long long)d.B * d.DH * d.DW;
    if (M < minm_env || (d.N & 7) || (d.SC & 7) || d.N < 64 || d.SC < 64) return false;
    if (d.variant != 0 || d.msplit > 0) return false;
    if (M * d.N >= 0x3fffffffLL || (long long)d.B * d.SH * d.SW * d.SC >= 0x3fffffffLL) return false;
    if ((long long)d.N * d.KH * d.KW * d.SC >= 0x7fffffffLL) return false;
    return true;
}

static int w2_cfg_of(const rt_conv_wgrad_desc& d) {       // 0: 256x256, 1: 128x256, 2: 256x128, 3: 128x128, 4: 128x128 x 3 taps
    static const int fuse_env = RT_TUNE("REFTR_W2_FUSE3", 1);
    static const int fuse_minw = RT_TUNE("REFTR_W2_FUSE3_MINW", 8);
    static const int fuse_maxc = RT_TUNE("REFTR_W2_FUSE3_MAXC", 128);
    // fused taps where the per-tap tile could not be wider than 128 x 128 anyway (layer2, the RES head): 2.3x on those; with 256
    // channels a side the per-tap 256 x 256 tile is as good (layer3: equal), and at W = 20 the row masks cost more than the taps
    // save (layer4: 150 -> 169 us) -- REFTR_W2_FUSE3_MAXC / _MINW widen the choice
    if (fuse_env && d.dil <= 1 && d.KH == 3 && d.KW == 3 && d.stride == 1 && d.pad == 1 && d.SH == d.DH && d.SW == d.DW && d.DW >= fuse_minw &&
        d.N <= fuse_maxc && d.SC <= fuse_maxc && (long long)d.B * d.SH * d.SW * d.SC * 2 < 0x3fffffffLL) return 4;
    const bool n_big = d.N > 128, c_big = d.SC > 128;
    return n_big ? (c_big ? 0 : 2) : (c_big ? 1 : 3);
}

// Launches every descriptor of `idx` (all eligible) as grouped v2 launches.  Split policy: see the file header.
int rt_w2_run(const rt_conv_wgrad_desc* descs, const int* idx, int n, float* workspace, long long workspace_bytes, hipStream_t s) {
    static const int target_env = RT_TUNE("REFTR_W2_TARGET", 0);
    static const int xcd_env = RT_TUNE("REFTR_W2_XCD", 2);      // 0 linear, 1 per-XCD table (round-robin units), 2 contiguous runs
    static const int minrows_env = RT_TUNE("REFTR_W2_MINROWS", 256);
#ifdef RT_LAB                                      // lab builds only (hipcc -DRT_LAB): ablation probes, wrong results
    static const int abl_env = RT_TUNE("REFTR_W2_ABL", 0);
#else
    constexpr int abl_env = 0;
#endif
    static const int pf_env = RT_TUNE("REFTR_W2_PF", 0);
    constexpr int CR = 32;
    static const int BNs[5] = {256, 128, 256, 128, 128}, BCs[5] = {256, 256, 128, 128, 128};
    static double wts[5] = {4, 2, 2, 1, 3};                   // cost of one chunk of a tile task, in 128x128-tile units (REFTR_W2_WTS="a,b,c,d,e")
    static bool wts_done = false;
    if (!wts_done) {
#ifdef RT_LAB
        if (const char* e = getenv("REFTR_W2_WTS")) sscanf(e, "%lf,%lf,%lf,%lf,%lf", &wts[0], &wts[1], &wts[2], &wts[3], &wts[4]);
#endif
        wts_done = true;
    }
    static const int slots[4] = {256, 256, 256, 512};         // resident workgroups on the chip (LDS: 130 / 98 / 98 / 66 KB each)
    for (int base = 0; base < n; base += W2_MAXP) {         // one launch (and one split policy) per W2_MAXP problems
        // ---- group-level split policy: cut the m axis only while the group has fewer tile tasks than resident slots; work is
        // counted in 128x128-tile chunk units so that problems of different tile configurations end up with equally long workgroups
        const int* list = idx + base; const int m = n - base < W2_MAXP ? n - base : W2_MAXP;
        long long tiles_total = 0; double work = 0;
        for (int k = 0; k < m; ++k) {
            const rt_conv_wgrad_desc& d = descs[list[k]];
            const int cfg = w2_cfg_of(d); const int BN = BNs[cfg], BC = BCs[cfg];
            const long long M = (long long)d.B * d.DH * d.DW;
            const long long tl = (long long)((d.N + BN - 1) / BN) * ((d.SC + BC - 1) / BC) * (cfg == 4 ? 3 : d.KH * d.KW);
            tiles_total += tl; work += (double)tl * (double)((M + CR - 1) / CR) * (double)wts[cfg];
        }
        const int cfg = 0;
        const int target = target_env > 0 ? target_env : slots[cfg];
        double per_wg = work / (double)target;                // chunks a workgroup should own so that ~`target` workgroups cover the group
        // The per-problem rounding of the split counts must not push the launch over the resident slots: a 257th workgroup of
        // a long task is a whole extra task time for everybody (measured: a layer3 stage went from 288 to 352 us).  Coarsen
        // per_wg until the rounded counts fit.
        static const int fit_env = RT_TUNE("REFTR_W2_FIT", 1);
        if (fit_env && tiles_total < target) {
            for (int it = 0; it < 24; ++it) {
                long long total = 0;
                for (int k = 0; k < m; ++k) {
                    const rt_conv_wgrad_desc& d = descs[list[k]];
                    const int cfg = w2_cfg_of(d); const int BN = BNs[cfg], BC = BCs[cfg];
                    const long long M = (long long)d.B * d.DH * d.DW;
                    const long long tl = (long long)((d.N + BN - 1) / BN) * ((d.SC + BC - 1) / BC) * (cfg == 4 ? 3 : d.KH * d.KW);
                    const int total_chunks = (int)((M + CR - 1) / CR);
                    int sp = (int)((double)total_chunks * (double)wts[cfg] / per_wg + 0.5);
                    const int maxs = (int)(M / minrows_env);
                    if (sp > maxs) sp = maxs;
                    if (sp < 1) sp = 1;
                    const int cps = (total_chunks + sp - 1) / sp;
                    total += tl * ((total_chunks + cps - 1) / cps);
                }
                if (total <= target) break;
                per_wg *= 1.03;
            }
        }
        W2Group g; W2Reduce r; g.n = 0; r.n = 0; g.xcd = xcd_env == 1; g.remap = xcd_env == 2; g.abl = abl_env; g.pf = pf_env;
        int rblocks = 0; long long ws_off = 0;
        double xload[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int xcount[8] = {0, 0, 0, 0, 0, 0, 0, 0}, lin_total = 0;
        for (int x = 0; x < 8; ++x) g.cum[x][0] = 0;
        auto flush = [&]() -> int {
            if (g.n > 0) {
                int blocks;
                if (xcd_env == 1) { int mx = 0; for (int x = 0; x < 8; ++x) if (xcount[x] > mx) mx = xcount[x]; blocks = mx * 8; }
                else blocks = lin_total;
                const int rc = w2_launch<CR, 4>(g, blocks, s);
                if (rc != RT_OK) return rc;
            }
            if (r.n > 0) {
                r.first[r.n] = rblocks;
                hipLaunchKernelGGL(w2_reduce_kernel, dim3((unsigned)rblocks), dim3(256), 0, s, r);
                RT_CHECK_LAUNCH();
            }
            g.n = 0; r.n = 0; rblocks = 0; ws_off = 0; lin_total = 0;
            for (int x = 0; x < 8; ++x) { xload[x] = 0; xcount[x] = 0; g.cum[x][0] = 0; }
            return RT_OK;
        };
        for (int k = 0; k < m; ++k) {
            const rt_conv_wgrad_desc& d = descs[list[k]];
            W2Prob p;
            p.cfg = w2_cfg_of(d);
            const int BN = BNs[p.cfg], BC = BCs[p.cfg];
            p.dy = (const bf16_t*)d.dy; p.x = (const bf16_t*)d.x; p.dw = d.dw; p.scale = d.scale; p.dbias = d.dbias; p.part = nullptr; p.sqacc = d.sqacc; p.g16 = (bf16_t*)d.g16;
            p.SH = d.SH; p.SW = d.SW; p.SC = d.SC; p.DH = d.DH; p.DW = d.DW; p.N = d.N; p.KH = d.KH; p.KW = d.KW; p.stride = d.stride; p.pad = d.pad; p.dil = d.dil > 1 ? d.dil : 1;
            const long long M = (long long)d.B * d.DH * d.DW;
            p.M = (int)M;
            p.n_tiles = (d.N + BN - 1) / BN; p.c_tiles = (d.SC + BC - 1) / BC;
            p.tiles = p.n_tiles * p.c_tiles * (p.cfg == 4 ? 3 : d.KH * d.KW);
            p.out_elems = d.N * d.KH * d.KW * d.SC;
            p.dy_bytes = (unsigned)(M * d.N * 2); p.x_bytes = (unsigned)((long long)d.B * d.SH * d.SW * d.SC * 2);
            p.simple = (d.KH == 1 && d.KW == 1 && d.stride == 1 && d.pad == 0) ? 1 : 0;
            p.accumulate = d.overwrite ? 0 : 1;
            const int total_chunks = (int)((M + CR - 1) / CR);
            int splits = 1;
            if (tiles_total < target) {
                splits = (int)((double)total_chunks * (double)wts[p.cfg] / per_wg + 0.5);
                const int maxs = (int)(M / minrows_env);
                if (splits > maxs) splits = maxs;
                if (splits < 1) splits = 1;
            }
            long long need = splits > 1 ? (long long)splits * p.out_elems * 4 : 0;
            if (need > 0 && (p.out_elems & 3)) { splits = 1; need = 0; }
            if (need > workspace_bytes) {                 // never larger than the caller's scratch
                splits = (int)(workspace_bytes / ((long long)p.out_elems * 4));
                if (splits < 2) splits = 1;
                need = splits > 1 ? (long long)splits * p.out_elems * 4 : 0;
            }
            if (g.n == W2_MAXP || (need > 0 && ws_off + need > workspace_bytes && g.n > 0)) {
                const int frc = flush();
                if (frc != RT_OK) return frc;
            }
            p.chunks_per_split = (total_chunks + splits - 1) / splits;
            p.splits = (total_chunks + p.chunks_per_split - 1) / p.chunks_per_split;
            if (p.splits > 1 && workspace) {
                p.part = workspace + ws_off / 4;
                r.part[r.n] = p.part; r.dw[r.n] = p.dw; r.scale[r.n] = p.scale; r.sqacc[r.n] = p.sqacc; r.g16[r.n] = p.g16; r.out_elems[r.n] = p.out_elems;
                r.row_elems[r.n] = d.KH * d.KW * d.SC; r.nsplit[r.n] = p.splits; r.accumulate[r.n] = p.accumulate; r.first[r.n] = rblocks; ++r.n;
                rblocks += (p.out_elems / 4 + 255) / 256;
                ws_off += ((long long)p.splits * p.out_elems * 4 + 255) / 256 * 256;
            } else if (p.splits > 1) {                    // no scratch: do not split
                p.splits = 1; p.chunks_per_split = total_chunks;
            }
            // XCD placement: split s -> XCD (s + xrot) % 8; xrot = the XCD with the least work so far in this launch
            int best = 0;
            for (int x = 1; x < 8; ++x) if (xload[x] < xload[best]) best = x;
            p.xrot = xcd_env == 1 ? best : 0;
            for (int x = 0; x < 8; ++x) {
                const int j0 = (x - p.xrot) & 7;                                  // first split of this problem on XCD x
                const int cnt = j0 < p.splits ? (p.splits - j0 + 7) / 8 : 0;      // splits j0, j0 + 8, ...
                xcount[x] += cnt * p.tiles; xload[x] += (double)cnt * p.tiles * p.chunks_per_split;
                g.cum[x][g.n + 1] = xcount[x];
            }
            static const int dbg_env = RT_TUNE("REFTR_W2_DEBUG", 0);
            if (dbg_env) fprintf(stderr, "[w2] M=%d N=%d C=%d k=%d s=%d cfg=%d tiles=%d splits=%d chunks/split=%d  (group tiles %lld, per_wg %.0f)\n",
                                 p.M, p.N, p.SC, p.KH, p.stride, p.cfg, p.tiles, p.splits, p.chunks_per_split, tiles_total, per_wg);
            lin_total += p.tiles * p.splits;
            if (xcd_env != 1) g.cum[0][g.n + 1] = lin_total;
            g.p[g.n] = p; ++g.n;
        }
        const int frc = flush();
        if (frc != RT_OK) return frc;
    }
    return RT_OK;
}
