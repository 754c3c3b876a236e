// The LDS-DMA implicit-GEMM body (template code shared by rt_gemm.hip and rt_gemm_pipe.hip, which instantiate different
// tile / schedule variants in parallel translation units).  Internal to the library.
#pragma once
#include "rt_gemm_common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

// ------------------------------------------------------------------------------------------------
// LDS-DMA variant: operand tiles go global -> LDS directly (buffer_load_dwordx4 ... lds), never through VGPRs, so
// the ds_write half of the LDS pipe (~80 B/clk/CU, as expensive as the MFMAs of a 128x128x64 tile) disappears and
// NS tiles can be in flight without holding staging registers.  The DMA writes lane l of a wave instruction to
// LDS base + 16*l (lane-linear), so the XOR swizzle is applied on the SOURCE side: lane l (row l>>3 of an 8-row
// group, slot l&7) fetches K chunk (l&7)^(l>>3) of its row.  Out-of-range offsets make the DMA write zeros.
// Completion is tracked with explicit counted vmcnt waits (the compiler cannot tell which stage a ds_read aliases).
template <int N> __device__ __forceinline__ void rt_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
typedef __attribute__((ext_vector_type(4))) int i32x4;
__device__ __forceinline__ i32x4 rt_make_rsrc(const void* ptr, unsigned bytes) {
    const uint64_t a = (uint64_t)ptr;
    return i32x4{(int)(uint32_t)a, (int)(uint32_t)(a >> 32), (int)bytes, 0x00020000};   // stride 0, raw buffer
}
// One 16-B-per-lane global -> LDS DMA: lane l lands at LDS byte address lds_base + 16*l (lds_base wave-uniform).
// Issued as inline asm so that the compiler's waitcnt pass does not turn every later ds_read into vmcnt(0).
__device__ __forceinline__ void rt_dma16(const i32x4 rsrc, unsigned lds_base, int voff, int soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 ::"s"(lds_base), "v"(voff), "s"(rsrc), "s"(soff)
                 : "memory", "m0");
}

// The body is a device function: it runs as a kernel of its own (conv_gemm_dma_kernel) or as one of up to 12 independent
// problems of a grouped launch (conv_gemm_dma_grouped_kernel); bx / by / gx stand for blockIdx.x / blockIdx.y / gridDim.x.
// NW = 4 waves arranged 2(n) x 2(m), or 8 waves 2(n) x 4(m): the same tile with smaller wave tiles and twice the waves per CU
// PIPE = 1 (rt_gemm_pipe.hip): the software-pipelined K loop -- see the `if constexpr (PIPE)` block.
// (Round 4's direct epilogue from permuted weight rows -- EPI = 1 -- was measured neutral and removed in round 5: profiles/r04v_*.)
template <int BM, int BN, int MODE, int NS, int NW = 4, int PIPE = 0>
__device__ __forceinline__ void gemm_dma_body(const bf16_t* __restrict__ src, const bf16_t* __restrict__ wgt, const GemmArgs& p,
                                              const int bx, const int by, const int gx) {
    // PIPE = 2 (rt_gemm_pp.hip): the 8 waves are TWO 4-wave groups that each own the whole tile for every other K tile
    // (K-parity ping-pong): staging and wave-tile geometry are those of a 4-wave workgroup, the epilogue is shared by all 8.
    constexpr bool PP = PIPE == 2;                                 // the ping-pong form
    constexpr int NWG = PP ? NW / 2 : NW;                  // waves of one compute group
    constexpr int NT = 64 * NW, GT = 64 * NWG, WM = NWG / 2, RPP = GT / 8;   // threads, group threads, waves along m, rows one DMA pass covers
    constexpr int TM = BM / (16 * WM), TN = BN / 32;
    constexpr int AJ = BN / RPP, BJ = BM / RPP;
    static_assert(NW == 4 || NW == 8, "4 or 8 waves"); static_assert(AJ >= 1 && BJ >= 1 && TM >= 1, "tile too small for the wave count");
    static_assert(!PP || (NW == 8 && (NS == 3 || NS == 4)), "ping-pong form: 8 waves, 3 or 4 stages");
    constexpr int A_BYTES = BN * 128, B_BYTES = BM * 128, BUF_BYTES = A_BYTES + B_BYTES;
    constexpr int LPT = AJ + BJ;                   // DMA instructions per thread per K tile
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef __attribute__((address_space(3))) void* lds_ptr;

    const int t = threadIdx.x;
    const int lane = t & 63, wave = (t >> 6) % NWG;               // wave inside its compute group
    const int grp = PP ? __builtin_amdgcn_readfirstlane(t >> 6) / NWG : 0;
    const int wn = wave & 1, wm = wave >> 1;
    const int li = lane & 15, lg = lane >> 4;

    const int n_tiles = (p.N + BN - 1) / BN;
    const int bid = rt_xcd_remap(bx, gx, p.xcd);
    // tile order inside an XCD's contiguous run: n fastest shares the activation rows of an m tile across its n tiles; m fastest
    // (dense products with few rows and many output features: the BERT Linears) shares a weight slab across the m tiles instead
    int tile_n = bid % n_tiles, tile_m = bid / n_tiles;
    if (MODE == 0 && p.mfast) { const int m_tiles = (p.M + BM - 1) / BM; tile_m = bid % m_tiles; tile_n = bid / m_tiles; }
    const int n0 = tile_n * BN, m0 = tile_m * BM;

    // MODE 3 (stride-2 backward-data): blockIdx.y is the output parity class (y&1, x&1).  Only taps with
    // kh = (y+pad) mod 2, kw = (x+pad) mod 2 reach a source pixel, so a class walks 1/4 of the taps on average
    // instead of multiplying zeros for the other 3/4; its rows are the class's pixels in (b, y/2, x/2) order.
    int cy = 0, cx = 0, ny = p.DH, nx = p.DW, kh0 = 0, kw0 = 0, Mloc = p.M;
    if (MODE == 3) {
        cy = by >> 1; cx = by & 1;
        ny = (p.DH - cy + 1) >> 1; nx = (p.DW - cx + 1) >> 1;
        Mloc = p.B * ny * nx;
        kh0 = (cy + p.pad) & 1; kw0 = (cx + p.pad) & 1;
        if (m0 >= Mloc) return;
    }

    const int srow = (t % GT) >> 3;
    const int chunk = (t & 7) ^ (srow & 7);        // source-side swizzle
    constexpr int OOB = 0x7fffffff;

    int a_off[AJ];
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
        int n = n0 + srow + RPP * j;
        a_off[j] = n < p.N ? (n * p.K + chunk * 8) * 2 : OOB;
    }
    int b_off[BJ], b_y[BJ], b_x[BJ];
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
        const int m = m0 + srow + RPP * j;
        const bool ok = m < Mloc;
        const int mm = ok ? m : 0;
        if (MODE == 0) {
            b_off[j] = ok ? (mm * p.SC + chunk * 8) * 2 : OOB; b_y[j] = 0; b_x[j] = 0;
        } else {
            int dx = mm % nx;
            const int tmp = mm / nx;
            int dy = tmp % ny;
            const int b = tmp / ny;
            if (MODE == 3) { dy = 2 * dy + cy; dx = 2 * dx + cx; }
            if (MODE == 1) { b_y[j] = dy * p.stride - p.pad; b_x[j] = dx * p.stride - p.pad; }
            else           { b_y[j] = dy + p.pad;            b_x[j] = dx + p.pad; }
            if (!ok) b_y[j] = -(1 << 28);           // every tap of a ragged row fails the bounds test
            b_off[j] = b * p.SH * p.SW * p.SC + chunk * 8;
        }
    }

    f32x4 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    int nk = p.K >> 6;
    if (MODE == 3) {
        const int nkh = kh0 < p.KH ? (p.KH - kh0 + 1) >> 1 : 0, nkw = kw0 < p.KW ? (p.KW - kw0 + 1) >> 1 : 0;
        nk = nkh * nkw * (p.SC >> 6);
    }
    int lk = 0, c0 = 0, kw = kw0, kh = kh0;
    const i32x4 rs_w = rt_make_rsrc(wgt, p.wgt_bytes), rs_x = rt_make_rsrc(src, p.src_bytes);
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr)smem + (unsigned)__builtin_amdgcn_readfirstlane(wave) * 1024u;

    auto advance_k = [&]() __attribute__((always_inline)) {      // the K-tile cursor (lk, kh, kw, c0) one tile on; saturates at the last tile
        if (lk + 1 < nk) {
            ++lk;
            if (MODE == 3) {
                c0 += 64;
                if (c0 >= p.SC) { c0 = 0; kw += 2; if (kw >= p.KW) { kw = kw0; kh += 2; } }
            } else if (MODE != 0) {
                c0 += 64;
                if (c0 >= p.SC) { c0 = 0; ++kw; if (kw >= p.KW) { kw = 0; ++kh; } }
            }
        }
    };
    // piece i of the current K tile: i < AJ -> 8-row group (this wave's) of weight slab i, else of activation slab i - AJ
    auto issue_piece = [&](int buf, int i) __attribute__((always_inline)) {
        const unsigned bA = lds0 + buf * BUF_BYTES;      // this wave's 8-row group of each 32-row slab
        const unsigned bB = bA + A_BYTES;
        const int k0b = MODE == 3 ? ((kh * p.KW + kw) * p.SC + c0) * 2 : lk << 7;
        if (i < AJ) { rt_dma16(rs_w, bA + i * (RPP * 128), a_off[i], k0b); return; }
        const int j = i - AJ;
        if (MODE == 0) {
            rt_dma16(rs_x, bB + j * (RPP * 128), b_off[j], k0b);
        } else {
            bool ok;
            int sy, sx;
            if (MODE == 1) { sy = b_y[j] + kh * p.dil; sx = b_x[j] + kw * p.dil; ok = true; }
            else if (MODE == 3) {
                const int ny_ = b_y[j] - kh, nx_ = b_x[j] - kw;      // even by construction of the class
                ok = (ny_ | nx_) >= 0;
                sy = ny_ >> 1; sx = nx_ >> 1;
            } else {
                const int ny = b_y[j] - kh * p.dil, nx = b_x[j] - kw * p.dil;
                const int msk = p.stride - 1;
                ok = ((ny | nx) >= 0) && (((ny | nx) & msk) == 0);
                sy = ny >> p.sshift; sx = nx >> p.sshift;
            }
            ok = ok && (unsigned)sy < (unsigned)p.SH && (unsigned)sx < (unsigned)p.SW;
            const int off = (b_off[j] + (sy * p.SW + sx) * p.SC) * 2;
            rt_dma16(rs_x, bB + j * (RPP * 128), ok ? off : OOB, c0 * 2);
        }
    };
    auto issue_only = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < LPT; ++i) issue_piece(buf, i);
    };
    auto issue_tile = [&](int buf) __attribute__((always_inline)) { issue_only(buf); advance_k(); };
    auto compute = [&](int buf) __attribute__((always_inline)) {
        const unsigned char* bA = smem + buf * BUF_BYTES;
        const unsigned char* bB = bA + A_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 af[TN], bfr[TM];
            const int slot = ((kk * 4 + lg) ^ (li & 7)) << 4;
#pragma unroll
            for (int a = 0; a < TN; ++a) {
                const int row = wn * (BN / 2) + a * 16 + li;
                af[a] = *reinterpret_cast<const bf16x8*>(bA + row * 128 + slot);
            }
#pragma unroll
            for (int b = 0; b < TM; ++b) {
                const int row = wm * (BM / WM) + b * 16 + li;
                bfr[b] = *reinterpret_cast<const bf16x8*>(bB + row * 128 + slot);
            }
#pragma unroll
            for (int a = 0; a < TN; ++a)
#pragma unroll
                for (int b = 0; b < TM; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a], bfr[b], acc[a][b], 0, 0, 0);
        }
    };

    // Epilogue operands first: the bf16 residual / ReLU-gate pieces this thread will need in the row-coalesced epilogue are
    // requested BEFORE the first operand tile, so that their HBM latency runs under the K loop instead of in front of the
    // stores (measured with REFTR_GEMM_ABL: the epilogue alone is as long as loads + MFMAs on the wide-output products, and
    // at <= 2 workgroups per CU in their epilogue at a time its reads are latency-, not bandwidth-bound).  They are the
    // oldest entries of this wave's vmcnt queue: the counted waits below stay conservative-exact.
    constexpr int EP_LD = BN + 4;                                  // padded fp32 row (bank spread)
    constexpr int HALVES = ((size_t)BM * EP_LD * 4 > (size_t)NS * BUF_BYTES) ? 2 : 1;    // 128x128 / 2 stages: two m-halves
    constexpr int ROWS = BM / HALVES;
    constexpr int CPR = BN / 8;                                    // 8-channel pieces per row
    constexpr int PIECES = (ROWS * CPR + NT - 1) / NT;              // the 32-row / 32-column tiles have fewer pieces than threads
    constexpr bool RAGGED_PIECES = (ROWS * CPR) % NT != 0;
    constexpr bool CAN_PRE = HALVES == 1 && PIECES <= 4;
    const bool epi_lds = p.epi_lds && (p.N & 7) == 0;
    const bool pre = CAN_PRE && epi_lds && p.prefetch && (p.res_bf16 || p.gate);
    bf16x8 pre_res[CAN_PRE ? PIECES : 1], pre_gate[CAN_PRE ? PIECES : 1];
    auto out_piece = [&](int idx, int& m, int& n) __attribute__((always_inline)) -> bool {
        const int rl = idx / CPR, cl = (idx - rl * CPR) * 8;
        m = m0 + rl; n = n0 + cl;
        if (m >= Mloc || n >= p.N) return false;
        if (MODE == 3) {
            const int xx = m % nx, tmp = m / nx, yy = tmp % ny, bb = tmp / ny;
            m = (bb * p.DH + 2 * yy + cy) * p.DW + 2 * xx + cx;
        }
        return true;
    };
    if (CAN_PRE && pre) {
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            int m, n;
            const bool ok = (!RAGGED_PIECES || i * NT + t < ROWS * CPR) && out_piece(i * NT + t, m, n);
            const size_t o = ok ? (size_t)m * p.N + n : 0;
            pre_res[i] = p.res_bf16 ? *reinterpret_cast<const bf16x8*>(p.res_bf16 + o) : bf16x8{};
            pre_gate[i] = p.gate ? *reinterpret_cast<const bf16x8*>(p.gate + o) : bf16x8{};
        }
    }

    if constexpr (PIPE == 2) {
        // K-parity ping-pong (round 6).  At one workgroup per CU (the 200-tile convolutions at B = 8) the loops below run the operand
        // DMA (0.33 us per K tile alone) and the MFMA side (0.49 us alone: barrier -> ds_read latency -> MFMAs, every wave in
        // lockstep) one after the other (0.65-0.72 us together, profiles/r05_gemm_abl_conv128.txt).  Here the two waves of a SIMD are
        // never in the same kind of segment: group g (waves 4g .. 4g+3, one per SIMD) owns the K tiles of parity g with the WHOLE
        // output tile as 2 x 2 wave tiles of (BM/2) x (BN/2), and the workgroup's timeline is a sequence of INTERVALS separated by
        // one raw s_barrier each,
        //     interval i:   group i&1      M(i):   per DMA piece of tile i+NS-1 two ds_read_b128 of tile i's fragments, counted waits
        //                   other group    X(i-1): the 2 * TN * TM MFMAs of tile i-1, nothing else
        // so every SIMD's matrix pipe has one wave issuing back-to-back MFMAs while its partner reads LDS and feeds the DMA queue:
        // one barrier per K tile (not two), 64 KB of fragment reads per 128 x 128 x 64 tile (64 x 64 wave tiles) instead of 96 KB,
        // no over-fetch past the last tile.  The groups' partial sums meet once, in the LDS-staged epilogue.
        //   NS = 4: tile k is issued in M(k-3) by the OTHER group, which waits for its pieces in M(k-1) (vmcnt leaves tile k+2 out).
        //   NS = 3: tile k is issued in M(k-2) by its OWN group, which waits for its pieces (vmcnt 0) behind the MFMAs of X(k-2).
        //   RAW  either way the wait precedes the barrier that ends interval k-1, and M(k) reads the tile after that barrier.
        //   WAR  stage (i+NS-1) % NS held tile i-1: read in M(i-1), RETIRED (lgkmcnt(0)) before the barrier ending interval i-1.
        // Measured (profiles/r06*_pp_*, LAB_NOTES.md round 6): MFMA segment 544-690 cycles, memory segment ~1050 (16 reads at ~23 + 8
        // pieces at ~85, additive in one in-order wave whether interleaved or not) -- 1.5-5 % faster than the pipelined form on the
        // layer3 shapes back to back, SLOWER inside the step at 4 stages (128 KB: no other stream's workgroup fits beside it).
        constexpr int DEPTH = NS - 1;
#ifdef RT_LAB       // probes (REFTR_GEMM_ABL): 1 no operand DMA, 2 no MFMAs, 8 no fragment reads (wrong results); 32 no priority raise (correct)
#define PP_ABL(bit) (p.abl & (bit))
#else
#define PP_ABL(bit) false
#endif
        auto bar = [&]() __attribute__((always_inline)) {
            __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0);
        };
        auto issue_adv = [&](int stage) __attribute__((always_inline)) { issue_only(stage); advance_k(); advance_k(); };
        bf16x8 fa[2][TN], fb[2][TM];
        if (PP_ABL(8)) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
                for (int a = 0; a < TN; ++a) fa[kk][a] = bf16x8{};
#pragma unroll
                for (int b = 0; b < TM; ++b) fb[kk][b] = bf16x8{};
            }
        }
        // the prologue stands in for the memory segments before interval 0
        if (!PP_ABL(1)) {
            if constexpr (NS == 4) {                                           // group g issues the tiles of parity g ^ 1
                if (grp == 0) {
                    advance_k();
                    if (1 < nk) issue_adv(1);
                } else {
                    issue_adv(0);
                    if (2 < nk) { issue_adv(2); rt_wait_vmcnt<LPT>(); } else rt_wait_vmcnt<0>();        // tile 0 landed
                }
            } else {                                                           // group g issues its own tiles
                if (grp) advance_k();
                if (grp < nk) issue_adv(grp);
                rt_wait_vmcnt<0>();
            }
        }
        bar();                                                                 // ... everyone's pieces
        const int iters = (nk >> 1) + (grp == 0 ? 1 : 0);
        if (grp) bar();                                                        // group 1 idles through interval 0
        int rs = grp, ws = (grp + DEPTH) % NS;                                 // stage of tile j / of tile j + DEPTH
        for (int it = 0, j = grp; it < iters; ++it, j += 2) {
            if (j < nk) {                                                      // M(j)
                const unsigned char* rA = smem + rs * BUF_BYTES;
                const unsigned char* rB = rA + A_BYTES;
                auto m_body = [&](auto with_dma) __attribute__((always_inline)) {
#pragma unroll
                    for (int i = 0; i < LPT; ++i) {
                        if constexpr (decltype(with_dma)::value) issue_piece(ws, i);
                        if (!PP_ABL(8)) {
#pragma unroll
                            for (int r = 2 * i; r < 2 * i + 2; ++r) {          // read r of the tile: kk-major, weights then activations
                                const int kk = r / LPT, q = r % LPT;
                                const int slot = ((kk * 4 + lg) ^ (li & 7)) << 4;
                                if (q < TN) fa[kk][q] = *reinterpret_cast<const bf16x8*>(rA + (wn * (BN / 2) + q * 16 + li) * 128 + slot);
                                else fb[kk][q - TN] = *reinterpret_cast<const bf16x8*>(rB + (wm * (BM / WM) + (q - TN) * 16 + li) * 128 + slot);
                            }
                        }
                    }
                };
                if (!PP_ABL(1) && j + DEPTH < nk) { m_body(std::true_type{}); advance_k(); advance_k(); }
                else m_body(std::false_type{});
                if (NS == 4 && !PP_ABL(1)) { if (j + DEPTH < nk) rt_wait_vmcnt<LPT>(); else rt_wait_vmcnt<0>(); }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            bar();
            if (j < nk) {                                                      // X(j)
                if (!PP_ABL(2)) {
                    if (!PP_ABL(32)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                        for (int a = 0; a < TN; ++a)
#pragma unroll
                            for (int b = 0; b < TM; ++b)
                                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[kk][a], fb[kk][b], acc[a][b], 0, 0, 0);
                    if (!PP_ABL(32)) __builtin_amdgcn_s_setprio(0);
                } else {
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
                        for (int a = 0; a < TN; ++a) asm volatile("" ::"v"(fa[kk][a]));
#pragma unroll
                        for (int b = 0; b < TM; ++b) asm volatile("" ::"v"(fb[kk][b]));
                    }
                }
                if (NS == 3) rt_wait_vmcnt<0>();                               // this group's tile j+2 has landed (its pieces)
            }
            bar();
            rs = (rs + 2) % NS; ws = (ws + 2) % NS;
        }
        if (grp) bar();
    } else if constexpr (PIPE == 1) {
        // Software-pipelined K loop (round 3).  The loop above is a DEPENDENT chain per K tile -- barrier -> fragment reads
        // (ds_read latency) -> MFMAs -> barrier, ~800 cycles for 128-256 cycles of MFMA work per wave -- that only other
        // resident workgroups hide (profiles/r02_tile_sweep_8wave.txt).  Here the MFMA fragments of tile kt+1 are read from LDS
        // into a SECOND register set while the MFMAs of tile kt run on the first, there is ONE barrier per K tile, and all NS
        // stages hold tiles in flight:
        //   iteration kt:  wait(tile kt+1 landed) | barrier | DMA tile kt+NS -> stage of tile kt | ds_read tile kt+1 -> regs B |
        //                  MFMAs on regs A (tile kt) | swap A, B
        // The barrier orders three things at once: tile kt+1's bytes are visible to every wave; every wave's reads of tile kt's
        // stage have returned (they are consumed by MFMAs issued before it) so the stage may be overwritten; and the reads of
        // tile kt+1 are issued one full MFMA phase before their first use.
        auto read_frags = [&](int buf, bf16x8 (&fa)[2][TN], bf16x8 (&fb)[2][TM]) __attribute__((always_inline)) {
            const unsigned char* bA = smem + buf * BUF_BYTES;
            const unsigned char* bB = bA + A_BYTES;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int slot = ((kk * 4 + lg) ^ (li & 7)) << 4;
#pragma unroll
                for (int a = 0; a < TN; ++a) fa[kk][a] = *reinterpret_cast<const bf16x8*>(bA + (wn * (BN / 2) + a * 16 + li) * 128 + slot);
#pragma unroll
                for (int b = 0; b < TM; ++b) fb[kk][b] = *reinterpret_cast<const bf16x8*>(bB + (wm * (BM / WM) + b * 16 + li) * 128 + slot);
            }
        };
        auto mma = [&](const bf16x8 (&fa)[2][TN], const bf16x8 (&fb)[2][TM]) __attribute__((always_inline)) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int a = 0; a < TN; ++a)
#pragma unroll
                    for (int b = 0; b < TM; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[kk][a], fb[kk][b], acc[a][b], 0, 0, 0);
        };
#pragma unroll
        for (int s = 0; s < NS; ++s) issue_tile(s);          // tiles 0 .. NS-1 (past the end: the last tile again, never consumed)
        bf16x8 fa0[2][TN], fb0[2][TM], fa1[2][TN], fb1[2][TM];
        rt_wait_vmcnt<(NS - 1) * LPT>();                     // tile 0 has landed (this thread's part) ...
        __syncthreads();                                     // ... everyone's
        read_frags(0, fa0, fb0);
        int rbuf = 1 % NS, wbuf = 0;
        auto step = [&](const bf16x8 (&ca)[2][TN], const bf16x8 (&cb)[2][TM], bf16x8 (&na)[2][TN], bf16x8 (&nb)[2][TM]) __attribute__((always_inline)) {
            rt_wait_vmcnt<(NS - 2) * LPT>();                 // tile kt+1 has landed (this thread's part)
            __syncthreads();
            issue_tile(wbuf);                                // tile kt+NS over tile kt's stage
            read_frags(rbuf, na, nb);                        // tile kt+1 -> the other register set, in flight under ...
            mma(ca, cb);                                     // ... the MFMAs of tile kt
            rbuf = rbuf + 1 == NS ? 0 : rbuf + 1;
            wbuf = wbuf + 1 == NS ? 0 : wbuf + 1;
        };
        for (int kt = 0; kt < nk; kt += 2) {
            step(fa0, fb0, fa1, fb1);
            if (kt + 1 >= nk) break;
            step(fa1, fb1, fa0, fb0);
        }
    } else {
    // prologue: tiles 0 .. NS-2 in flight (past the end the last tile is re-fetched into a free stage)
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) if (!(p.abl & 1)) issue_tile(s);
    int cbuf = 0, lbuf = NS - 1;
    if (p.early) {
        // issue-before-wait schedule: tile kt+NS-1 is requested BEFORE the wait for tile kt, so NS-1 tiles (not NS-2) are
        // in flight while the workgroup is parked; the price is a second barrier per K tile (stage release).
        for (int kt = 0; kt < nk; ++kt) {
            if (!(p.abl & 1)) issue_tile(lbuf);        // its stage was released by the barrier that ended iteration kt-1
            rt_wait_vmcnt<(NS - 1) * LPT>();           // tile kt has landed (this thread's part)
            __syncthreads();
            if (!(p.abl & 2)) compute(cbuf);
            __syncthreads();                           // everyone is done reading stage cbuf
            cbuf = cbuf + 1 == NS ? 0 : cbuf + 1;
            lbuf = lbuf + 1 == NS ? 0 : lbuf + 1;
        }
    } else {
        for (int kt = 0; kt < nk; ++kt) {
            rt_wait_vmcnt<(NS - 2) * LPT>();           // tile kt has landed (this thread's part)
            __syncthreads();                           // ... everyone's part; and stage lbuf (tile kt-1) is no longer read
            issue_tile(lbuf);                          // tile kt+NS-1
            compute(cbuf);
            cbuf = cbuf + 1 == NS ? 0 : cbuf + 1;
            lbuf = lbuf + 1 == NS ? 0 : lbuf + 1;
        }
    }
    }
    rt_wait_vmcnt<0>();                            // drain the over-fetched tail before the workgroup's LDS is released
    if (p.abl & 4) {                               // ablation probe (REFTR_GEMM_ABL, wrong results): no epilogue
#pragma unroll
        for (int a = 0; a < TN; ++a)
#pragma unroll
            for (int b = 0; b < TM; ++b) asm volatile("" ::"v"(acc[a][b]));
        return;
    }

    // Row-coalesced epilogue: the MFMA C/D layout gives a lane 4 channels of one pixel, i.e. 8-B (bf16) pieces on 16
    // different rows per store instruction.  Staging the fp32 tile through LDS (the operand stages are dead now) turns every
    // residual / gate read and every store into 16-B pieces of ONE row per lane, 128+ contiguous bytes per row.
    static_assert((size_t)ROWS * EP_LD * 4 <= (size_t)NS * BUF_BYTES, "epilogue tile does not fit the LDS stages");
    if (epi_lds) {
        float* tile = reinterpret_cast<float*>(smem);
        if constexpr (PP) {
            // the two groups' partial sums (odd / even K tiles) meet here: group 1 parks its accumulators in the fp32 tile, group 0
            // adds them to its own at the same lane positions (same wave tiling in both groups) and carries on as the tile's one owner
            static_assert(HALVES == 1, "ping-pong form: the fp32 tile must fit the stages");
            __syncthreads();
            if (grp == 1) {
#pragma unroll
                for (int a = 0; a < TN; ++a)
#pragma unroll
                    for (int b = 0; b < TM; ++b)
                        *reinterpret_cast<f32x4*>(tile + (wm * (BM / WM) + b * 16 + li) * EP_LD + wn * (BN / 2) + a * 16 + lg * 4) = acc[a][b];
            }
            __syncthreads();
            if (grp == 0) {
#pragma unroll
                for (int a = 0; a < TN; ++a)
#pragma unroll
                    for (int b = 0; b < TM; ++b)
                        acc[a][b] += *reinterpret_cast<const f32x4*>(tile + (wm * (BM / WM) + b * 16 + li) * EP_LD + wn * (BN / 2) + a * 16 + lg * 4);
            }
        }
#pragma unroll
        for (int h = 0; h < HALVES; ++h) {
            __syncthreads();    // every wave has drained its own DMA tail (vmcnt 0 above) and finished reading the stages / tile
            if (PP ? grp == 0 : (HALVES == 1 || wm / (WM / 2) == h)) {
#pragma unroll
                for (int a = 0; a < TN; ++a)
#pragma unroll
                    for (int b = 0; b < TM; ++b)
                        *reinterpret_cast<f32x4*>(tile + ((HALVES == 1 ? wm : wm % (WM / 2)) * (BM / WM) + b * 16 + li) * EP_LD + wn * (BN / 2) + a * 16 + lg * 4) = acc[a][b];
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < PIECES; ++i) {
                const int idx = i * NT + t;
                const int rl = idx / CPR, cl = (idx - rl * CPR) * 8;
                int m, n;
                if (RAGGED_PIECES && idx >= ROWS * CPR) continue;
                if (!out_piece(h * ROWS * CPR + idx, m, n)) continue;
                const f32x4 lo4 = *reinterpret_cast<const f32x4*>(tile + rl * EP_LD + cl), hi4 = *reinterpret_cast<const f32x4*>(tile + rl * EP_LD + cl + 4);
                const f32x8 v8 = f32x8{lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
                if (CAN_PRE && pre) epilogue8<true>(p, m, n, v8, pre_res[CAN_PRE ? i : 0], pre_gate[CAN_PRE ? i : 0]);
                else epilogue8(p, m, n, v8);
            }
        }
        return;
    }

#pragma unroll
    for (int a = 0; a < TN; ++a) {
        const int n = n0 + wn * (BN / 2) + a * 16 + lg * 4;
        if (n >= p.N) continue;
#pragma unroll
        for (int b = 0; b < TM; ++b) {
            int m = m0 + wm * (BM / WM) + b * 16 + li;
            if (m >= Mloc) continue;
            if (MODE == 3) {            // class-local row -> pixel row of the NHWC output
                const int xx = m % nx, tmp = m / nx, yy = tmp % ny, bb = tmp / ny;
                m = (bb * p.DH + 2 * yy + cy) * p.DW + 2 * xx + cx;
            }
            epilogue4(p, m, n, acc[a][b]);
        }
    }
}

template <int BM, int BN, int MODE, int NS, int MINB, int NW = 4, int PIPE = 0>
__global__ __launch_bounds__(64 * NW, MINB) void conv_gemm_dma_kernel(const bf16_t* __restrict__ src,
                                                                      const bf16_t* __restrict__ wgt, const GemmArgs p) {
    gemm_dma_body<BM, BN, MODE, NS, NW, PIPE>(src, wgt, p, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.x);
}

template <int BM, int BN, int NS, int MINB, int NW = 4, int PIPE = 0>
int launch_gemm_dma(const GemmArgs& a, hipStream_t s) {
    const int mt = (a.M + BM - 1) / BM, nt = (a.N + BN - 1) / BN;
    const size_t smem = (size_t)NS * (BM + BN) * 128;
    const dim3 grid((unsigned)(mt * nt)), block(64 * NW);
    const bool dense = (a.KH == 1 && a.KW == 1 && a.stride == 1 && a.pad == 0 && a.SH == a.DH && a.SW == a.DW);
    static const int par_env = RT_TUNE("REFTR_S2PARITY", 1);
    auto set_smem = [&](const void* f) {
        if (smem > 65536) (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    };
    if (dense) {
        set_smem((const void*)conv_gemm_dma_kernel<BM, BN, 0, NS, MINB, NW, PIPE>);
        // distinct operand slabs an XCD's run of R tiles touches: R / n_tiles + min(R, n_tiles) (n fastest) vs the same with m_tiles
        static const int mfast_env = RT_TUNE("REFTR_MFAST", 1);
        GemmArgs am = a;
        const double R = (double)(mt * nt) / 8.0;
        const double cn = R / nt + (R < nt ? R : nt), cm = R / mt + (R < mt ? R : mt);
        am.mfast = (mfast_env && a.xcd && mt * nt >= 16 && cm < cn) ? 1 : 0;
        hipLaunchKernelGGL((conv_gemm_dma_kernel<BM, BN, 0, NS, MINB, NW, PIPE>), grid, block, smem, s, a.src, a.wgt, am);
    } else if (!a.transposed) {
        set_smem((const void*)conv_gemm_dma_kernel<BM, BN, 1, NS, MINB, NW, PIPE>);
        hipLaunchKernelGGL((conv_gemm_dma_kernel<BM, BN, 1, NS, MINB, NW, PIPE>), grid, block, smem, s, a.src, a.wgt, a);
    } else if (a.stride == 2 && par_env) {
        const int m_cls = a.B * ((a.DH + 1) / 2) * ((a.DW + 1) / 2);           // largest parity class
        // grid.x padded to a multiple of 8: block (x, y) has linear id y * gridDim.x + x, so only then do the four parity classes
        // of a tile range (they gather from the same dy rows) sit on the same XCD as the tile map assumes (surplus blocks exit)
        const dim3 grid3((unsigned)(((((m_cls + BM - 1) / BM) * nt) + 7) / 8 * 8), 4);
        set_smem((const void*)conv_gemm_dma_kernel<BM, BN, 3, NS, MINB, NW, PIPE>);
        hipLaunchKernelGGL((conv_gemm_dma_kernel<BM, BN, 3, NS, MINB, NW, PIPE>), grid3, block, smem, s, a.src, a.wgt, a);
    } else {
        set_smem((const void*)conv_gemm_dma_kernel<BM, BN, 2, NS, MINB, NW, PIPE>);
        hipLaunchKernelGGL((conv_gemm_dma_kernel<BM, BN, 2, NS, MINB, NW, PIPE>), grid, block, smem, s, a.src, a.wgt, a);
    }
    RT_CHECK_LAUNCH();
    return RT_OK;
}

// dense rows only (Linears): the small-tile variants for the M = B * L products of the language branch
template <int BM, int BN, int NS, int MINB, int NW = 4, int PIPE = 0>
int launch_gemm_dma_dense(const GemmArgs& a, hipStream_t s) {
    const int mt = (a.M + BM - 1) / BM, nt = (a.N + BN - 1) / BN;
    const size_t smem = (size_t)NS * (BM + BN) * 128;
    const bool dense = (a.KH == 1 && a.KW == 1 && a.stride == 1 && a.pad == 0 && a.SH == a.DH && a.SW == a.DW);
    if (!dense) return RT_ERR_UNSUPPORTED;
    if (smem > 65536) (void)hipFuncSetAttribute((const void*)conv_gemm_dma_kernel<BM, BN, 0, NS, MINB, NW, PIPE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    GemmArgs am = a;
    const double R = (double)(mt * nt) / 8.0;
    const double cn = R / nt + (R < nt ? R : nt), cm = R / mt + (R < mt ? R : mt);
    am.mfast = (a.xcd && mt * nt >= 16 && cm < cn) ? 1 : 0;
    hipLaunchKernelGGL((conv_gemm_dma_kernel<BM, BN, 0, NS, MINB, NW, PIPE>), dim3((unsigned)(mt * nt)), dim3(64 * NW), smem, s, a.src, a.wgt, am);
    RT_CHECK_LAUNCH();
    return RT_OK;
}

}  // namespace
