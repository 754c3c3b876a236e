// rt_conv_gemm: bf16 MFMA implicit-GEMM for conv forward / backward-data and every Linear on the
// RefTR path (see include/reftr_hip.h for the reference call sites this replaces).
//
// Shape of the computation (per workgroup, 256 threads = 4 waves arranged 2(n) x 2(m)):
//   D[n, m] = sum_k Wt[n, k] * Xg[m, k]      (weights are the MFMA "A" operand, gathered pixels "B")
// so that each lane ends up with 4 CONSECUTIVE output channels of one pixel (C/D layout of
// v_mfma_f32_16x16x32_bf16: col = lane&15 -> pixel, row = 4*(lane>>4)+r -> channel) and the NHWC
// epilogue store is an 8-byte (bf16x4) / 16-byte (f32x4) contiguous write.
//
// K pipeline: 64-wide K tiles, global -> VGPR (predicated 16-B loads: conv padding / ragged M,N read as
// zero) -> XOR-swizzled LDS (slot ^= row&7, conflict-free for ds_read_b128 fragment reads), LDS double
// buffered with ONE barrier per K tile; next tile's global loads are in flight under the MFMAs.
#include "rt_common.h"
#include <stdlib.h>

namespace {

struct GemmArgs {
    const bf16_t* src; const bf16_t* wgt;
    bf16_t* out_bf16; float* out_f32; bf16_t* out_preact; float* acc2_f32;
    const float* bias; const float* res_f32; const bf16_t* res_bf16; const bf16_t* gate; const bf16_t* preact; const bf16_t* dtanh;
    int B, SH, SW, SC, DH, DW, N, KH, KW, stride, pad, transposed, act, res_first;
    float gate_scale, drop_p; uint32_t drop_seed; int drop_shift;
    int M, K, sshift, xcd, early, epi_lds, abl, prefetch, mfast;
    unsigned src_bytes, wgt_bytes;
    const uint32_t* seed_dev;
};

// epilogue of one lane's 4 consecutive output features of row m:
// +bias -> [+res] -> act -> dropout -> [+res] -> *gate -> *gelu'(preact) -> *(1 - dtanh^2) -> store
__device__ __forceinline__ void epilogue4(const GemmArgs& p, int m, int n, f32x4 v) {
    if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n);
    const size_t o = (size_t)m * p.N + n;
    if (p.out_preact) {
        bf16x4 pv;
#pragma unroll
        for (int r = 0; r < 4; ++r) pv[r] = (bf16_t)v[r];
        *reinterpret_cast<bf16x4*>(p.out_preact + o) = pv;
    }
    if (p.res_first) {
        if (p.res_f32) v += *reinterpret_cast<const f32x4*>(p.res_f32 + o);
        if (p.res_bf16) {
            const bf16x4 rr = *reinterpret_cast<const bf16x4*>(p.res_bf16 + o);
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
        if (p.res_f32) v += *reinterpret_cast<const f32x4*>(p.res_f32 + o);
        if (p.res_bf16) {
            const bf16x4 rr = *reinterpret_cast<const bf16x4*>(p.res_bf16 + o);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += (float)rr[r];
        }
    }
    if (p.gate) {
        const bf16x4 gg = *reinterpret_cast<const bf16x4*>(p.gate + o);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = ((float)gg[r] > 0.f) ? v[r] * p.gate_scale : 0.f;
    }
    if (p.preact) {
        const bf16x4 uu = *reinterpret_cast<const bf16x4*>(p.preact + o);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] *= rt_gelu_grad((float)uu[r]);
    }
    if (p.dtanh) {
        const bf16x4 tt = *reinterpret_cast<const bf16x4*>(p.dtanh + o);
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
__device__ __forceinline__ void epilogue8(const GemmArgs& p, int m, int n, f32x8 v, const bf16x8 pres = bf16x8{}, const bf16x8 pgate = bf16x8{}) {
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

// MODE 0: dense rows (1x1, stride 1, pad 0: every Linear and most bottleneck convs)
// MODE 1: forward conv gather       MODE 2: transposed (backward-data) gather
template <int BM, int BN, int MODE>
__global__ __launch_bounds__(256, 2) void conv_gemm_kernel(const bf16_t* __restrict__ src,
                                                        const bf16_t* __restrict__ wgt, const GemmArgs p) {
    constexpr int TM = BM / 32, TN = BN / 32;      // 16x16 MFMA tiles per wave along m / n
    constexpr int AJ = BN / 32, BJ = BM / 32;      // 16-B staging chunks per thread
    constexpr int A_BYTES = BN * 128, B_BYTES = BM * 128, BUF_BYTES = A_BYTES + B_BYTES;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wn = wave & 1, wm = wave >> 1;
    const int li = lane & 15, lg = lane >> 4;

    const int n_tiles = (p.N + BN - 1) / BN;
    const int tile_n = blockIdx.x % n_tiles, tile_m = blockIdx.x / n_tiles;
    const int n0 = tile_n * BN, m0 = tile_m * BM;

    const int chunk = t & 7, srow = t >> 3;

    // Per staged row: element offset of its (tap 0, channel chunk) source address, validity, and for the
    // gather modes the pixel coordinates the per-tap bounds test needs.
    int a_off[AJ]; bool a_ok[AJ];
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
        const int n = n0 + srow + 32 * j;
        a_ok[j] = n < p.N;
        a_off[j] = a_ok[j] ? n * p.K + chunk * 8 : 0;
    }
    int b_off[BJ], b_y[BJ], b_x[BJ]; bool b_ok[BJ];
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
        const int m = m0 + srow + 32 * j;
        b_ok[j] = m < p.M;
        const int mm = b_ok[j] ? m : 0;
        if (MODE == 0) {
            b_off[j] = mm * p.SC + chunk * 8; b_y[j] = 0; b_x[j] = 0;
        } else {
            const int dx = mm % p.DW;
            const int tmp = mm / p.DW;
            const int dy = tmp % p.DH;
            const int b = tmp / p.DH;
            if (MODE == 1) { b_y[j] = dy * p.stride - p.pad; b_x[j] = dx * p.stride - p.pad; }
            else           { b_y[j] = dy + p.pad;            b_x[j] = dx + p.pad; }
            b_off[j] = b * p.SH * p.SW * p.SC + chunk * 8;
        }
    }

    f32x4 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = p.K >> 6;
    // running K-tile position of the NEXT tile to be loaded: k0 = kt*64 = ((kh*KW)+kw)*SC + c0
    int lk = 0, c0 = 0, kw = 0, kh = 0;

    // Two register stages (R0 / R1) + two LDS buffers: the global loads of tile kt+2 are issued while tile kt is
    // being multiplied and tile kt+1 is still in flight, so a K tile's HBM/L2 latency is spread over two iterations.
    uint4 ra0[AJ], rb0[BJ], ra1[AJ], rb1[BJ];

    // Operand tiles are fetched with buffer loads: an out-of-range byte offset (padding taps, ragged M / N rows)
    // returns zeros from the hardware bounds check, so the loads are straight-line code — no exec-mask branches, and
    // the compiler keeps counted vmcnt waits (tile kt+2 stays in flight while tile kt+1 is written to LDS).
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(wgt), 0, p.wgt_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(src), 0, p.src_bytes, 0x00020000);
    constexpr int OOB = 0x7fffffff;
    auto load_tiles = [&](uint4 (&ra)[AJ], uint4 (&rb)[BJ]) __attribute__((always_inline)) {
        const int k0b = lk << 7;                                  // byte offset of the K tile inside a row
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs_w, a_ok[j] ? a_off[j] * 2 : OOB, k0b, 0);
            ra[j] = make_uint4(v[0], v[1], v[2], v[3]);
        }
#pragma unroll
        for (int j = 0; j < BJ; ++j) {
            if (MODE == 0) {
                const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs_x, b_ok[j] ? b_off[j] * 2 : OOB, k0b, 0);
                rb[j] = make_uint4(v[0], v[1], v[2], v[3]);
            } else {
                bool ok = b_ok[j];
                int sy, sx;
                if (MODE == 1) { sy = b_y[j] + kh; sx = b_x[j] + kw; }
                else {
                    const int ny = b_y[j] - kh, nx = b_x[j] - kw;
                    const int msk = p.stride - 1;
                    ok = ok && ((ny | nx) >= 0) && (((ny | nx) & msk) == 0);
                    sy = ny >> p.sshift; sx = nx >> p.sshift;
                }
                ok = ok && (unsigned)sy < (unsigned)p.SH && (unsigned)sx < (unsigned)p.SW;
                const int off = (b_off[j] + (sy * p.SW + sx) * p.SC) * 2;
                const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs_x, ok ? off : OOB, c0 * 2, 0);
                rb[j] = make_uint4(v[0], v[1], v[2], v[3]);
            }
        }
        // advance to the next K tile; past the end the last tile is simply re-loaded (never consumed), which keeps
        // every load unconditional and the compiler's vmcnt bookkeeping exact
        if (lk + 1 < nk) {
            ++lk;
            if (MODE != 0) {
                c0 += 64;
                if (c0 >= p.SC) { c0 = 0; ++kw; if (kw >= p.KW) { kw = 0; ++kh; } }
            }
        }
    };
    auto store_tiles = [&](int buf, const uint4 (&ra)[AJ], const uint4 (&rb)[BJ]) __attribute__((always_inline)) {
        unsigned char* bA = smem + buf * BUF_BYTES;
        unsigned char* bB = bA + A_BYTES;
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const int row = srow + 32 * j;
            *reinterpret_cast<uint4*>(bA + row * 128 + ((chunk ^ (row & 7)) << 4)) = ra[j];
        }
#pragma unroll
        for (int j = 0; j < BJ; ++j) {
            const int row = srow + 32 * j;
            *reinterpret_cast<uint4*>(bB + row * 128 + ((chunk ^ (row & 7)) << 4)) = rb[j];
        }
    };
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
                const int row = wm * (BM / 2) + b * 16 + li;
                bfr[b] = *reinterpret_cast<const bf16x8*>(bB + row * 128 + slot);
            }
#pragma unroll
            for (int a = 0; a < TN; ++a)
#pragma unroll
                for (int b = 0; b < TM; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a], bfr[b], acc[a][b], 0, 0, 0);
        }
    };

    load_tiles(ra0, rb0);                 // tile 0
    load_tiles(ra1, rb1);                 // tile 1 (or tile 0 again when nk == 1)
    store_tiles(0, ra0, rb0);
    __syncthreads();
    for (int kt = 0; kt < nk; kt += 2) {
        // even step: LDS[0] = tile kt, R1 = tile kt+1 (in flight), R0 free
        load_tiles(ra0, rb0);             // tile kt+2
        compute(0);
        store_tiles(1, ra1, rb1);
        __syncthreads();
        if (kt + 1 >= nk) break;
        // odd step: LDS[1] = tile kt+1, R0 = tile kt+2 (in flight), R1 free
        load_tiles(ra1, rb1);             // tile kt+3
        compute(1);
        store_tiles(0, ra0, rb0);
        __syncthreads();
    }

    // ---- epilogue (shared with the skinny kernel) ----
#pragma unroll
    for (int a = 0; a < TN; ++a) {
        const int n = n0 + wn * (BN / 2) + a * 16 + lg * 4;
        if (n >= p.N) continue;
#pragma unroll
        for (int b = 0; b < TM; ++b) {
            const int m = m0 + wm * (BM / 2) + b * 16 + li;
            if (m >= p.M) continue;
            epilogue4(p, m, n, acc[a][b]);
        }
    }
}

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
template <int BM, int BN, int MODE, int NS, int NW = 4>
__device__ __forceinline__ void gemm_dma_body(const bf16_t* __restrict__ src, const bf16_t* __restrict__ wgt, const GemmArgs& p,
                                              const int bx, const int by, const int gx) {
    constexpr int NT = 64 * NW, WM = NW / 2, RPP = NT / 8;        // threads, waves along m, rows one DMA pass covers
    constexpr int TM = BM / (16 * WM), TN = BN / 32;
    constexpr int AJ = BN / RPP, BJ = BM / RPP;
    static_assert(NW == 4 || NW == 8, "4 or 8 waves"); static_assert(AJ >= 1 && BJ >= 1 && TM >= 1, "tile too small for the wave count");
    constexpr int A_BYTES = BN * 128, B_BYTES = BM * 128, BUF_BYTES = A_BYTES + B_BYTES;
    constexpr int LPT = AJ + BJ;                   // DMA instructions per thread per K tile
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef __attribute__((address_space(3))) void* lds_ptr;

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
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

    const int srow = t >> 3;
    const int chunk = (t & 7) ^ (srow & 7);        // source-side swizzle
    constexpr int OOB = 0x7fffffff;

    int a_off[AJ];
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
        const int n = n0 + srow + RPP * j;
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

    auto issue_tile = [&](int buf) __attribute__((always_inline)) {
        const unsigned bA = lds0 + buf * BUF_BYTES;      // this wave's 8-row group of each 32-row slab
        const unsigned bB = bA + A_BYTES;
        const int k0b = MODE == 3 ? ((kh * p.KW + kw) * p.SC + c0) * 2 : lk << 7;
#pragma unroll
        for (int j = 0; j < AJ; ++j) rt_dma16(rs_w, bA + j * (RPP * 128), a_off[j], k0b);
#pragma unroll
        for (int j = 0; j < BJ; ++j) {
            if (MODE == 0) {
                rt_dma16(rs_x, bB + j * (RPP * 128), b_off[j], k0b);
            } else {
                bool ok;
                int sy, sx;
                if (MODE == 1) { sy = b_y[j] + kh; sx = b_x[j] + kw; ok = true; }
                else if (MODE == 3) {
                    const int ny_ = b_y[j] - kh, nx_ = b_x[j] - kw;      // even by construction of the class
                    ok = (ny_ | nx_) >= 0;
                    sy = ny_ >> 1; sx = nx_ >> 1;
                } else {
                    const int ny = b_y[j] - kh, nx = b_x[j] - kw;
                    const int msk = p.stride - 1;
                    ok = ((ny | nx) >= 0) && (((ny | nx) & msk) == 0);
                    sy = ny >> p.sshift; sx = nx >> p.sshift;
                }
                ok = ok && (unsigned)sy < (unsigned)p.SH && (unsigned)sx < (unsigned)p.SW;
                const int off = (b_off[j] + (sy * p.SW + sx) * p.SC) * 2;
                rt_dma16(rs_x, bB + j * (RPP * 128), ok ? off : OOB, c0 * 2);
            }
        }
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
    constexpr int PIECES = ROWS * CPR / NT;
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
            const bool ok = out_piece(i * NT + t, m, n);
            const size_t o = ok ? (size_t)m * p.N + n : 0;
            pre_res[i] = p.res_bf16 ? *reinterpret_cast<const bf16x8*>(p.res_bf16 + o) : bf16x8{};
            pre_gate[i] = p.gate ? *reinterpret_cast<const bf16x8*>(p.gate + o) : bf16x8{};
        }
    }

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
#pragma unroll
        for (int h = 0; h < HALVES; ++h) {
            __syncthreads();    // every wave has drained its own DMA tail (vmcnt 0 above) and finished reading the stages / tile
            if (HALVES == 1 || wm / (WM / 2) == h) {
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

template <int BM, int BN, int MODE, int NS, int MINB, int NW = 4>
__global__ __launch_bounds__(64 * NW, MINB) void conv_gemm_dma_kernel(const bf16_t* __restrict__ src,
                                                                      const bf16_t* __restrict__ wgt, const GemmArgs p) {
    gemm_dma_body<BM, BN, MODE, NS, NW>(src, wgt, p, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.x);
}

// Several independent dense products in ONE launch (descriptors by value in the kernel arguments: graph-safe, no device
// tables): the q/k and v projections of an encoder layer, the twelve cross-attention K / V projections of the decoder, pairs
// of backward-data products.  On the latency-bound transformer chains every launch costs ~4.5 us whatever it computes.
struct GemmGroup { GemmArgs j[12]; int first[13]; int n; };
template <int BM, int BN, int NS, int MINB>
__global__ __launch_bounds__(256, MINB) void conv_gemm_dma_grouped_kernel(const GemmGroup g) {
    int lo = 0;
    for (int i = 1; i < g.n; ++i) if (g.first[i] <= (int)blockIdx.x) lo = i;
    lo = __builtin_amdgcn_readfirstlane(lo);
    const int lin = (int)blockIdx.x - g.first[lo];
    const int gx = g.first[lo + 1] - g.first[lo];
    gemm_dma_body<BM, BN, 0, NS>(g.j[lo].src, g.j[lo].wgt, g.j[lo], lin, 0, gx);
}

// ------------------------------------------------------------------------------------------------
// Skinny path (M <= 16 rows: the decoder / query-encoder / box-head Linears over B*n_q tokens).  No LDS tiles:
// a workgroup owns 16 output features, its 4 waves split K four ways and stream both operands straight from
// global memory into MFMA fragments (A = 16 weight rows, B = the <=16 token rows), then reduce through LDS.
__global__ __launch_bounds__(256) void skinny_gemm_kernel(const bf16_t* __restrict__ src, const bf16_t* __restrict__ wgt,
                                                          const GemmArgs p) {
    __shared__ f32x4 red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int nrow = n0 + li;
    const bool n_ok = nrow < p.N, m_ok = li < p.M;
    const bf16_t* wp = wgt + (size_t)(n_ok ? nrow : 0) * p.K + lg * 8;
    const bf16_t* xp = src + (size_t)(m_ok ? li : 0) * p.K + lg * 8;
    const int ksteps = p.K >> 5;                       // 32-wide steps
    const int per = (ksteps + 3) >> 2;
    const int k_begin = wave * per, k_end = min(k_begin + per, ksteps);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const uint4 z = make_uint4(0, 0, 0, 0);
#pragma unroll 4
    for (int ks = k_begin; ks < k_end; ++ks) {
        uint4 wv = *reinterpret_cast<const uint4*>(wp + ks * 32);
        uint4 xv = *reinterpret_cast<const uint4*>(xp + ks * 32);
        if (!n_ok) wv = z;
        if (!m_ok) xv = z;
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&wv), *reinterpret_cast<bf16x8*>(&xv), acc, 0, 0, 0);
    }
    red[wave][lane] = acc;
    __syncthreads();
    if (wave != 0) return;
    acc = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
    const int n = n0 + lg * 4;
    if (n < p.N && li < p.M) epilogue4(p, li, n, acc);
}

template <int BM, int BN>
int launch_gemm(const GemmArgs& a, hipStream_t s) {
    const int mt = (a.M + BM - 1) / BM, nt = (a.N + BN - 1) / BN;
    const size_t smem = 2 * (size_t)(BM + BN) * 128;
    const dim3 grid((unsigned)(mt * nt)), block(256);
    const bool dense = (a.KH == 1 && a.KW == 1 && a.stride == 1 && a.pad == 0 && a.SH == a.DH && a.SW == a.DW);
    if (dense)
        hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, 0>), grid, block, smem, s, a.src, a.wgt, a);
    else if (!a.transposed)
        hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, 1>), grid, block, smem, s, a.src, a.wgt, a);
    else
        hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, 2>), grid, block, smem, s, a.src, a.wgt, a);
    RT_CHECK_LAUNCH();
    return RT_OK;
}

template <int BM, int BN, int NS, int MINB, int NW = 4>
int launch_gemm_dma(const GemmArgs& a, hipStream_t s) {
    const int mt = (a.M + BM - 1) / BM, nt = (a.N + BN - 1) / BN;
    const size_t smem = (size_t)NS * (BM + BN) * 128;
    const dim3 grid((unsigned)(mt * nt)), block(64 * NW);
    const bool dense = (a.KH == 1 && a.KW == 1 && a.stride == 1 && a.pad == 0 && a.SH == a.DH && a.SW == a.DW);
    static const int par_env = getenv("REFTR_S2PARITY") ? atoi(getenv("REFTR_S2PARITY")) : 1;
    auto set_smem = [&](const void* f) {
        if (smem > 65536) (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    };
    if (dense) {
        set_smem((const void*)conv_gemm_dma_kernel<BM, BN, 0, NS, MINB, NW>);
        // distinct operand slabs an XCD's run of R tiles touches: R / n_tiles + min(R, n_tiles) (n fastest) vs the same with m_tiles
        static const int mfast_env = getenv("REFTR_MFAST") ? atoi(getenv("REFTR_MFAST")) : 1;
        GemmArgs am = a;
        const double R = (double)(mt * nt) / 8.0;
        const double cn = R / nt + (R < nt ? R : nt), cm = R / mt + (R < mt ? R : mt);
        am.mfast = (mfast_env && a.xcd && mt * nt >= 16 && cm < cn) ? 1 : 0;
        hipLaunchKernelGGL((conv_gemm_dma_kernel<BM, BN, 0, NS, MINB, NW>), grid, block, smem, s, a.src, a.wgt, am);
    } else if (!a.transposed) {
        set_smem((const void*)conv_gemm_dma_kernel<BM, BN, 1, NS, MINB, NW>);
        hipLaunchKernelGGL((conv_gemm_dma_kernel<BM, BN, 1, NS, MINB, NW>), grid, block, smem, s, a.src, a.wgt, a);
    } else if (a.stride == 2 && par_env) {
        const int m_cls = a.B * ((a.DH + 1) / 2) * ((a.DW + 1) / 2);           // largest parity class
        // grid.x padded to a multiple of 8: block (x, y) has linear id y * gridDim.x + x, so only then do the four parity classes
        // of a tile range (they gather from the same dy rows) sit on the same XCD as the tile map assumes (surplus blocks exit)
        const dim3 grid3((unsigned)(((((m_cls + BM - 1) / BM) * nt) + 7) / 8 * 8), 4);
        set_smem((const void*)conv_gemm_dma_kernel<BM, BN, 3, NS, MINB, NW>);
        hipLaunchKernelGGL((conv_gemm_dma_kernel<BM, BN, 3, NS, MINB, NW>), grid3, block, smem, s, a.src, a.wgt, a);
    } else {
        set_smem((const void*)conv_gemm_dma_kernel<BM, BN, 2, NS, MINB, NW>);
        hipLaunchKernelGGL((conv_gemm_dma_kernel<BM, BN, 2, NS, MINB, NW>), grid, block, smem, s, a.src, a.wgt, a);
    }
    RT_CHECK_LAUNCH();
    return RT_OK;
}

}  // namespace

static int fill_gemm_args(const rt_conv_gemm_desc* d, GemmArgs& a) {
    if (!d || !d->src || !d->wgt || (!d->out_bf16 && !d->out_f32)) return RT_ERR_BADARG;
    if (d->SC <= 0 || (d->SC & 63) || (d->N & 3) || d->N <= 0) return RT_ERR_UNSUPPORTED;
    if (d->stride != 1 && d->stride != 2) return RT_ERR_UNSUPPORTED;
    if (d->KH <= 0 || d->KW <= 0 || d->B <= 0 || d->DH <= 0 || d->DW <= 0) return RT_ERR_BADARG;
    a.src = (const bf16_t*)d->src; a.wgt = (const bf16_t*)d->wgt;
    a.out_bf16 = (bf16_t*)d->out_bf16; a.out_f32 = d->out_f32; a.out_preact = (bf16_t*)d->out_preact;
    a.bias = d->bias; a.res_f32 = d->res_f32; a.res_bf16 = (const bf16_t*)d->res_bf16;
    a.gate = (const bf16_t*)d->gate; a.preact = (const bf16_t*)d->preact; a.dtanh = (const bf16_t*)d->dtanh;
    a.B = d->B; a.SH = d->SH; a.SW = d->SW; a.SC = d->SC; a.DH = d->DH; a.DW = d->DW; a.N = d->N;
    a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad = d->pad; a.transposed = d->transposed;
    a.act = d->act; a.res_first = d->res_first; a.gate_scale = d->gate_scale; a.drop_p = d->drop_p; a.drop_seed = d->drop_seed; a.seed_dev = d->seed_dev;
    a.drop_shift = d->drop_shift;
    a.acc2_f32 = d->acc2_f32;
    if (a.drop_shift < 0 || a.drop_shift > 16) return RT_ERR_BADARG;
    const long long M = (long long)d->B * d->DH * d->DW;
    if (M > 0x7fffffffLL / 4) return RT_ERR_UNSUPPORTED;
    // 32-bit element offsets inside the kernel
    if ((long long)d->B * d->SH * d->SW * d->SC >= 0x3fffffffLL || (long long)d->N * d->KH * d->KW * d->SC >= 0x3fffffffLL) return RT_ERR_UNSUPPORTED;
    a.M = (int)M; a.K = d->KH * d->KW * d->SC; a.sshift = d->stride == 2 ? 1 : 0;
    static const int xcd_env = getenv("REFTR_XCD") ? atoi(getenv("REFTR_XCD")) : 1;
    a.xcd = xcd_env;
    static const int early_env = getenv("REFTR_EARLY") ? atoi(getenv("REFTR_EARLY")) : 3;
    a.early = early_env & 1;
    static const int epi_env = getenv("REFTR_EPI") ? atoi(getenv("REFTR_EPI")) : 3;
    a.epi_lds = epi_env & 1;
    static const int abl_env = getenv("REFTR_GEMM_ABL") ? atoi(getenv("REFTR_GEMM_ABL")) : 0;   // 1 no loads, 2 no MFMA, 4 no epilogue
    a.abl = abl_env;
    static const int pre_env = getenv("REFTR_EPI_PREFETCH") ? atoi(getenv("REFTR_EPI_PREFETCH")) : 1;
    a.prefetch = pre_env;
    a.mfast = 0;
    a.src_bytes = (unsigned)((long long)d->B * d->SH * d->SW * d->SC * 2);
    a.wgt_bytes = (unsigned)((long long)d->N * d->KH * d->KW * d->SC * 2);
    return RT_OK;
}

extern "C" int rt_conv_gemm(const rt_conv_gemm_desc* d, rt_stream_t stream) {
    GemmArgs a;
    const int frc = fill_gemm_args(d, a);
    if (frc != RT_OK) return frc;
    hipStream_t s = (hipStream_t)stream;

    const bool dense = (a.KH == 1 && a.KW == 1 && a.stride == 1 && a.pad == 0 && a.SH == a.DH && a.SW == a.DW);
    if (d->tile_hint == 0 && dense && a.M <= 16) {
        hipLaunchKernelGGL(skinny_gemm_kernel, dim3((unsigned)((a.N + 15) / 16)), dim3(256), 0, s, a.src, a.wgt, a);
        RT_CHECK_LAUNCH();
        return RT_OK;
    }
    int hint = d->tile_hint;
    if (hint == 0) {
        // Tile choice from the in-step sweeps (benchmarks/tile_sweep.py, profiles/r01e_tile_sweep.txt): small K streams
        // best through many 64x64 workgroups; 128x128 needs >= 1.5 waves of tiles over the 256 CUs to pay off.
        const long long t128 = (long long)((a.M + 127) / 128) * ((a.N + 127) / 128);
        const long long t12864 = (long long)((a.M + 127) / 128) * ((a.N + 63) / 64);
        static const int smallk = getenv("REFTR_SMALLK") ? atoi(getenv("REFTR_SMALLK")) : 256;
        static const int dma = getenv("REFTR_DMA") ? atoi(getenv("REFTR_DMA")) : 1;
        if (a.K <= smallk) hint = 3;
        else if (a.N > 64 && t128 >= 384) hint = 1;
        else if (t12864 >= 256) hint = 2;
        else hint = 3;
        static const int tilev = getenv("REFTR_TILEV") ? atoi(getenv("REFTR_TILEV")) : 2;
        if (dma && tilev >= 2) {
            // round 2 (profiles/r02_tile_sweep_8wave.txt): the 128x128 tile runs on 8-wave workgroups (2 x 4 waves, 16 waves per CU
            // at two workgroups: beats the 4-wave form on every shape); it takes over the long reductions with >= 1.5 rounds of
            // tiles (or >= 0.75 rounds from K = 2048 up) and the short ones whose tiles fill exactly one round of 2 per CU
            const long long t256 = (long long)((a.M + 255) / 256) * ((a.N + 127) / 128);
            if (a.K < 1024) hint = (a.N >= 128 && t128 >= 384 && t128 <= 512) ? 51 : 31;
            else if (a.K >= 2048 && a.N >= 128 && t256 >= 512) hint = 62;     // big products only (4096^3: 898 TF/s); none in the step
            else if (a.N > 64 && (t128 >= 384 || (a.K >= 2048 && t128 >= 192))) hint = 51;
            else if (t12864 >= 256) hint = 21;
            else hint = 33;
        } else if (dma && tilev) {
            // with the issue-before-wait schedule (profiles/r01g_tile_sweep_early.txt): short K streams best through
            // 64x64 workgroups; 128x128 pays off from K >= 1024 with >= 1.5 waves of tiles; 128x64 in between
            if (a.K < 1024) hint = 31;
            else if (a.N > 64 && t128 >= 384) hint = 11;
            else if (t12864 >= 256) hint = 21;
            else hint = 33;
        } else if (dma) {   // previous choice (A/B)
            if (hint == 1) hint = 11;
            else if (hint == 2) hint = a.K >= 1024 ? 22 : 21;
            else hint = a.K >= 1024 ? 33 : 31;
        }
    }
    switch (hint) {
        case 1: return launch_gemm<128, 128>(a, s);
        case 2: return launch_gemm<128, 64>(a, s);
        case 3: return launch_gemm<64, 64>(a, s);
        // LDS-DMA variants (tile, stages, min workgroups / CU)
        case 11: return launch_gemm_dma<128, 128, 2, 2>(a, s);
        case 12: return launch_gemm_dma<128, 128, 3, 2>(a, s);
        case 13: return launch_gemm_dma<128, 128, 4, 2>(a, s);
        case 21: return launch_gemm_dma<128, 64, 2, 2>(a, s);
        case 22: return launch_gemm_dma<128, 64, 3, 2>(a, s);
        case 31: return launch_gemm_dma<64, 64, 2, 4>(a, s);
        case 32: return launch_gemm_dma<64, 64, 4, 2>(a, s);
        case 33: return launch_gemm_dma<64, 64, 3, 3>(a, s);
        // 8-wave workgroups (2 x 4 waves) on the 128-row tiles
        case 51: return launch_gemm_dma<128, 128, 2, 2, 8>(a, s);
        case 52: return launch_gemm_dma<128, 128, 3, 1, 8>(a, s);
        case 53: return launch_gemm_dma<128, 64, 2, 2, 8>(a, s);
        case 54: return launch_gemm_dma<128, 64, 3, 2, 8>(a, s);
        case 61: return launch_gemm_dma<256, 128, 2, 1, 8>(a, s);
        case 62: return launch_gemm_dma<256, 128, 3, 1, 8>(a, s);
        case 63: return launch_gemm_dma<128, 256, 2, 1, 8>(a, s);
        default: return RT_ERR_BADARG;
    }
}

extern "C" int rt_conv_gemm_grouped(const rt_conv_gemm_desc* descs, int n, rt_stream_t stream) {
    if (!descs || n <= 0) return RT_ERR_BADARG;
    hipStream_t s = (hipStream_t)stream;
    static const int grp_env = getenv("REFTR_GEMM_GROUP") ? atoi(getenv("REFTR_GEMM_GROUP")) : 1;
    // groupable: dense products the single-launch heuristic would give the 64x64 / 2-stage / 4-per-CU variant (K < 1024, M > 16)
    bool ok = grp_env && n >= 2 && n <= 12;
    GemmGroup g;
    int blocks = 0;
    for (int i = 0; i < n && ok; ++i) {
        const rt_conv_gemm_desc& d = descs[i];
        GemmArgs a;
        const int rc = fill_gemm_args(&d, a);
        if (rc != RT_OK) return rc;
        const bool dense = (a.KH == 1 && a.KW == 1 && a.stride == 1 && a.pad == 0 && a.SH == a.DH && a.SW == a.DW);
        if (!dense || a.M <= 16 || a.K >= 1024 || d.tile_hint != 0) { ok = false; break; }
        g.j[i] = a; g.first[i] = blocks;
        blocks += ((a.M + 63) / 64) * ((a.N + 63) / 64);
    }
    if (!ok) {                                   // anything else: the same products as single launches, in order
        for (int i = 0; i < n; ++i) { const int rc = rt_conv_gemm(descs + i, stream); if (rc != RT_OK) return rc; }
        return RT_OK;
    }
    g.first[n] = blocks; g.n = n;
    for (int i = n + 1; i < 13; ++i) g.first[i] = blocks;
    constexpr size_t smem = (size_t)2 * (64 + 64) * 128;
    hipLaunchKernelGGL((conv_gemm_dma_grouped_kernel<64, 64, 2, 4>), dim3((unsigned)blocks), dim3(256), smem, s, g);
    RT_CHECK_LAUNCH();
    return RT_OK;
}

extern "C" int rt_abi_version(void) { return 22; }

extern "C" int rt_device_arch(int dev, char* buf, int buflen) {
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, dev);
    if (e != hipSuccess) return (int)e;
    if (!buf || buflen <= 0) return RT_ERR_BADARG;
    int i = 0;
    for (; i < buflen - 1 && prop.gcnArchName[i]; ++i) buf[i] = prop.gcnArchName[i];
    buf[i] = 0;
    return RT_OK;
}
