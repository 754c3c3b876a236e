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
#include "rt_gemm_dma.h"
#include <stdio.h>

namespace {

#ifdef RT_LAB      // the register-staged tiles of round 1 (hints 1-3): A/B baseline of the sweeps, lab library only
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

#endif  // RT_LAB

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
// Round 3: these launches are latency chains (86 per step at ~5 us for <= 1 MB of weights each), so the kernel is built around
// round trips: (1) the epilogue's operands (bias, residuals, gates) are requested FIRST -- they depend on nothing computed here;
// (2) a wave requests its whole K slice (CH k-steps = 2 * CH loads in flight; bounds-checked buffer loads, so ragged N / M rows
// and steps past the slice are out-of-range zeros and the code is branch-free) before the first MFMA instead of 4 steps at a
// time: linear2 of a decoder layer (K = 2048 over 16 workgroups) went from four dependent L2/HBM round trips to two.
template <int CH>
__global__ __launch_bounds__(256) void skinny_gemm_kernel(const bf16_t* __restrict__ src, const bf16_t* __restrict__ wgt,
                                                          const GemmArgs p) {
    __shared__ f32x4 red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int nrow = n0 + li;
    const int n = n0 + lg * 4;
    const bool out_ok = wave == 0 && n < p.N && li < p.M;
    Epi4Pre pre;
    if (out_ok) pre = epi4_prefetch(p, li, n);
    constexpr int OOB = 0x7fffffff;
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(wgt), 0, p.wgt_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(src), 0, p.src_bytes, 0x00020000);
    const int w_off = nrow < p.N ? (nrow * p.K + lg * 8) * 2 : OOB;
    const int x_off = li < p.M ? (li * p.K + lg * 8) * 2 : OOB;
    const int ksteps = p.K >> 5;                       // 32-wide steps
    const int per = (ksteps + 3) >> 2;
    const int k_begin = wave * per, k_end = min(k_begin + per, ksteps);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    for (int ks = k_begin; ks < k_end; ks += CH) {
        u32x4 wv[CH], xv[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const bool in = ks + j < k_end;
            wv[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, in ? w_off : OOB, (ks + j) << 6, 0);
            xv[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, in ? x_off : OOB, (ks + j) << 6, 0);
        }
#pragma unroll
        for (int j = 0; j < CH; ++j)
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&wv[j]), *reinterpret_cast<bf16x8*>(&xv[j]), acc, 0, 0, 0);
    }
    red[wave][lane] = acc;
    __syncthreads();
    if (wave != 0) return;
    acc = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
    if (out_ok) epilogue4<true>(p, li, n, acc, &pre);
}

// round-2 form, kept as the A/B baseline (REFTR_SKINNY_V=1): 4 k-steps in flight, epilogue operands loaded after the reduction
__global__ __launch_bounds__(256) void skinny_gemm_kernel_v1(const bf16_t* __restrict__ src, const bf16_t* __restrict__ wgt,
                                                             const GemmArgs p) {
    __shared__ f32x4 red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int nrow = n0 + li;
    const bool n_ok = nrow < p.N, m_ok = li < p.M;
    const bf16_t* wp = wgt + (size_t)(n_ok ? nrow : 0) * p.K + lg * 8;
    const bf16_t* xp = src + (size_t)(m_ok ? li : 0) * p.K + lg * 8;
    const int ksteps = p.K >> 5;
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

#ifdef RT_LAB
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

#endif  // RT_LAB

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
    a.dil = d->dil > 1 ? d->dil : 1;
    if (a.dil > 1 && d->stride != 1) return RT_ERR_UNSUPPORTED;
    if (a.drop_shift < 0 || a.drop_shift > 16) return RT_ERR_BADARG;
    const long long M = (long long)d->B * d->DH * d->DW;
    if (M > 0x7fffffffLL / 4) return RT_ERR_UNSUPPORTED;
    // 32-bit element offsets inside the kernel
    if ((long long)d->B * d->SH * d->SW * d->SC >= 0x3fffffffLL || (long long)d->N * d->KH * d->KW * d->SC >= 0x3fffffffLL) return RT_ERR_UNSUPPORTED;
    a.M = (int)M; a.K = d->KH * d->KW * d->SC; a.sshift = d->stride == 2 ? 1 : 0;
    static const int xcd_env = RT_TUNE("REFTR_XCD", 1);
    a.xcd = xcd_env;
    static const int early_env = RT_TUNE("REFTR_EARLY", 3);
    a.early = early_env & 1;
    static const int epi_env = RT_TUNE("REFTR_EPI", 3);
    a.epi_lds = epi_env & 1;
#ifdef RT_LAB                                      // lab builds only (hipcc -DRT_LAB): 1 no loads, 2 no MFMA, 4 no epilogue -- wrong results
    static const int abl_env = RT_TUNE("REFTR_GEMM_ABL", 0);
    a.abl = abl_env;
#else
    a.abl = 0;
#endif
    static const int pre_env = RT_TUNE("REFTR_EPI_PREFETCH", 1);
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
        const dim3 grid((unsigned)((a.N + 15) / 16));
        const int per = ((a.K >> 5) + 3) >> 2;          // k-steps per wave: the whole slice in flight when it fits 8 steps
        static const int skinny_v = RT_TUNE("REFTR_SKINNY_V", 2);
        if (skinny_v == 1) hipLaunchKernelGGL(skinny_gemm_kernel_v1, grid, dim3(256), 0, s, a.src, a.wgt, a);
        else if (per <= 2) hipLaunchKernelGGL(skinny_gemm_kernel<2>, grid, dim3(256), 0, s, a.src, a.wgt, a);
        else if (per <= 4) hipLaunchKernelGGL(skinny_gemm_kernel<4>, grid, dim3(256), 0, s, a.src, a.wgt, a);
        else hipLaunchKernelGGL(skinny_gemm_kernel<8>, grid, dim3(256), 0, s, a.src, a.wgt, a);
        RT_CHECK_LAUNCH();
        return RT_OK;
    }
    int hint = d->tile_hint;
    if (a.dil > 1 && hint >= 1 && hint <= 3) return RT_ERR_UNSUPPORTED;      // the register-staged tiles have no dilation
    if (hint == 0) {
        // Tile choice from the in-step sweeps (benchmarks/tile_sweep.py, profiles/r01e_tile_sweep.txt): small K streams
        // best through many 64x64 workgroups; 128x128 needs >= 1.5 waves of tiles over the 256 CUs to pay off.
        const long long t128 = (long long)((a.M + 127) / 128) * ((a.N + 127) / 128);
        const long long t12864 = (long long)((a.M + 127) / 128) * ((a.N + 63) / 64);
        static const int smallk = RT_TUNE("REFTR_SMALLK", 256);
        static const int dma = RT_TUNE("REFTR_DMA", 1);
        if (a.K <= smallk) hint = 3;
        else if (a.N > 64 && t128 >= 384) hint = 1;
        else if (t12864 >= 256) hint = 2;
        else hint = 3;
        static const int tilev = RT_TUNE("REFTR_TILEV", 3);
        if (dma && tilev >= 3) {
            // round 3 (profiles/r03_tile_sweep_warm.txt / _cold.txt, r03_instep_ab.txt).  Two lessons: (1) back-to-back launches of
            // one shape on warm caches are a misleading yardstick -- the software-pipelined K loop (hints 2xx: fragments of tile
            // kt+1 read under the MFMAs of tile kt, one barrier per K tile) wins 8-20 % there and LOSES in the step, where the
            // weights come from HBM and the kernels are bound by bytes in flight per CU; the sweep now has a cold mode (FLUSH=1)
            // and only what wins in both is adopted: the pipelined 3-stage forms (64 x 64 on the layer4-sized 3x3 convolutions,
            // 128 x 128 at ONE workgroup per CU where the tiles do not fill two per CU anyway: layer3's 3x3).  (2) The M = B * L
            // Linears of the language branch (<= 96 tiles of 64 x 64: 60 CUs pulling at ~50 GB/s each) run on 32 x 32 tiles
            // (4x the workgroups), the other few-tile products on the 3-stage 64 x 64 tile also for K < 1024.
            static const int smallt = RT_TUNE("REFTR_SMALLT", 1);
            static const int pipe = RT_TUNE("REFTR_PIPE", 3);         // bit 0: 128x128 / 3 stages, bit 1: 64x64 / 3 stages
            const long long t64 = (long long)((a.M + 63) / 64) * ((a.N + 63) / 64);
            const long long t256 = (long long)((a.M + 255) / 256) * ((a.N + 127) / 128);
            // round 4 (profiles/r04o_deep_stage_cold.txt): at <= 1 workgroup per CU the 32 x 32 form is bound by the K tiles it keeps in
            // flight -- 6 stages instead of 3 take the cold K >= 768 products from 17.3 / 14.6 / 7.8 us to 11.1 / 9.9 / 5.8 (8 stages: no better)
            static const int deep = RT_TUNE("REFTR_DEEP", 1);
            static const int k1024 = RT_TUNE("REFTR_K1024", 1);
            if (dense && smallt && a.M <= 1024 && t64 < 256 && (a.N & 7) == 0) hint = t64 <= 96 ? ((deep && a.K >= 512) ? 285 : 281) : 33;
            else if (a.K < 1024) hint = (a.N >= 128 && t128 >= 384 && t128 <= 512) ? 51 : 31;
            else if (dense && a.K >= 2048 && a.N >= 128 && t256 >= 512) hint = 262;     // big products only; none in the step
            else if (a.N > 64 && (t128 >= 384 || (a.K >= 2048 && t128 >= 192))) hint = (!dense && (pipe & 1) && a.K >= 2048 && t128 <= 256) ? 252 : 51;
            else if (k1024 && dense && a.N > 64 && a.K >= 1024 && t128 >= 192) hint = 51;      // layer3's 1024 -> 256 (200 tiles): 18.6-19.1 us cold against 21.7-22.9 on hint 21
            else if (t12864 >= 256) hint = 21;
            else hint = (!dense && (pipe & 2)) ? 233 : 33;
            // round 6 (profiles/r06c_sweep_*.txt): the K-parity ping-pong 128 x 128 form where ONE round of <= 256 tiles walks a long
            // reduction -- layer3's 3 x 3 convolutions (K = 2304: 252 before) and its 1024 -> 256 products (K = 1024, 200 tiles: 21 before)
            static const int pp = RT_TUNE("REFTR_PP", 0);      // off: loses 0.07-0.2 ms inside the step (profiles/r06_pingpong_gemm.txt)
            if ((a.N & 7) == 0 && a.epi_lds && a.N >= 128) {
                if ((pp & 1) && hint == 252) hint = (pp & 4) ? 351 : 352;
                else if ((pp & 2) && dense && a.K >= 1024 && t128 >= 128 && t128 <= 256) hint = (pp & 4) ? 351 : 352;
            }
#ifdef RT_LAB
            // round 6: the short-K / wide-N dense products (a bottleneck's conv3 and the backward-data of its conv1, the encoder's
            // linear1 and the backward-data of its linear2) are epilogue-bound; the activation-stationary form (rt_gemm_astat.hip) was
            // built for them and is SLOWER (28.2 vs 25.0 us cold on layer3's 256 -> 1024, +0.17 ms in the step): lab switch, off
            static const int astat = RT_TUNE("REFTR_ASTAT", 0);
            if (astat && rt_gemm_astat_ok(a) && a.M >= 1024 && a.N >= 2 * a.K && a.N >= 256) hint = 501;
#endif
        } else if (dma && tilev >= 2) {
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
    if (a.dil > 1 && hint >= 1 && hint <= 3) return RT_ERR_UNSUPPORTED;      // (REFTR_DMA=0)
#ifdef RT_LAB
    // in-step autotune (benchmarks/instep_autotune.py): REFTR_HINT_OVERRIDE = "transposed,KH,stride,M,K,N=hint;..." replaces the heuristic's
    // choice for exactly those launches; read on every call so that the script can re-capture the step with another table
    if (d->tile_hint == 0) {
        const char* ov = getenv("REFTR_HINT_OVERRIDE");
        while (ov && *ov) {
            int t_, kh_, s_, m_, k_, n_, h_, used = 0;
            if (sscanf(ov, "%d,%d,%d,%d,%d,%d=%d%n", &t_, &kh_, &s_, &m_, &k_, &n_, &h_, &used) == 7) {
                if (t_ == (a.transposed ? 1 : 0) && kh_ == a.KH && s_ == a.stride && m_ == a.M && k_ == a.K && n_ == a.N) { hint = h_; break; }
                ov += used;
            }
            while (*ov && *ov != ';') ++ov;
            if (*ov == ';') ++ov;
        }
    }
#endif
    // The product library instantiates only the variants its (constant) heuristics can choose; every other tile / stage / schedule
    // variant that was built and measured (LAB_NOTES.md, profiles/*tile_sweep*) lives in the LAB library (-DRT_LAB), where the
    // sweeps and tests/test_gemm_gpu.py reach it through tile_hint.  (Round 6: libreftr_hip.so 16.3 MB -> see DESIGN.md section 5.)
    switch (hint) {
        // LDS-DMA variants (tile, stages, min workgroups / CU[, waves])
        case 21: return launch_gemm_dma<128, 64, 2, 2>(a, s);
        case 31: return launch_gemm_dma<64, 64, 2, 4>(a, s);
        case 33: return launch_gemm_dma<64, 64, 3, 3>(a, s);
        case 51: return launch_gemm_dma<128, 128, 2, 2, 8>(a, s);
        case 233: case 252: case 262: case 281: case 285: return rt_launch_gemm_pipe(a, hint, s);
#ifdef RT_LAB
        case 501: return rt_launch_gemm_astat(a, s);
        case 1: return launch_gemm<128, 128>(a, s);          // register-staged tiles (round 1)
        case 2: return launch_gemm<128, 64>(a, s);
        case 3: return launch_gemm<64, 64>(a, s);
        case 11: return launch_gemm_dma<128, 128, 2, 2>(a, s);
        case 12: return launch_gemm_dma<128, 128, 3, 2>(a, s);
        case 13: return launch_gemm_dma<128, 128, 4, 2>(a, s);
        case 22: return launch_gemm_dma<128, 64, 3, 2>(a, s);
        case 32: return launch_gemm_dma<64, 64, 4, 2>(a, s);
        // 8-wave workgroups (2 x 4 waves) on the 128-row tiles
        case 52: return launch_gemm_dma<128, 128, 3, 1, 8>(a, s);
        case 53: return launch_gemm_dma<128, 64, 2, 2, 8>(a, s);
        case 54: return launch_gemm_dma<128, 64, 3, 2, 8>(a, s);
        case 61: return launch_gemm_dma<256, 128, 2, 1, 8>(a, s);
        case 62: return launch_gemm_dma<256, 128, 3, 1, 8>(a, s);
        case 63: return launch_gemm_dma<128, 256, 2, 1, 8>(a, s);
        case 211: case 221: case 231: case 251: case 261:
        case 81: case 282: case 283: case 284: case 286: case 287: case 288: case 234: case 236: return rt_launch_gemm_pipe(a, hint, s);
        case 351: case 321: case 323: case 331: case 352: case 322: case 332: return rt_launch_gemm_pp(a, hint, s);
#endif
        default: return RT_ERR_BADARG;
    }
}

extern "C" int rt_conv_gemm_grouped(const rt_conv_gemm_desc* descs, int n, rt_stream_t stream) {
    if (!descs || n <= 0) return RT_ERR_BADARG;
    hipStream_t s = (hipStream_t)stream;
    static const int grp_env = RT_TUNE("REFTR_GEMM_GROUP", 1);
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

extern "C" int rt_abi_version(void) { return 33; }

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
