// rt_conv_gemm, activation-stationary form for the SHORT-K / WIDE-N dense products (round 6; tile hints 501 / 502):
//   out[m, n] = epilogue( sum_k x[m, k] * w[n, k] ),   K = 128 or 256,  N = 4 ... 16 x K
// -- a bottleneck's conv3 (c -> 4c) and the backward-data of its conv1, the encoder's linear1 (256 -> 2048) and the backward-data of
// its linear2.  Why they need their own form (profiles/r06e_abl_lin.txt, cold launches like the step's): these launches move 10-30 x more
// bytes through their EPILOGUE (output + residual / ReLU-gate rows) than through their operands, and the tile kernels run load -> MFMA ->
// epilogue once per workgroup with every workgroup in the same phase -- layer3's 256 -> 1024: 25.5 us = 8 skeleton + 3 operand DMA + 12.5
// epilogue (52 MB at HBM speed, during which nothing else happens) against ~13 us for the launch's 59 MB at 4.5 TB/s.
// Here a workgroup (4 waves) owns 64 rows of x for the WHOLE reduction -- their MFMA fragments stay in registers (K / 4 VGPRs per lane) --
// and walks a range of 64-column output tiles: per tile the weight slab (64 x K, LDS-DMA, two stages, source-side XOR swizzle as in
// rt_gemm_dma.h), 2 * K / 8 MFMAs per wave, and the tile's epilogue (wave-private LDS transpose -> 16-B row pieces, the tile's residual /
// gate pieces requested one tile ahead) -- so output stores, residual reads and weight reads of DIFFERENT tiles are in flight together for
// the whole life of the workgroup, x is read once (not once per column tile), and a launch is ONE round of <= 2 workgroups per CU with
// one barrier per tile.
// RESULT (profiles/r06_astat_gemm.txt): correct (tests/test_gemm_gpu.py, hint 501) and slower -- a tile costs ~2 us, not ~0.35: the one
// counter gfx950 has for loads AND stores (vmcnt, in order) makes the per-tile wait for the next slab also a wait for the previous tile's
// stores and for the residual rows requested one tile ago, i.e. one far-memory round trip per tile per workgroup; a deeper prefetch only
// moves the same wait.  The way out is separate epilogue waves (own counters) fed through LDS; not built.
#include "rt_gemm_dma.h"

#ifdef RT_LAB        // measured slower than the tile kernels, alone and in the step (profiles/r06_astat_gemm.txt): lab library only
namespace {

template <int KB>        // K = 64 * KB
__global__ __launch_bounds__(256, 2) void gemm_astat_kernel(const bf16_t* __restrict__ src, const bf16_t* __restrict__ wgt, const GemmArgs p,
                                                           const int m_tiles, const int tiles_per_split) {
    constexpr int BM = 64, BN = 64;
    constexpr int STAGE = BN * KB * 128;                       // a weight slab: KB k-blocks of [64 rows][128 B]
    constexpr int LPT = 2 * KB;                                // DMA pieces per thread per slab (4 waves x 1 KB per round)
    constexpr int EPI = 4096;                                  // wave-private fp32 transpose tile: 32 rows x 128 B, 16-B slot ^= row & 7
    constexpr int OOB = 0x7fffffff;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef __attribute__((address_space(3))) void* lds_ptr;

    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wn = wave & 1, wm = wave >> 1;
    const int li = lane & 15, lg = lane >> 4;
    const int bid = rt_xcd_remap((int)blockIdx.x, (int)gridDim.x, p.xcd);
    // consecutive ids share the column range (the weight slabs) on one XCD's L2
    const int split = bid / m_tiles, tile_m = bid - split * m_tiles;
    const int m0 = tile_m * BM;
    const int n_tiles = (p.N + BN - 1) / BN;
    const int jt0 = split * tiles_per_split;
    const int nt = min(tiles_per_split, n_tiles - jt0);       // this workgroup's column tiles jt0 .. jt0 + nt - 1
    if (nt <= 0) return;
    const int K = KB * 64;

    const i32x4 rs_w = rt_make_rsrc(wgt, p.wgt_bytes), rs_x = rt_make_rsrc(src, p.src_bytes);
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr)smem;
    // DMA geometry: round r of a slab covers rows 32 * (r & 1) + 8 * wave + (lane >> 3) of k-block r >> 1
    const int drow = lane >> 3;
    const int dchunk = (lane & 7) ^ (drow & 7);                // source-side swizzle (rows are 8-aligned per wave instruction)
    auto issue_slab = [&](const i32x4 rs, int row0, int rows_end, int stage) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < LPT; ++r) {
            const int row = (r & 1) * 32 + wave * 8 + drow, kb = r >> 1;
            const int g = row0 + row;
            const int off = g < rows_end ? (g * K + kb * 64 + dchunk * 8) * 2 : OOB;
            rt_dma16(rs, lds0 + stage * STAGE + kb * 8192 + ((r & 1) * 32 + wave * 8) * 128, off, 0);
        }
    };

    // ---- L2 prefetch of this workgroup's epilogue operands (bf16 residual / gate rows of its 64 x (64 nt) output block): one 4-byte LDS-DMA
    // touch per 128-B line into a scratch word, all of them up front -- they land with the x tile and the first slab (one HBM round trip for
    // the workgroup instead of one per column tile: the loop's own requests, one tile ahead, then find their lines in this XCD's L2)
    if (!(p.abl & 64)) {                                       // (lab: REFTR_GEMM_ABL bit 64 switches the touches off, results unchanged)
        auto touch = [&](const void* base) __attribute__((always_inline)) {
            if (base == nullptr) return;
            const i32x4 rs = rt_make_rsrc(base, 0x7ffffffc);
            const unsigned scratch = lds0 + 2 * STAGE + wave * EPI;
            for (int i = t; i < 64 * nt; i += 256) {
                const int m = m0 + (i & 63), n = (jt0 + (i >> 6)) * BN;
                const int off = (m < p.M && n < p.N) ? (m * p.N + n) * 2 : OOB;
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds" ::"s"(scratch), "v"(off), "s"(rs) : "memory", "m0");
            }
        };
        touch(p.res_bf16);
        touch(p.gate);
    }
    // ---- x rows m0 .. m0 + 63 (stage 1) and the first weight slab (stage 0)
    issue_slab(rs_x, m0, p.M, 1);
    issue_slab(rs_w, jt0 * BN, p.N, 0);
    rt_wait_vmcnt<0>();
    __syncthreads();
    bf16x8 xf[KB][2][2];                                       // [k-block][k-half][m fragment]: this wave's 32 rows, stationary
    {
        const unsigned char* xs = smem + STAGE;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int slot = ((kk * 4 + lg) ^ (li & 7)) << 4;
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    xf[kb][kk][b] = *reinterpret_cast<const bf16x8*>(xs + kb * 8192 + (wm * 32 + b * 16 + li) * 128 + slot);
            }
    }
    __syncthreads();                                           // stage 1 may be overwritten

    // epilogue pieces of this lane: rows wm * 32 + (lane >> 2) + 16 * i, 8 columns at wn * 32 + (lane & 3) * 8
    const int er = lane >> 2, ec = (lane & 3) * 8;
    unsigned char* etile = smem + 2 * STAGE + wave * EPI;
    const bool want_res = p.res_bf16 != nullptr, want_gate = p.gate != nullptr, want_bias = p.bias != nullptr;
    struct EpiPre { bf16x8 res[2], gate[2]; f32x4 b0, b1; };
    auto prefetch_epi = [&](int jt, EpiPre& e) __attribute__((always_inline)) {
        const int n = jt * BN + wn * 32 + ec;
        const bool nok = n < p.N;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = m0 + wm * 32 + er + 16 * i;
            const size_t o = (m < p.M && nok) ? (size_t)m * p.N + n : 0;
            e.res[i] = want_res ? *reinterpret_cast<const bf16x8*>(p.res_bf16 + o) : bf16x8{};
            e.gate[i] = want_gate ? *reinterpret_cast<const bf16x8*>(p.gate + o) : bf16x8{};
        }
        const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
        e.b0 = want_bias ? *reinterpret_cast<const f32x4*>(p.bias + (nok ? n : 0)) : z;
        e.b1 = want_bias ? *reinterpret_cast<const f32x4*>(p.bias + (nok ? n : 0) + 4) : z;
    };
    EpiPre cur, nxt;
    prefetch_epi(jt0, cur);
    GemmArgs pe = p;                                           // the epilogue proper: the bias is added here from the prefetched words
    pe.bias = nullptr;

    for (int j = 0; j < nt; ++j) {
        const int jt = jt0 + j, st = j & 1;
        const bool more = j + 1 < nt;
        // slab j has landed and is visible, and every wave is done with slab j-1's stage (wait + barrier of the previous iteration / the
        // prologue): the next tile's epilogue operands and the next slab are requested now, in flight under this tile's MFMAs
        if (more) { prefetch_epi(jt + 1, nxt); issue_slab(rs_w, (jt + 1) * BN, p.N, st ^ 1); }
        f32x4 acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        const unsigned char* ws = smem + st * STAGE;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int slot = ((kk * 4 + lg) ^ (li & 7)) << 4;
                bf16x8 wf[2];
#pragma unroll
                for (int a = 0; a < 2; ++a) wf[a] = *reinterpret_cast<const bf16x8*>(ws + kb * 8192 + (wn * 32 + a * 16 + li) * 128 + slot);
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[a], xf[kb][kk][b], acc[a][b], 0, 0, 0);
            }
        // ---- C/D layout (a lane holds 4 consecutive columns of one row) -> wave-private transpose tile
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int row = b * 16 + li, slot = (a * 4 + lg) ^ (row & 7);
                *reinterpret_cast<f32x4*>(etile + row * 128 + slot * 16) = acc[a][b];
            }
        if (more) {
            // one wait per tile, placed BEFORE this tile's stores are issued (gfx950 has one counter for loads and stores: behind them it
            // would also wait for the stores): the next slab and the next tile's epilogue operands have landed (this thread's), and the
            // previous tile's stores -- a whole MFMA phase old -- are out
            rt_wait_vmcnt<0>();
            __syncthreads();                                   // ... everyone's pieces; every wave is done reading slab j
        }
        // ---- tile jt's epilogue: 16-B row pieces (same wave wrote the tile: program order + the compiler's lgkmcnt, no barrier)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = er + 16 * i, s0 = ec >> 2;         // fp32 slots s0, s0 + 1 of the row
            f32x4 lo4 = *reinterpret_cast<const f32x4*>(etile + row * 128 + ((s0 ^ (row & 7)) << 4));
            f32x4 hi4 = *reinterpret_cast<const f32x4*>(etile + row * 128 + (((s0 + 1) ^ (row & 7)) << 4));
            lo4 += cur.b0; hi4 += cur.b1;
            const int m = m0 + wm * 32 + row, n = jt * BN + wn * 32 + ec;
            if (m < p.M && n < p.N)
                epilogue8<true>(pe, m, n, f32x8{lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]}, cur.res[i], cur.gate[i]);
        }
        if (more) cur = nxt;
    }
}

template <int KB>
int launch_astat(const GemmArgs& a, hipStream_t s) {
    const int m_tiles = (a.M + 63) / 64, n_tiles = (a.N + 63) / 64;
    // one round of <= 2 workgroups per CU: split the column range until the grid has ~384-512 workgroups, >= 4 tiles per workgroup
    int splits = 1;
    while (m_tiles * splits * 2 <= 512 && (n_tiles + splits * 2 - 1) / (splits * 2) >= 4) splits *= 2;
    const int tps = (n_tiles + splits - 1) / splits;
    const size_t smem = (size_t)2 * 64 * KB * 128 + 4 * 4096;
    if (smem > 65536) (void)hipFuncSetAttribute((const void*)gemm_astat_kernel<KB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL((gemm_astat_kernel<KB>), dim3((unsigned)(m_tiles * splits)), dim3(256), smem, s, a.src, a.wgt, a, m_tiles, tps);
    RT_CHECK_LAUNCH();
    return RT_OK;
}

}  // namespace

// dense rows, K = 128 / 256, N a multiple of 8 (16-B row pieces), bf16 residual / gate operands prefetched (the other epilogue operands
// are read in place)
bool rt_gemm_astat_ok(const GemmArgs& a) {
    const bool dense = (a.KH == 1 && a.KW == 1 && a.stride == 1 && a.pad == 0 && a.SH == a.DH && a.SW == a.DW);
    return dense && (a.K == 128 || a.K == 256) && (a.N & 7) == 0;
}
int rt_launch_gemm_astat(const GemmArgs& a, hipStream_t s) {
    if (!rt_gemm_astat_ok(a)) return RT_ERR_UNSUPPORTED;
    return a.K == 128 ? launch_astat<2>(a, s) : launch_astat<4>(a, s);
}
#else
bool rt_gemm_astat_ok(const GemmArgs&) { return false; }
int rt_launch_gemm_astat(const GemmArgs&, hipStream_t) { return RT_ERR_BADARG; }
#endif
