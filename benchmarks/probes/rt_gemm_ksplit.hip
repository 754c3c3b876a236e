// rt_conv_gemm, K-split variant (see rt_gemm_common.h / rt_gemm.hip for the family).
#include "rt_gemm_common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
__device__ __forceinline__ i32x4 ks_make_rsrc(const void* ptr, unsigned bytes) {
    const uint64_t a = (uint64_t)ptr;
    return i32x4{(int)(uint32_t)a, (int)(uint32_t)(a >> 32), (int)bytes, 0x00020000};   // stride 0, raw buffer, bounds-checked
}
// 16-B-per-lane buffer load hidden from the compiler's waitcnt bookkeeping (see the kernel): dst is valid only after the
// caller's own s_waitcnt; an out-of-range voff returns zeros
__device__ __forceinline__ void ks_load16(u32x4& dst, const i32x4 rsrc, int voff, int soff) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(dst) : "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}

// ------------------------------------------------------------------------------------------------
// K-split path for dense products with a LONG reduction and FEW output tiles (the BERT Linears at M = B*L rows, the encoder's
// linear2 and its backward-data, layer3/4's 1x1 reductions).  On those the LDS-staged kernels are latency-bound: a lone
// workgroup per CU walks barrier -> fragment reads -> MFMAs -> barrier at ~0.33 us per 64-wide K step whatever is in flight
// (profiles/r02_tile_sweep_8wave.txt).  Here there is NO LDS in the K loop and NO barrier: the four waves of a workgroup
// split K four ways, each streams its own slice of both operands straight from L2 into MFMA fragments (buffer loads, D k-steps
// in flight in rotating registers, compiler-counted vmcnt) and accumulates the full 64 x 64 tile; the four partial tiles meet
// once, in LDS, in front of the row-coalesced epilogue.  Same bytes from L2 as the 64 x 64 LDS tile (every operand element is
// fetched by exactly one wave), no ds_write / ds_read traffic, no dependent chain per K step.
template <int D>
__global__ __launch_bounds__(256, 2) void ksplit_gemm_kernel(const bf16_t* __restrict__ src, const bf16_t* __restrict__ wgt,
                                                             const GemmArgs p) {
    constexpr int EP_LD = 68;                                      // padded fp32 row of the 64-wide partial tiles
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* red = reinterpret_cast<float*>(smem);                   // [4 waves][64 rows][EP_LD]
    const int t = threadIdx.x;
    const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int li = lane & 15, lg = lane >> 4;

    const int n_tiles = (p.N + 63) / 64, m_tiles = (p.M + 63) / 64;
    const int bid = rt_xcd_remap((int)blockIdx.x, (int)gridDim.x, p.xcd);
    int tile_n = bid % n_tiles, tile_m = bid / n_tiles;
    if (p.mfast) { tile_m = bid % m_tiles; tile_n = bid / m_tiles; }
    const int n0 = tile_n * 64, m0 = tile_m * 64;

    constexpr int OOB = 0x7fffffff;
    const i32x4 rs_w = ks_make_rsrc(wgt, p.wgt_bytes), rs_x = ks_make_rsrc(src, p.src_bytes);
    int a_off[4], b_off[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + j * 16 + li, m = m0 + j * 16 + li;
        a_off[j] = n < p.N ? (n * p.K + lg * 8) * 2 : OOB;          // lane (li, lg): row li of the fragment, k = 8 * lg .. + 8
        b_off[j] = m < p.M ? (m * p.SC + lg * 8) * 2 : OOB;
    }
    const int steps = p.K >> 5;                                    // 32-wide k-steps
    const int per = (steps + 3) >> 2;
    const int ks0 = wave * per, ks1 = min(ks0 + per, steps);

    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    // The operand loads are inline asm with hand-counted waits: hipcc's waitcnt pass puts ONE vmcnt(0) at the top of the
    // rotating-stage loop (all D stages drain together once per iteration), which is exactly the stall this kernel exists to
    // avoid.  Rules kept (cdna_hip_programming.md 5.7): every destination is an "=v" output of its load statement and a "+v"
    // operand of the wait statement in front of its first use, so no consumer is scheduled above the wait; past this wave's K
    // slice the loads become out-of-range requests (zeros, no traffic), so every stage is always loaded, waited for with the
    // same count 8 * (D - 1), and multiplied; the queue is drained before the epilogue's compiler-counted loads.
    u32x4 fa[D][4], fb[D][4];
    auto load = [&](int d, int ks) __attribute__((always_inline)) {
        const bool in = ks < ks1;
        const int kb = ks << 6;                                    // byte offset of the k-step inside a row (wave-uniform)
#pragma unroll
        for (int j = 0; j < 4; ++j) ks_load16(fa[d][j], rs_w, in ? a_off[j] : OOB, kb);
#pragma unroll
        for (int j = 0; j < 4; ++j) ks_load16(fb[d][j], rs_x, in ? b_off[j] : OOB, kb);
    };
    auto wait = [&](int d) __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(%8)"
                     : "+v"(fa[d][0]), "+v"(fa[d][1]), "+v"(fa[d][2]), "+v"(fa[d][3]), "+v"(fb[d][0]), "+v"(fb[d][1]), "+v"(fb[d][2]), "+v"(fb[d][3])
                     : "n"(8 * (D - 1)) : "memory");
    };
    auto mma = [&](int d) __attribute__((always_inline)) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&fa[d][a]), *reinterpret_cast<bf16x8*>(&fb[d][b]),
                                                                  acc[a][b], 0, 0, 0);
    };
#pragma unroll
    for (int d = 0; d < D; ++d) load(d, ks0 + d);
    for (int ks = ks0; ks < ks1; ks += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            wait(d);                              // stage d has landed; the D - 1 younger stages stay in flight
            __builtin_amdgcn_sched_barrier(0);
            mma(d);
            __builtin_amdgcn_sched_barrier(0);    // the MFMAs read the stage before its registers are re-targeted
            load(d, ks + D + d);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the over-issued tail: nothing of ours stays in flight past here

    // the four K-partials of the tile meet in LDS; then every thread owns 16-B pieces of ONE output row (epilogue8)
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
            *reinterpret_cast<f32x4*>(red + ((size_t)(wave * 64 + b * 16 + li)) * EP_LD + a * 16 + lg * 4) = acc[a][b];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = i * 256 + t;
        const int rl = idx >> 3, cl = (idx & 7) * 8;
        const int m = m0 + rl, n = n0 + cl;
        if (m >= p.M || n >= p.N) continue;
        f32x4 lo = f32x4{0.f, 0.f, 0.f, 0.f}, hi = lo;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            lo += *reinterpret_cast<const f32x4*>(red + ((size_t)(w * 64 + rl)) * EP_LD + cl);
            hi += *reinterpret_cast<const f32x4*>(red + ((size_t)(w * 64 + rl)) * EP_LD + cl + 4);
        }
        epilogue8(p, m, n, f32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
    }
}

template <int D>
static int launch_gemm_ksplit(const GemmArgs& a, hipStream_t s) {
    const int mt = (a.M + 63) / 64, nt = (a.N + 63) / 64;
    constexpr size_t smem = (size_t)4 * 64 * 68 * 4;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)ksplit_gemm_kernel<D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        attr_set = true;
    }
    GemmArgs am = a;
    const double R = (double)(mt * nt) / 8.0;
    const double cn = R / nt + (R < nt ? R : nt), cm = R / mt + (R < mt ? R : mt);
    am.mfast = (a.xcd && mt * nt >= 16 && cm < cn) ? 1 : 0;
    hipLaunchKernelGGL((ksplit_gemm_kernel<D>), dim3((unsigned)(mt * nt)), dim3(256), smem, s, a.src, a.wgt, am);
    RT_CHECK_LAUNCH();
    return RT_OK;
}

}  // namespace

int rt_launch_gemm_ksplit(const GemmArgs& a, int depth, hipStream_t s) {
    switch (depth) {
        case 2: return launch_gemm_ksplit<2>(a, s);
        case 3: return launch_gemm_ksplit<3>(a, s);
        default: return launch_gemm_ksplit<4>(a, s);
    }
}
