// rt_bottleneck_fwd -- one FROZEN stride-1 ResNet bottleneck of layer1 as ONE launch (round 4):
//   conv1 1x1 cin->64 + FrozenBN + ReLU  ->  conv2 3x3 64->64 + FrozenBN + ReLU  ->  conv3 1x1 64->256 + FrozenBN
//   (+ identity | + downsample 1x1 cin->256 + FrozenBN)  -> ReLU
// (models/modeling/backbone.py:87-89 freezes conv1 / layer1; torchvision Bottleneck, v1.5).  Nothing in a frozen block is
// needed again (no weight gradient reads h1 / h2), so the three launches' round trips -- 419 MB per block at 8 x 160 x 160,
// where block input + block output are 210 MB -- are pure overhead: here a workgroup owns an 8 x 16 pixel tile of the OUTPUT,
// recomputes conv1 on the 10 x 18 haloed tile, and h1 / h2 live only in LDS.
//
// Layout of a workgroup (256 threads = 4 waves, 80 KB LDS, two workgroups per CU):
//   stage 1  h1[192(180 used) x 64] = X_halo[192 x cin] . W1^T     X streamed in 64-channel chunks (24 KB, two slots, LDS-DMA),
//                                                                   wave w owns haloed rows 48w .. 48w+47
//   stage 2  h2[128 x 64] = sum over the 9 taps of h1[shifted rows] . W2[tap]^T      wave w owns output rows 2w, 2w+1
//   stage 3  out[128 x 256] = h2 . W3^T (+ X_centre . Wd^T) in four 64-channel quarters, epilogue in registers
// Every weight operand travels as an 8 KB piece (64 output features x 64 input channels) through a 4-slot LDS ring, three
// pieces ahead of its use (LDS-DMA, counted vmcnt waits: one barrier per piece).  The rows of a piece are stored in LDS in
// MFMA-tile order of a PERMUTED feature order -- LDS row a*16 + i holds feature (i>>2)*16 + a*4 + (i&3) -- so that after the
// four 16-feature MFMAs a lane holds 16 CONSECUTIVE features of its pixel: h1 / h2 go to LDS as two 16-B writes per pixel, and
// the output / residual as two 16-B global accesses per lane, 128 contiguous bytes per pixel and wave instruction, without an
// LDS staging pass.  Rounding points are those of the rt_conv_gemm launches it replaces (bf16 h1, h2, downsample output, out;
// fp32 accumulate, bias, residual, ReLU): the two paths differ by summation order only.
#include "rt_common.h"
#include <stdlib.h>

namespace {

typedef __attribute__((ext_vector_type(4))) int i32x4;

__device__ __forceinline__ i32x4 bk_rsrc(const void* ptr, unsigned bytes) {
    const uint64_t a = (uint64_t)ptr;
    return i32x4{(int)(uint32_t)a, (int)(uint32_t)(a >> 32), (int)bytes, 0x00020000};
}
// 16 B per lane, global -> LDS: lane l lands at lds_base + 16 l (lds_base wave-uniform); out-of-range offsets write zeros
__device__ __forceinline__ void bk_dma16(const i32x4 rsrc, unsigned lds_base, int voff, int soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 ::"s"(lds_base), "v"(voff), "s"(rsrc), "s"(soff)
                 : "memory", "m0");
}
// wait until at most n vector-memory operations of this wave are outstanding (n is a compile-time constant after unrolling)
__device__ __forceinline__ void bk_wait(int n) {
    switch (n) {
#define BK_W(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
        BK_W(0) BK_W(1) BK_W(2) BK_W(3) BK_W(4) BK_W(5) BK_W(6) BK_W(7) BK_W(8) BK_W(9) BK_W(10) BK_W(11) BK_W(12)
        BK_W(13) BK_W(14) BK_W(15) BK_W(16) BK_W(17) BK_W(18) BK_W(19) BK_W(20) BK_W(21) BK_W(22) BK_W(23) BK_W(24)
#undef BK_W
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

struct BnkArgs {
    const bf16_t *x, *w1, *w2, *w3, *wd;
    const float *b1, *b2, *b3, *bd;
    bf16_t* out;
    int B, H, W, tiles_x, tiles_y;
    unsigned x_bytes;
};

constexpr int BK_TH = 8, BK_TW = 16, BK_HW = BK_TW + 2, BK_HALO = (BK_TH + 2) * BK_HW, BK_XROWS = 192;
constexpr int BK_XSLOT = BK_XROWS * 128, BK_WSLOT = 64 * 128, BK_LDS = 2 * BK_XSLOT + 4 * BK_WSLOT;
constexpr int BK_LX = BK_XROWS / 32, BK_LW = 2;            // DMA instructions per thread: one X chunk, one weight piece

// TW / NW / R: tile width in pixels, waves, ring slots.  (16, 4, 4) is the only instantiation: 8 x 16 tile, 80 KB of LDS, two
// workgroups per CU.  (Round 4 also measured (32, 8, 8) -- one 8-wave workgroup per CU on an 8 x 32 tile -- and a persistent form with
// the weights in registers: both bit-identical and slower, removed in round 5; profiles/r04ac_bottleneck_bench_three_forms.txt.)
template <int CIN, bool DOWN, int TW = 16, int NW = 4, int R = 4>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void bottleneck_fwd_kernel(const BnkArgs p) {
    static_assert((R & (R - 1)) == 0 && R >= 4 && TW % 16 == 0 && NW * 2 == BK_TH * TW / 16, "8 rows x TW / 16 pixel tiles, two per wave");
    constexpr int HWD = TW + 2, HALO = (BK_TH + 2) * HWD, XROWS = NW * 48, XSLOT = XROWS * 128, RPP = NW * 8, LX = XROWS / RPP, LW = 64 / RPP,
                  PASS = RPP * 128, CT = TW / 16;
    static_assert(HALO <= XROWS && BK_TH * TW * 128 <= XSLOT, "the haloed tile / h2 fit an X slot");
    constexpr int NK1 = CIN / 64;                         // conv1 K chunks
    constexpr int NP = NK1 + 9 + (DOWN ? 8 : 4);          // weight pieces in order of use: conv1 chunks, conv2 taps, conv3 (+downsample) quarters
    static_assert(NK1 == 1 || !DOWN, "the downsample variant keeps the whole input tile resident: cin = 64");
    constexpr unsigned T1_OFF = DOWN ? XSLOT : 0;      // h1 [192 x 64] bf16: over X slot 0 once conv1 is done (identity variant)
    constexpr unsigned T2_OFF = XSLOT;                 // h2 [128 x 64] bf16: X slot 1 (identity) / over h1 after a barrier (downsample)
    constexpr unsigned WR_OFF = 2 * XSLOT;             // weight ring, 4 x 8 KB
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef __attribute__((address_space(3))) void* lds_ptr;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 15, lg = lane >> 4;
    // vertically adjacent tiles share two haloed rows, horizontally adjacent ones two columns: contiguous runs per XCD
    const int bid = rt_xcd_remap((int)blockIdx.x, (int)gridDim.x, 1);
    const int tx = bid % p.tiles_x, tyb = bid / p.tiles_x, ty = tyb % p.tiles_y, b = tyb / p.tiles_y;
    const int oy0 = ty * BK_TH, ox0 = tx * TW;

    const int srow = t >> 3;
    const int chunk = (t & 7) ^ (srow & 7);               // source-side swizzle (see rt_gemm_dma.h)
    constexpr int OOB = 0x7fffffff;
    int x_off[LX];
#pragma unroll
    for (int j = 0; j < LX; ++j) {
        const int hp = srow + RPP * j, hr = hp / HWD, hc = hp - hr * HWD;
        const int y = oy0 - 1 + hr, x = ox0 - 1 + hc;
        const bool ok = hp < HALO && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
        x_off[j] = ok ? (((b * p.H + y) * p.W + x) * CIN + chunk * 8) * 2 : OOB;
    }
    int w1_off[LW], w2_off[LW], w3_off[LW];
#pragma unroll
    for (int j = 0; j < LW; ++j) {
        const int r = srow + RPP * j, a = r >> 4, i = r & 15;
        const int ch = (i >> 2) * 16 + a * 4 + (i & 3);  // permuted feature order
        w1_off[j] = (ch * CIN + chunk * 8) * 2;
        w2_off[j] = (ch * 576 + chunk * 8) * 2;
        w3_off[j] = (ch * 64 + chunk * 8) * 2;
    }
    const i32x4 rs_x = bk_rsrc(p.x, p.x_bytes), rs_w1 = bk_rsrc(p.w1, 64 * CIN * 2), rs_w2 = bk_rsrc(p.w2, 64 * 576 * 2),
                rs_w3 = bk_rsrc(p.w3, 256 * 64 * 2), rs_wd = bk_rsrc(DOWN ? p.wd : p.w3, 256 * 64 * 2);
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr)smem + (unsigned)__builtin_amdgcn_readfirstlane(wave) * 1024u;

    auto issue_x = [&](int kc) __attribute__((always_inline)) {
        const unsigned base = lds0 + (kc & 1) * XSLOT;
#pragma unroll
        for (int j = 0; j < LX; ++j) bk_dma16(rs_x, base + j * PASS, x_off[j], kc * 128);
    };
    auto issue_w = [&](int pc) __attribute__((always_inline)) {      // pc is a compile-time constant at every call site
        const unsigned base = lds0 + WR_OFF + (pc & (R - 1)) * BK_WSLOT;
        if (pc < NK1) {
#pragma unroll
            for (int j = 0; j < LW; ++j) bk_dma16(rs_w1, base + j * PASS, w1_off[j], pc * 128);
        } else if (pc < NK1 + 9) {
#pragma unroll
            for (int j = 0; j < LW; ++j) bk_dma16(rs_w2, base + j * PASS, w2_off[j], (pc - NK1) * 128);
        } else {
            const int i = pc - NK1 - 9, q = DOWN ? i >> 1 : i;
            const bool dn = DOWN && (i & 1);
#pragma unroll
            for (int j = 0; j < LW; ++j) bk_dma16(dn ? rs_wd : rs_w3, base + j * PASS, w3_off[j], q * 8192);
        }
    };
    // A fragments (weights) of piece pc: 4 feature tiles x 2 K halves
    auto w_frags = [&](int pc, bf16x8 (&wf)[2][4]) __attribute__((always_inline)) {
        const unsigned char* base = smem + WR_OFF + (pc & (R - 1)) * BK_WSLOT;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int a = 0; a < 4; ++a)
                wf[kk][a] = *reinterpret_cast<const bf16x8*>(base + (a * 16 + li) * 128 + (((kk * 4 + lg) ^ (li & 7)) << 4));
    };
    // 16 consecutive features of one LDS row (pixel) as two 16-B pieces
    auto put_row = [&](unsigned off, int row, const float (&v)[16]) __attribute__((always_inline)) {
        bf16x8 lo, hi;
#pragma unroll
        for (int e = 0; e < 8; ++e) { lo[e] = (bf16_t)v[e]; hi[e] = (bf16_t)v[8 + e]; }
        unsigned char* r = smem + off + row * 128;
        *reinterpret_cast<bf16x8*>(r + (((lg * 2) ^ (row & 7)) << 4)) = lo;
        *reinterpret_cast<bf16x8*>(r + (((lg * 2 + 1) ^ (row & 7)) << 4)) = hi;
    };

    // identity shortcut: the 4 x 16 features per lane and pixel the stage-3 epilogue adds are requested FIRST -- they are the oldest
    // entries of the wave's vmcnt queue (the counted waits below stay exact), and their lines are the ones the X chunks fetch anyway
    int opix[2];
    bool oin[2];
#pragma unroll
    for (int bb = 0; bb < 2; ++bb) {
        const int mt = wave * 2 + bb, oy = oy0 + mt / CT, ox = ox0 + (mt % CT) * 16 + li;
        oin[bb] = oy < p.H && ox < p.W;
        opix[bb] = (b * p.H + oy) * p.W + ox;
    }
    bf16x8 res[4][2][2];
    if (!DOWN) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) {
                const bf16_t* rp = p.x + (size_t)(oin[bb] ? opix[bb] : 0) * 256 + q * 64 + lg * 16;
                res[q][bb][0] = *reinterpret_cast<const bf16x8*>(rp);
                res[q][bb][1] = *reinterpret_cast<const bf16x8*>(rp + 8);
            }
    }

    float b1v[16], b2v[16];
#pragma unroll
    for (int e = 0; e < 16; e += 4) {
        const f32x4 u = *reinterpret_cast<const f32x4*>(p.b1 + lg * 16 + e), w = *reinterpret_cast<const f32x4*>(p.b2 + lg * 16 + e);
#pragma unroll
        for (int r = 0; r < 4; ++r) { b1v[e + r] = u[r]; b2v[e + r] = w[r]; }
    }

    // ---- prologue: pieces 0 .. 2 and the first two X chunks in flight.  `marks`: DMA instructions issued so far / when an
    // operand was issued (all compile-time after unrolling): the wait for an operand allows exactly the younger ones outstanding.
    int issued = 0, x_mark[NK1 > 1 ? NK1 : 2], w_mark[NP];
    issue_x(0); issued += LX; x_mark[0] = issued;
    issue_w(0); issued += LW; w_mark[0] = issued;
    if (NK1 > 1) { issue_x(1); issued += LX; x_mark[1] = issued; }
#pragma unroll
    for (int pc = 1; pc < R - 1; ++pc) { issue_w(pc); issued += LW; w_mark[pc] = issued; }

    // ---- stage 1: conv1 on the haloed tile
    f32x4 acc1[4][3];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int bb = 0; bb < 3; ++bb) acc1[a][bb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kc = 0; kc < NK1; ++kc) {
        const int need = x_mark[kc] > w_mark[kc] ? x_mark[kc] : w_mark[kc];
        bk_wait(issued - need);
        __syncthreads();                                 // chunk kc / piece kc visible; everyone is done with chunk kc-1 / piece kc-1
        if (kc >= 1 && kc + 1 < NK1) { issue_x(kc + 1); issued += LX; x_mark[kc + 1] = issued; }
        if (kc + R - 1 < NP) { issue_w(kc + R - 1); issued += LW; w_mark[kc + R - 1] = issued; }
        bf16x8 wf[2][4];
        w_frags(kc, wf);
        const unsigned char* xs = smem + (kc & 1) * XSLOT;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 xf[3];
#pragma unroll
            for (int bb = 0; bb < 3; ++bb)
                xf[bb] = *reinterpret_cast<const bf16x8*>(xs + ((wave * 3 + bb) * 16 + li) * 128 + (((kk * 4 + lg) ^ (li & 7)) << 4));
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int bb = 0; bb < 3; ++bb)
                    acc1[a][bb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kk][a], xf[bb], acc1[a][bb], 0, 0, 0);
        }
    }
    if (!DOWN) __syncthreads();                          // h1 goes over X slot 0: every wave is done with the last chunk held there
#pragma unroll
    for (int bb = 0; bb < 3; ++bb) {
        const int hp = (wave * 3 + bb) * 16 + li, hr = hp / HWD, hc = hp - hr * HWD;
        const int y = oy0 - 1 + hr, x = ox0 - 1 + hc;
        // conv2 pads h1 with ZEROS outside the image (not with conv1 of zeros)
        const bool in = hp < HALO && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
        float v[16];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[a * 4 + r] = in ? fmaxf(acc1[a][bb][r] + b1v[a * 4 + r], 0.f) : 0.f;
        put_row(T1_OFF, hp, v);
    }

    // ---- stage 2: conv2, nine taps over h1
    f32x4 acc2[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int bb = 0; bb < 2; ++bb) acc2[a][bb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) {
        const int pc = NK1 + tp, kh = tp / 3, kw = tp % 3;
        bk_wait(issued - w_mark[pc]);
        __syncthreads();                                 // (tp = 0: h1 complete)
        if (pc + R - 1 < NP) { issue_w(pc + R - 1); issued += LW; w_mark[pc + R - 1] = issued; }
        bf16x8 wf[2][4];
        w_frags(pc, wf);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 xf[2];
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) {
                const int mt = wave * 2 + bb, row = (mt / CT + kh) * HWD + kw + (mt % CT) * 16 + li;
                xf[bb] = *reinterpret_cast<const bf16x8*>(smem + T1_OFF + row * 128 + (((kk * 4 + lg) ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int bb = 0; bb < 2; ++bb)
                    acc2[a][bb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kk][a], xf[bb], acc2[a][bb], 0, 0, 0);
        }
    }
    if (DOWN) __syncthreads();                           // h2 goes over h1
#pragma unroll
    for (int bb = 0; bb < 2; ++bb) {
        float v[16];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[a * 4 + r] = fmaxf(acc2[a][bb][r] + b2v[a * 4 + r], 0.f);
        put_row(T2_OFF, (wave * 2 + bb) * 16 + li, v);
    }

    // ---- stage 3: conv3 (+ downsample) in four 64-feature quarters; a wave reads only the h2 rows it wrote itself
    bf16x8 hf[2][2], cf[2][2];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int pc = NK1 + 9 + (DOWN ? 2 * q : q);
        bk_wait(issued - w_mark[pc]);
        __syncthreads();                                 // (q = 0: h2 complete)
        if (pc + R - 1 < NP) { issue_w(pc + R - 1); issued += LW; w_mark[pc + R - 1] = issued; }
        f32x4 bq[4], bdq[4];                             // younger than every piece waited for so far: they can only make a wait longer
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            bq[a] = *reinterpret_cast<const f32x4*>(p.b3 + q * 64 + lg * 16 + a * 4);
            if (DOWN) bdq[a] = *reinterpret_cast<const f32x4*>(p.bd + q * 64 + lg * 16 + a * 4);
        }
        if (q == 0) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int bb = 0; bb < 2; ++bb) {
                    const int m = (wave * 2 + bb) * 16 + li;
                    hf[kk][bb] = *reinterpret_cast<const bf16x8*>(smem + T2_OFF + m * 128 + (((kk * 4 + lg) ^ (m & 7)) << 4));
                    if (DOWN) {
                        const int mt = wave * 2 + bb, hp = (mt / CT + 1) * HWD + 1 + (mt % CT) * 16 + li;
                        cf[kk][bb] = *reinterpret_cast<const bf16x8*>(smem + hp * 128 + (((kk * 4 + lg) ^ (hp & 7)) << 4));
                    }
                }
        }
        f32x4 acc3[4][2], accd[4][2];                    // the downsample branch keeps accumulators of its own: it is rounded to bf16
#pragma unroll                                           // before it joins (the identity tensor of the unfused launches)
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) { acc3[a][bb] = f32x4{0.f, 0.f, 0.f, 0.f}; accd[a][bb] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        {
            bf16x8 wf[2][4];
            w_frags(pc, wf);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int bb = 0; bb < 2; ++bb)
                        acc3[a][bb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kk][a], hf[kk][bb], acc3[a][bb], 0, 0, 0);
        }
        if (DOWN) {
            bk_wait(issued - w_mark[pc + 1]);
            __syncthreads();
            if (pc + R < NP) { issue_w(pc + R); issued += LW; w_mark[pc + R] = issued; }
            bf16x8 wf[2][4];
            w_frags(pc + 1, wf);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int bb = 0; bb < 2; ++bb)
                        accd[a][bb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kk][a], cf[kk][bb], accd[a][bb], 0, 0, 0);
        }
#pragma unroll
        for (int bb = 0; bb < 2; ++bb) {
            if (!oin[bb]) continue;
            bf16x8 lo, hi;
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int e = a * 4 + r;
                    float v = acc3[a][bb][r] + bq[a][r];
                    if (!DOWN) v += (float)(e < 8 ? res[q][bb][0][e & 7] : res[q][bb][1][e & 7]);
                    else v += (float)(bf16_t)(accd[a][bb][r] + bdq[a][r]);
                    v = fmaxf(v, 0.f);
                    if (e < 8) lo[e] = (bf16_t)v; else hi[e - 8] = (bf16_t)v;
                }
            bf16_t* op = p.out + (size_t)opix[bb] * 256 + q * 64 + lg * 16;
            *reinterpret_cast<bf16x8*>(op) = lo;
            *reinterpret_cast<bf16x8*>(op + 8) = hi;
        }
    }
}

template <int CIN, bool DOWN>
int launch_bottleneck(const BnkArgs& a, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)bottleneck_fwd_kernel<CIN, DOWN>, hipFuncAttributeMaxDynamicSharedMemorySize, BK_LDS);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const unsigned grid = (unsigned)(a.B * a.tiles_x * a.tiles_y);
    hipLaunchKernelGGL((bottleneck_fwd_kernel<CIN, DOWN>), dim3(grid), dim3(256), BK_LDS, s, a);
    RT_CHECK_LAUNCH();
    return RT_OK;
}

}  // namespace

extern "C" int rt_bottleneck_fwd(const rt_bottleneck_desc* d, rt_stream_t stream) {
    if (!d || !d->x || !d->w1 || !d->w2 || !d->w3 || !d->b1 || !d->b2 || !d->b3 || !d->out) return RT_ERR_BADARG;
    if (d->planes != 64 || d->B <= 0 || d->H <= 0 || d->W <= 0 || d->form < 0 || d->form > 1) return RT_ERR_UNSUPPORTED;      // forms 2 / 3 (round 4: persistent / 8-wave, slower) were removed
    const long long in_bytes = (long long)d->B * d->H * d->W * d->cin * 2, out_elems = (long long)d->B * d->H * d->W * 256;
    if (in_bytes >= (1ll << 31) || out_elems >= (1ll << 31)) return RT_ERR_UNSUPPORTED;      // 32-bit buffer offsets / pixel indices
    BnkArgs a;
    a.x = (const bf16_t*)d->x; a.w1 = (const bf16_t*)d->w1; a.w2 = (const bf16_t*)d->w2; a.w3 = (const bf16_t*)d->w3;
    a.wd = (const bf16_t*)d->wd; a.b1 = d->b1; a.b2 = d->b2; a.b3 = d->b3; a.bd = d->bd; a.out = (bf16_t*)d->out;
    a.B = d->B; a.H = d->H; a.W = d->W;
    a.tiles_x = (d->W + BK_TW - 1) / BK_TW; a.tiles_y = (d->H + BK_TH - 1) / BK_TH;
    a.x_bytes = (unsigned)in_bytes;
    hipStream_t s = (hipStream_t)stream;
    if (d->cin == 64 && d->wd && d->bd) return launch_bottleneck<64, true>(a, s);
    if (d->cin == 256 && !d->wd) return launch_bottleneck<256, false>(a, s);
    return RT_ERR_UNSUPPORTED;
}
