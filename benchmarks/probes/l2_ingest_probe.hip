// How fast can ONE compute unit pull L2-resident bytes, and does the path matter?
// Every workgroup re-reads its own `region` KB (larger than the 32-KB L1, all regions of an XCD together smaller than its 4-MB L2) for
// `passes` passes, through
//   mode 0: global_load_dwordx4 into VGPRs (U loads in flight per thread, xor-folded so that nothing is optimised away),
//   mode 1: buffer_load_dwordx4 ... lds (LDS-DMA, the GEMM kernels' staging path) into a ring of S slots of 16 KB, waited with counted vmcnt,
//   mode 2: like 0, then ds_write_b128 into LDS (the register-staged path).
// Prints GB/s per compute unit and TB/s for the chip at 1 ... 4 workgroups per CU and 256 / 512 threads.
//   hipcc --offload-arch=gfx950 -O3 l2_ingest_probe.hip -o l2_ingest_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__device__ __forceinline__ i32x4 make_rsrc(const void* ptr, unsigned bytes) {
    const uint64_t a = (uint64_t)ptr;
    return i32x4{(int)(uint32_t)a, (int)(uint32_t)(a >> 32), (int)bytes, 0x00020000};
}
__device__ __forceinline__ void dma16(const i32x4 rsrc, unsigned lds_base, int voff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_base), "v"(voff), "s"(rsrc) : "memory", "m0");
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// mode 0 / 2: each thread walks the region in 16-B pieces, NT * 16 bytes per step, U steps in flight
template <int NT, int U, int TO_LDS>
__global__ __launch_bounds__(NT) void vgpr_kernel(const unsigned char* __restrict__ base, int region_bytes, int passes, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned char* reg = base + (size_t)blockIdx.x * region_bytes;
    const int steps = region_bytes / (NT * 16);
    u32x4 acc = {0, 0, 0, 0};
    for (int p = 0; p < passes; ++p) {
        for (int s = 0; s < steps; s += U) {
            u32x4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                int st = s + u; if (st >= steps) st -= steps * (st / steps);
                v[u] = *reinterpret_cast<const u32x4*>(reg + (size_t)st * NT * 16 + threadIdx.x * 16);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (TO_LDS) *reinterpret_cast<u32x4*>(smem + ((u & 3) * NT + threadIdx.x) * 16) = v[u];
                else acc ^= v[u];
            }
        }
        if (TO_LDS) { __syncthreads(); acc ^= *reinterpret_cast<u32x4*>(smem + threadIdx.x * 16); __syncthreads(); }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

// mode 1: LDS-DMA into a ring of S slots (slot = NT * 16 bytes); before re-using a slot the thread waits until at most S - 1 of its own
// DMAs are outstanding
template <int NT, int S>
__global__ __launch_bounds__(NT) void dma_kernel(const unsigned char* __restrict__ base, int region_bytes, int passes, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned char* reg = base + (size_t)blockIdx.x * region_bytes;
    const i32x4 rsrc = make_rsrc(reg, (unsigned)region_bytes);
    const int steps = region_bytes / (NT * 16);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    typedef __attribute__((address_space(3))) unsigned char* lds_ptr;
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr)smem;
    int slot = 0;
    for (int p = 0; p < passes; ++p) {
        for (int s = 0; s < steps; ++s) {
            dma16(rsrc, (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + (unsigned)((slot * NT + wave * 64) * 16))), (s * NT + wave * 64 + lane) * 16);
            slot = (slot + 1 == S) ? 0 : slot + 1;
            wait_vmcnt<S - 1>();
        }
    }
    wait_vmcnt<0>();
    __syncthreads();
    if (*reinterpret_cast<unsigned*>(smem + threadIdx.x * 4) == 0x12345678u) sink[0] = 1;
}

// mode 3: the GEMM kernels' operand pattern -- a tile of R rows at a row pitch of `pitch` bytes, walked along K in 128-byte steps: a wave
// instruction covers 8 rows x 128 B (8 lanes x 16 B per row), a step R rows x 128 B.  Same bytes per pass as the linear walk when R * pitch is
// the region; what changes is which L2 channels the simultaneously issued lines fall into.
template <int NT, int S>
__global__ __launch_bounds__(NT) void dma_rows_kernel(const unsigned char* __restrict__ base, int R, int pitch, int kbytes, size_t wg_stride, int passes,
                                                      unsigned* sink, int seg = 128) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned char* reg = base + (size_t)blockIdx.x * wg_stride;
    const i32x4 rsrc = make_rsrc(reg, (unsigned)(R * pitch));
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    typedef __attribute__((address_space(3))) unsigned char* lds_ptr;
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr)smem;
    const int lpr = seg >> 4;                               // lanes per row segment (seg bytes of a row per instruction)
    const int rpw = 64 / lpr;                               // rows one wave instruction covers
    const int rows_per_pass = (NT / 64) * rpw;              // rows one instruction of every wave covers together
    int slot = 0;
    for (int p = 0; p < passes; ++p)
        for (int k = 0; k < kbytes; k += seg)
            for (int r0 = 0; r0 < R; r0 += rows_per_pass) {
                const int row = r0 + wave * rpw + lane / lpr;
                dma16(rsrc, (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + (unsigned)((slot * NT + wave * 64) * 16))), row * pitch + k + (lane % lpr) * 16);
                slot = (slot + 1 == S) ? 0 : slot + 1;
                wait_vmcnt<S - 1>();
            }
    wait_vmcnt<0>();
    __syncthreads();
    if (*reinterpret_cast<unsigned*>(smem + threadIdx.x * 4) == 0x12345678u) sink[0] = 1;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <typename F> static float time_us(F launch) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    launch(); CK(hipDeviceSynchronize());                    // warm: fills the L2s
    CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms * 1e3f;
}

int main() {
    const int CUS = 256;
    unsigned* sink; CK(hipMalloc(&sink, 4));
    unsigned char* buf; const size_t cap = (size_t)4 * CUS * 128 * 1024; CK(hipMalloc(&buf, cap)); CK(hipMemset(buf, 1, cap));
    const int passes = 64;
    printf("%-34s %6s %9s %12s %10s\n", "path", "wg/CU", "region KB", "GB/s per CU", "chip TB/s");
    for (int wpc = 1; wpc <= 4; wpc *= 2) {
        const int region = (64 / wpc) * 1024;                 // per CU 64 KB (twice its L1), per XCD 2 MB of its 4-MB L2
        const int grid = CUS * wpc;
        const double bytes = (double)grid * region * passes;
        auto report = [&](const char* name, float us) {
            printf("%-34s %6d %9d %12.1f %10.2f\n", name, wpc, region / 1024, bytes / CUS / us * 1e-3, bytes / us * 1e-6);
        };
        report("global_load x4 -> VGPR, 256 thr, U=4", time_us([&] { hipLaunchKernelGGL((vgpr_kernel<256, 4, 0>), dim3(grid), dim3(256), 0, 0, buf, region, passes, sink); }));
        report("global_load x4 -> VGPR, 256 thr, U=8", time_us([&] { hipLaunchKernelGGL((vgpr_kernel<256, 8, 0>), dim3(grid), dim3(256), 0, 0, buf, region, passes, sink); }));
        report("global_load x4 -> VGPR, 512 thr, U=8", time_us([&] { hipLaunchKernelGGL((vgpr_kernel<512, 8, 0>), dim3(grid), dim3(512), 0, 0, buf, region, passes, sink); }));
        report("global_load x4 -> ds_write, 256, U=8", time_us([&] { hipLaunchKernelGGL((vgpr_kernel<256, 8, 1>), dim3(grid), dim3(256), 4 * 256 * 16, 0, buf, region, passes, sink); }));
        report("LDS-DMA x4, 256 thr, ring 4", time_us([&] { hipLaunchKernelGGL((dma_kernel<256, 4>), dim3(grid), dim3(256), 4 * 256 * 16, 0, buf, region, passes, sink); }));
        report("LDS-DMA x4, 256 thr, ring 8", time_us([&] { hipLaunchKernelGGL((dma_kernel<256, 8>), dim3(grid), dim3(256), 8 * 256 * 16, 0, buf, region, passes, sink); }));
        report("LDS-DMA x4, 512 thr, ring 4", time_us([&] { hipLaunchKernelGGL((dma_kernel<512, 4>), dim3(grid), dim3(512), 4 * 512 * 16, 0, buf, region, passes, sink); }));
        report("LDS-DMA x4, 512 thr, ring 8", time_us([&] { hipLaunchKernelGGL((dma_kernel<512, 8>), dim3(grid), dim3(512), 8 * 512 * 16, 0, buf, region, passes, sink); }));
    }
    // the same with ONE workgroup on the whole chip (a single CU alone)
    {
        const int region = 64 * 1024; const double bytes = (double)region * passes;
        float us = time_us([&] { hipLaunchKernelGGL((vgpr_kernel<256, 8, 0>), dim3(1), dim3(256), 0, 0, buf, region, passes, sink); });
        printf("%-34s %6s %9d %12.1f\n", "one CU alone: global_load, U=8", "-", 64, bytes / us * 1e-3);
        us = time_us([&] { hipLaunchKernelGGL((dma_kernel<256, 8>), dim3(1), dim3(256), 8 * 256 * 16, 0, buf, region, passes, sink); });
        printf("%-34s %6s %9d %12.1f\n", "one CU alone: LDS-DMA, ring 8", "-", 64, bytes / us * 1e-3);
        us = time_us([&] { hipLaunchKernelGGL((dma_kernel<512, 8>), dim3(1), dim3(512), 8 * 512 * 16, 0, buf, region, passes, sink); });
        printf("%-34s %6s %9d %12.1f\n", "one CU alone: LDS-DMA 512, ring 8", "-", 64, bytes / us * 1e-3);
    }
    // GEMM-shaped walks: 256 workgroups (1 per CU), 32 rows each, K = 2048 bytes per row (64 KB per workgroup, L2-resident)
    printf("\n%-44s %12s %10s\n", "row-tile walk (LDS-DMA, 256 thr, ring 8)", "GB/s per CU", "chip TB/s");
    {
        const int R = 32, kb = 2048;
        const int pitches[] = {2048, 2048 + 128, 2048 + 256, 4096, 4096 + 128, 8192, 8192 + 128, 1536, 512};
        for (int pi = 0; pi < 9; ++pi) {
            const int pitch = pitches[pi];
            const int kbytes = pitch < kb ? pitch & ~127 : kb;
            const size_t wgs = (size_t)R * pitch;
            if (wgs * 256 > cap) continue;
            const double bytes = 256.0 * R * kbytes * passes;
            const float us = time_us([&] { hipLaunchKernelGGL((dma_rows_kernel<256, 8>), dim3(256), dim3(256), 8 * 256 * 16, 0, buf, R, pitch, kbytes, wgs, passes, sink); });
            char name[64]; snprintf(name, 64, "pitch %5d B, %4d B of each row", pitch, kbytes);
            printf("%-44s %12.1f %10.2f\n", name, bytes / 256 / us * 1e-3, bytes / us * 1e-6);
        }
        // how many bytes of a row one instruction takes: 128 (8 rows per instruction, the GEMM kernels' K tile of 64 bf16) ... 1024 (one row)
        for (int seg = 128; seg <= 1024; seg *= 2) {
            const int pitch = 4096;
            const double bytes = 256.0 * R * kb * passes;
            const float us = time_us([&] { hipLaunchKernelGGL((dma_rows_kernel<256, 8>), dim3(256), dim3(256), 8 * 256 * 16, 0, buf, R, pitch, kb, (size_t)R * pitch, passes, sink, seg); });
            char name[64]; snprintf(name, 64, "pitch  4096 B, %4d B of a row per instr", seg);
            printf("%-44s %12.1f %10.2f\n", name, bytes / 256 / us * 1e-3, bytes / us * 1e-6);
            const float us2 = time_us([&] { hipLaunchKernelGGL((dma_rows_kernel<512, 8>), dim3(256), dim3(512), 8 * 512 * 16, 0, buf, R, pitch, kb, (size_t)R * pitch, passes, sink, seg); });
            snprintf(name, 64, "   same, 512 threads");
            printf("%-44s %12.1f %10.2f\n", name, bytes / 256 / us2 * 1e-3, bytes / us2 * 1e-6);
        }
        // the same rows, but every workgroup of a run of 4 reads the SAME tile (an operand shared by the n tiles of a GEMM)
        for (int pi = 0; pi < 2; ++pi) {
            const int pitch = pi ? 4096 + 128 : 4096;
            const double bytes = 256.0 * R * kb * passes;
            const float us = time_us([&] { hipLaunchKernelGGL((dma_rows_kernel<256, 8>), dim3(256), dim3(256), 8 * 256 * 16, 0, buf, R, pitch, kb, (size_t)0, passes, sink); });
            char name[64]; snprintf(name, 64, "pitch %5d B, ALL workgroups one tile", pitch);
            printf("%-44s %12.1f %10.2f\n", name, bytes / 256 / us * 1e-3, bytes / us * 1e-6);
        }
    }
    // cold: every workgroup streams 8 MB of its own once (HBM / Infinity Cache, nothing re-read)
    {
        unsigned char* big; const size_t per = (size_t)8 << 20; CK(hipMalloc(&big, per * 256)); CK(hipMemset(big, 1, per * 256));
        const float us = time_us([&] { hipLaunchKernelGGL((dma_kernel<256, 8>), dim3(256), dim3(256), 8 * 256 * 16, 0, big, (int)per, 1, sink); });
        printf("%-44s %12.1f %10.2f\n", "cold stream, 8 MB per workgroup, LDS-DMA", (double)per / us * 1e-3, (double)per * 256 / us * 1e-6);
    }
    return 0;
}
