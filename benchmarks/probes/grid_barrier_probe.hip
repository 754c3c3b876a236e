// What does ONE stage of a cooperative (multi-workgroup, persistent) decoder kernel cost on this chip?
// A stage = every workgroup reads the full 8 x 256 fp32 row block its predecessors wrote, does its slice of the next Linear
// (weights prefetched into registers BEFORE the barrier wait), writes its slice, grid barrier.  G workgroups take part:
//   placement 0: G consecutive workgroups (round-robin over the 8 XCDs -> hand-offs cross XCD L2s),
//   placement 1: grid = 8 G, only blockIdx % 8 == 0 takes part (all on ONE XCD: one L2 is the coherence point).
//   sync 0: agent-scope release / acquire atomics as the compiler emits them (buffer_wbl2 sc1 / buffer_inv sc1),
//   sync 1: hand-off data through sc0 sc1 (write-through / cache-bypassing) loads and stores, relaxed atomics, no L2 writeback or
//           invalidate.
// Prints microseconds per stage.   hipcc --offload-arch=gfx950 -O3 grid_barrier_probe.hip -o grid_barrier_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f4 ld_bypass(const float* p) {
    f4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void st_through(float* p, f4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}

template <int SYNC, int WLOADS>
__global__ __launch_bounds__(256) void stage_kernel(float* buf, unsigned* cnt, const f4* __restrict__ w, int G, int stages, int confine,
                                                    float* sink) {
    int p;
    if (confine) {
        if ((blockIdx.x & 7) != 0) return;
        p = blockIdx.x >> 3;
    } else {
        p = blockIdx.x;
    }
    const int t = threadIdx.x;
    const int slice = 2048 / G;                 // floats of the row block this workgroup writes per stage
    float acc = 0.f;
    const int wave = t >> 6;
    constexpr int WPT = (WLOADS * 256 + 191) / 192;          // f4 per thread of waves 1-3 (wave 0 owns the barrier: its vmcnt queue
                                                             // must not hold weight loads in front of the counter reads)
    f4 wreg[WPT > 0 ? WPT : 1];
    auto wload = [&](int s) {
        if (wave > 0) {
#pragma unroll
            for (int j = 0; j < WPT; ++j)
                wreg[j] = w[((size_t)s * G + p) * (WLOADS * 256) + (j * 192 + (t - 64)) % (WLOADS * 256)];
        }
    };
    if (WLOADS > 0) wload(0);
    for (int s = 0; s < stages; ++s) {
        float* in = buf + (s & 1) * 2048;
        float* out = buf + ((s + 1) & 1) * 2048;
        // ---- read the full row block
        f4 a, b;
        if (SYNC == 1) {
            a = ld_bypass(in + t * 4);
            b = ld_bypass(in + 1024 + t * 4);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            a = *(const f4*)(in + t * 4);
            b = *(const f4*)(in + 1024 + t * 4);
        }
        float v = a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w;
        if (WLOADS > 0 && wave > 0) {
#pragma unroll
            for (int j = 0; j < WPT; ++j) v += wreg[j].x * 1e-9f + wreg[j].w * 1e-9f;
        }
        acc += v;
        // ---- write own slice
        if (t * 4 < slice) {
            f4 o = {v, v * 0.5f, v * 0.25f, 1.f};
            if (SYNC == 1) st_through(out + p * slice + t * 4, o);
            else *(f4*)(out + p * slice + t * 4) = o;
        }
        if (SYNC == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // ---- the NEXT stage's weights are requested before the barrier wait (they do not depend on activations)
        if (WLOADS > 0 && s + 1 < stages) wload(s + 1);
        // ---- grid barrier
        __syncthreads();
        if (t == 0) {
            const unsigned target = (unsigned)(s + 1) * G;
            int guard = 0;
            if (SYNC == 0) {
                __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                while (__hip_atomic_load(cnt, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target && ++guard < (1 << 22)) {}
            } else {
                __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && ++guard < (1 << 22)) {}
            }
            if (guard >= (1 << 22)) sink[1] = 1.f;          // a lost participant: reported, never a hang
        }
        __syncthreads();
    }
    if (acc == 12345.678f) sink[0] = acc;
}

template <int SYNC, int WL>
static float run(float* buf, unsigned* cnt, f4* w, int G, int stages, int confine, float* sink) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        hipMemsetAsync(cnt, 0, 4, 0);
        hipMemsetAsync(buf, 0, 4096 * 4, 0);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((stage_kernel<SYNC, WL>), dim3(confine ? G * 8 : G), dim3(256), 0, 0, buf, cnt, w, G, stages, confine, sink);
        hipEventRecord(e1, 0);
        if (hipEventSynchronize(e1) != hipSuccess) { printf("launch failed\n"); exit(1); }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    float flag[2]; hipMemcpy(flag, sink, 8, hipMemcpyDeviceToHost);
    if (flag[1] != 0.f) { printf("barrier timed out (G %d confine %d)\n", G, confine); hipMemset(sink, 0, 8); return -1.f; }
    return best * 1000.f / stages;
}

int main() {
    float* buf; unsigned* cnt; f4* w; float* sink;
    hipMalloc(&buf, 4096 * 4); hipMalloc(&cnt, 64); hipMalloc(&sink, 64); hipMemset(sink, 0, 64);
    const int stages = 400;
    const size_t wbytes = (size_t)stages * 256 * 4 * 256 * 16;       // stages x G(<=256) x WLOADS(<=4) x 256 threads x 16 B
    hipMalloc(&w, wbytes); hipMemset(w, 0, wbytes);
    printf("us per stage (8 x 256 fp32 row block hand-off + grid barrier), %d stages per launch\n", stages);
    printf("%4s %8s %6s | %10s %10s %10s\n", "G", "place", "sync", "no weights", "4 KB/wg", "16 KB/wg");
    const int Gs[] = {8, 16, 32, 64, 128, 256};
    for (int confine = 0; confine < 2; ++confine)
        for (int G : Gs) {
            if (confine && G > 32) continue;         // one XCD = 32 CUs
            for (int sync = 0; sync < 2; ++sync) {
                float a, b, c;
                if (sync == 0) {
                    a = run<0, 0>(buf, cnt, w, G, stages, confine, sink);
                    b = run<0, 1>(buf, cnt, w, G, stages, confine, sink);
                    c = run<0, 4>(buf, cnt, w, G, stages, confine, sink);
                } else {
                    a = run<1, 0>(buf, cnt, w, G, stages, confine, sink);
                    b = run<1, 1>(buf, cnt, w, G, stages, confine, sink);
                    c = run<1, 4>(buf, cnt, w, G, stages, confine, sink);
                }
                printf("%4d %8s %6s | %10.2f %10.2f %10.2f\n", G, confine ? "one XCD" : "spread", sync ? "sc0sc1" : "rel/acq", a, b, c);
                fflush(stdout);
            }
        }
    return 0;
}
