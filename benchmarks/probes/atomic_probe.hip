// How fast are fp32 global atomics when `nblk` workgroups add 128x128 fp32 tiles into `ntile` distinct output tiles?
// pattern 0: MFMA C/D layout (16 lanes x 64 B on 4 rows per instruction); pattern 1: 64 lanes x 256 B contiguous;
// pattern 2: plain stores (pattern 1 addressing).  rot: rotate the starting row per split to de-synchronise addresses.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
__global__ __launch_bounds__(256) void k(float* out, int ntile, int ld, int pattern, int rot) {
    const int tile = blockIdx.x % ntile, split = blockIdx.x / ntile;
    float* base = out + (size_t)tile * 128 * ld;       // tiles stacked along rows; row length ld >= 128
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (pattern == 0) {
        const int li = lane & 15, lg = lane >> 4, wn = wave & 1, wc = wave >> 1;
        for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) for (int r = 0; r < 4; ++r) {
            int aa = rot ? (a + split) & 3 : a;
            const int n = wn * 64 + aa * 16 + lg * 4 + r, c = wc * 64 + b * 16 + li;
            atomicAdd(base + (size_t)n * ld + c, 1.0f);
        }
    } else {
        for (int i = 0; i < 64; ++i) {
            int ii = rot ? (i + split * 5) & 63 : i;
            const int row = ii * 2 + (wave >> 1), c = (wave & 1) * 64 + lane;
            if (pattern == 1) atomicAdd(base + (size_t)row * ld + c, 1.0f);
            else base[(size_t)row * ld + c] = 1.0f;
        }
    }
}
int main(int argc, char** argv) {
    float* d; hipMalloc(&d, 64 << 20); hipMemset(d, 0, 64 << 20);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int cfgs[][2] = {{16, 25}, {4, 100}, {2, 128}, {64, 4}, {36, 16}, {256, 2}, {512, 1}};
    for (auto& c : cfgs) for (int pattern = 0; pattern < 3; ++pattern) for (int rot = 0; rot < 2; ++rot) {
        const int ntile = c[0], nsplit = c[1];
        for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k, dim3(ntile * nsplit), dim3(256), 0, 0, d, ntile, 128, pattern, rot);
        hipEventRecord(e0);
        for (int w = 0; w < 10; ++w) hipLaunchKernelGGL(k, dim3(ntile * nsplit), dim3(256), 0, 0, d, ntile, 128, pattern, rot);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 100.0, mb = (double)ntile * nsplit * 65536 / 1e6;
        printf("tiles %3d x splits %3d  pattern %d rot %d: %7.1f us  (%6.1f MB, %5.2f TB/s)\n", ntile, nsplit, pattern, rot, us, mb, mb / us / 1e6 * 1e6 / 1e6);
    }
    return 0;
}
