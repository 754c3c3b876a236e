#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __bf16 bf16_t;
__global__ void k(const uint32_t* src, uint32_t* out, unsigned nbytes) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[64 * 4 * 2];
    for (int i = threadIdx.x; i < 512; i += 64) lds[i] = 0xdeadbeef;
    __syncthreads();
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(src), 0, nbytes, 0x00020000);
    // lane l loads 16 B at byte offset: even lanes in range, odd lanes out of range
    int voff = (threadIdx.x & 1) ? 0x7fffffff : (int)threadIdx.x * 16;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, voff, 0, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);   // vmcnt(0) etc
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 64) out[i] = lds[i];
}
int main() {
    uint32_t h[1024]; for (int i = 0; i < 1024; i++) h[i] = i + 1;
    uint32_t *d, *o; hipMalloc(&d, 4096); hipMalloc(&o, 2048);
    hipMemcpy(d, h, 4096, hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, o, 4096);
    uint32_t r[512]; hipMemcpy(r, o, 2048, hipMemcpyDeviceToHost);
    for (int l = 0; l < 6; l++) printf("lane %d: %x %x %x %x\n", l, r[l*4], r[l*4+1], r[l*4+2], r[l*4+3]);
    printf("tail: %x %x\n", r[256], r[300]);
    return 0;
}
