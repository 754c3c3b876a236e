// Host input pipeline on the device: Pillow-compatible antialiased bilinear resampling of uint8 images
// (datasets/transforms.py:81-116 -> torchvision F.resize on PIL images -> Pillow ImagingResample, 8-bit path) and
// ToTensor + Normalize + batch padding (transforms.py:233-263, util/collate_fn.py:24-41) in one pass.
// Integer / byte work, HBM-bound: bit-exact by construction -- the 22-bit fixed-point filter taps are computed on the host
// in double precision exactly as Pillow does (reftr_amd/data/resample.py) and the kernels only multiply-accumulate int32.
#include "rt_common.h"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;

__device__ __forceinline__ uint8_t clip8(int v) {
    v >>= PRECISION_BITS;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// One resampling pass along `axis` (0 = rows / vertical, 1 = columns / horizontal) of src[n0][n1][C] (uint8):
// out[i][j][c] = clip8(2^21 + sum_t src[..][lo + t][..] * k[t]); bounds[o] = {lo, n taps}, coeffs[o][ksize].
__global__ __launch_bounds__(256) void resample_u8_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                          const int* __restrict__ bounds, const int* __restrict__ coeffs,
                                                          int n0, int n1, int C, int out_len, int ksize, int axis) {
    const int o0 = axis == 0 ? out_len : n0, o1 = axis == 1 ? out_len : n1;
    const size_t total = (size_t)o0 * o1 * C;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C); const size_t t = i / C;
        const int x = (int)(t % o1), y = (int)(t / o1);
        const int o = axis == 0 ? y : x;
        const int lo = bounds[2 * o], n = bounds[2 * o + 1];
        const int* k = coeffs + (size_t)o * ksize;
        int acc = 1 << (PRECISION_BITS - 1);
        if (axis == 1) {
            const uint8_t* s = src + ((size_t)y * n1 + lo) * C + c;
            for (int j = 0; j < n; ++j) acc += (int)s[(size_t)j * C] * k[j];
        } else {
            const uint8_t* s = src + ((size_t)lo * n1 + x) * C + c;
            for (int j = 0; j < n; ++j) acc += (int)s[(size_t)j * n1 * C] * k[j];
        }
        dst[i] = clip8(acc);
    }
}

// tab[b] = {device pointer of image b (uint8 [h][w][3]), h, w}; out[b][c][y][x] = (u8/255 - mean[c]) / std[c] inside the
// image, 0 in the padding; mask[b][y][x] = 1 in the padding (NestedTensor convention: True = padded).
__global__ __launch_bounds__(256) void collate_norm_kernel(const long long* __restrict__ tab, float* __restrict__ out,
                                                           uint8_t* __restrict__ mask, int B, int H, int W,
                                                           float m0, float m1, float m2, float s0, float s1, float s2) {
    const size_t total = (size_t)B * H * W;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int x = (int)(i % W); const size_t t = i / W; const int y = (int)(t % H); const int b = (int)(t / H);
        const uint8_t* img = reinterpret_cast<const uint8_t*>(tab[3 * b]);
        const int h = (int)tab[3 * b + 1], w = (int)tab[3 * b + 2];
        const bool in = y < h && x < w;
        float v0 = 0.f, v1 = 0.f, v2 = 0.f;
        if (in) {
            const uint8_t* px = img + ((size_t)y * w + x) * 3;
            v0 = ((float)px[0] / 255.0f - m0) / s0; v1 = ((float)px[1] / 255.0f - m1) / s1; v2 = ((float)px[2] / 255.0f - m2) / s2;
        }
        const size_t plane = (size_t)H * W, o = (size_t)b * 3 * plane + (size_t)y * W + x;
        out[o] = v0; out[o + plane] = v1; out[o + 2 * plane] = v2;
        mask[i] = in ? 0 : 1;
    }
}

}  // namespace

extern "C" int rt_resample_u8(const void* src, void* dst, const int32_t* bounds, const int32_t* coeffs, int n0, int n1, int C,
                              int out_len, int ksize, int axis, rt_stream_t stream) {
    if (!src || !dst || !bounds || !coeffs) return RT_ERR_BADARG;
    if (n0 <= 0 || n1 <= 0 || C <= 0 || out_len <= 0 || ksize <= 0 || (axis != 0 && axis != 1)) return RT_ERR_BADARG;
    const size_t total = (size_t)(axis == 0 ? out_len : n0) * (axis == 1 ? out_len : n1) * C;
    size_t blocks = (total + 255) / 256; if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(resample_u8_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)src,
                       (uint8_t*)dst, bounds, coeffs, n0, n1, C, out_len, ksize, axis);
    RT_CHECK_LAUNCH();
    return RT_OK;
}

extern "C" int rt_img_collate_norm(const int64_t* table, float* out, uint8_t* mask, int B, int H, int W, const float* mean3,
                                   const float* std3, rt_stream_t stream) {
    if (!table || !out || !mask || !mean3 || !std3 || B <= 0 || H <= 0 || W <= 0) return RT_ERR_BADARG;
    size_t blocks = ((size_t)B * H * W + 255) / 256; if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(collate_norm_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const long long*)table, out,
                       mask, B, H, W, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2]);
    RT_CHECK_LAUNCH();
    return RT_OK;
}
