/*
 * reftr_hip.h — C ABI of libreftr_hip.so, the MI355X (gfx950) kernel library that sits under the
 * Python protocol of ubc-vision/RefTR's training hot path.
 *
 * The reference has NO FFI of its own: its boundary is the Python object protocol
 *   models/__init__.py:4-10          build_reftr(args) -> (model, criterion, postprocessors)
 *   models/reftr_transformer.py:159  model(samples) -> {'pred_boxes','phrase_mask','aux_outputs'}
 *   models/criterion.py:166          criterion(outputs, targets) -> {loss_bbox, loss_giou, ...}
 *   engine_vg.py:22-78               train_one_epoch(...)
 * and every arithmetic op underneath it is an implicit ATen/oneDNN/cuDNN call made by a torch.nn
 * module.  Each entry point below replaces one such implicit call; the comment on it names the
 * reference call site (file:line under /root/reference) whose arithmetic it carries.
 *
 * Conventions
 *   - plain pointers + sizes, no torch types; every pointer is DEVICE memory unless stated;
 *   - `stream` is a hipStream_t passed as void*; kernels are only enqueued, never synchronised;
 *   - no allocation inside the library: workspaces/outputs are caller-provided;
 *   - return value: 0 = ok, <0 = RT_ERR_* (argument/shape not supported), >0 = hipError_t;
 *   - activations are bf16 (NHWC for images, [rows, features] for tokens), statistics / residual
 *     streams / losses / optimizer state are fp32, masks are uint8 (1 = padded / ignore);
 *   - thread-compatible: one caller thread per device.
 */
#ifndef REFTR_HIP_H
#define REFTR_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* rt_stream_t; /* hipStream_t */

enum { RT_OK = 0, RT_ERR_BADARG = -1, RT_ERR_UNSUPPORTED = -2 };
enum { RT_ACT_NONE = 0, RT_ACT_RELU = 1, RT_ACT_GELU = 2, RT_ACT_TANH = 3 };

/* Library identity: returns the ABI version (bumped on any signature change). */
int rt_abi_version(void);
/* Writes the gfx arch string of device `dev` ("gfx950...") into buf; >0 hip error if no device. */
int rt_device_arch(int dev, char* buf, int buflen);

/* --------------------------------------------------------------------------------------------
 * rt_conv_gemm — bf16 MFMA implicit-GEMM: out[m, n] = epilogue( sum_k gather(src)[m, k] * wgt[n, k] )
 *
 * One kernel carries every dense contraction whose reduction axis is contiguous in both operands:
 *   - Conv2d forward, 1x1 / 3x3, stride 1|2 (torchvision ResNet bottlenecks reached through
 *     models/modeling/backbone.py:101-109,119-121; input_proj conv models/reftr_transformer.py:121-125);
 *     FrozenBatchNorm2d (backbone.py:70-80) is folded: scale into wgt, shift into `bias`;
 *     residual add + ReLU of the bottleneck tail fused (`res_bf16`, act);
 *   - Conv2d backward-data (transposed = 1, wgt = [Cin][KH][KW][Cout] re-layout) with the ReLU mask of
 *     the producer fused (`gate`), residual-branch gradient accumulated (`res_*`);
 *   - every nn.Linear forward / backward-data on the path (nn.MultiheadAttention in/out projections
 *     models/modeling/transformer.py:151,211-212; FFNs :153-155; BERT dense layers; mlp_mapping
 *     models/reftr_transformer.py:14-23; QueryEncoder :31-39; bbox MLP backbone.py:26-38) as a 1x1
 *     "conv" over B = rows, SH = SW = 1.
 *
 * src   bf16 [B, SH, SW, SC]  (NHWC)         wgt  bf16 [N][KH][KW][SC]
 * out   [B, DH, DW, N] written as bf16 (out_bf16) and/or fp32 (out_f32)
 * forward gather   : src pixel = (dy*stride - pad + kh, dx*stride - pad + kw)
 * transposed gather: src pixel = ((dy + pad - kh)/stride, (dx + pad - kw)/stride) where divisible
 * epilogue order   : +bias[n] -> act -> dropout(drop_p, drop_seed; index m*N+n) -> +res -> *gate -> *gelu'(preact)
 * constraints      : SC % 64 == 0, N % 4 == 0, stride in {1,2}
 * ------------------------------------------------------------------------------------------ */
typedef struct rt_conv_gemm_desc {
    const void*  src;
    const void*  wgt;
    void*        out_bf16;
    float*       out_f32;
    const float* bias;      /* [N] or NULL */
    const float* res_f32;   /* [M, N] or NULL */
    const void*  res_bf16;  /* [M, N] or NULL */
    const void*  gate;      /* bf16 [M, N] or NULL: out *= (gate > 0 ? gate_scale : 0) */
    const void*  preact;    /* bf16 [M, N] or NULL: out *= gelu'(preact) */
    int32_t B, SH, SW, SC;
    int32_t DH, DW, N;
    int32_t KH, KW, stride, pad;
    int32_t transposed;
    int32_t act;
    float    gate_scale;
    float    drop_p;
    uint32_t drop_seed;
    int32_t  tile_hint;     /* 0 = auto; otherwise 1: 128x128, 2: 128(m)x64(n), 3: 64x64 */
} rt_conv_gemm_desc;
int rt_conv_gemm(const rt_conv_gemm_desc* d, rt_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * rt_conv_wgrad — bf16 MFMA weight-gradient GEMM, reduction over rows (pixels / tokens):
 *   dw[n][kh][kw][c] += scale[n] * sum_m dy[m, n] * gather(x)[m, (kh,kw,c)]
 * Replaces the conv_backward(weight) / addmm(grad^T, input) calls autograd issues for every trainable
 * Conv2d (layer2-4, backbone.py:87-89; input_proj) and nn.Linear on the path.
 * dy  bf16 [B, DH, DW, N]     x  bf16 [B, SH, SW, SC]     dw  fp32 [N][KH][KW][SC] (accumulated, atomics)
 * Both operands are staged in their natural row-major layout and turned into MFMA fragments with the
 * gfx950 LDS transpose read (ds_read_b64_tr_b16).  The M axis is split over blocks (`msplit`, 0=auto).
 * constraints: SC % 16 == 0, N % 4 == 0
 * ------------------------------------------------------------------------------------------ */
typedef struct rt_conv_wgrad_desc {
    const void*  dy;
    const void*  x;
    float*       dw;
    const float* scale;   /* [N] or NULL (FrozenBN scale of the conv's BN) */
    int32_t B, SH, SW, SC;
    int32_t DH, DW, N;
    int32_t KH, KW, stride, pad;
    int32_t msplit;
} rt_conv_wgrad_desc;
int rt_conv_wgrad(const rt_conv_wgrad_desc* d, rt_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* REFTR_HIP_H */
