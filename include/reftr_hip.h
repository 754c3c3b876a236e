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

enum { RT_OK = 0, RT_ERR_BADARG = -1, RT_ERR_UNSUPPORTED = -2, RT_ERR_COMM = -3 };
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
 * epilogue order   : +bias[n] -> act -> dropout(drop_p, drop_seed; index m*N+n) -> +res -> *gate -> *gelu'(preact) -> *(1-dtanh^2)
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
    int32_t  tile_hint;     /* 0 = auto (the product path); otherwise a tile / stage / schedule variant by number (csrc/rt_gemm.hip): the product library
                               answers RT_ERR_BADARG for the variants only the lab library (-DRT_LAB) instantiates */
    void*    out_preact;    /* bf16 [M, N] or NULL: value after +bias, BEFORE act (saved for GELU backward) */
    const void* dtanh;      /* bf16 [M, N] or NULL: out *= (1 - dtanh^2)  (backward of a tanh output, BERT pooler) */
    int32_t  res_first;     /* 1: add res_* BEFORE act (bottleneck tail relu(bn(conv) + identity)); 0: after dropout */
    const uint32_t* seed_dev; /* optional DEVICE word: effective dropout seed = hash(*seed_dev, drop_seed) (hipGraph replay) */
    int32_t drop_shift;       /* dropout granularity: one keep/drop decision per 2^drop_shift consecutive output features of a row
                                 (hash index = (m * N + n) >> drop_shift).  With drop_shift = log2(head_dim) this is
                                 nn.MultiheadAttention's probability dropout for a ONE-key softmax (decoder self-attention with one
                                 query per image, transformer.py:231-236): the attention launch folds into the V projection */
    float* acc2_f32;          /* optional second fp32 destination [M, N] that the result is ADDED to (out_f32 / out_bf16 still receive
                                 the result itself): the q/k-side input gradient of an attention layer is both passed on and summed into
                                 the gradient of `pos` / `query_pos`, which every layer re-adds (transformer.py:154-156,168-175) */
    int32_t dil;              /* dilation of the taps (0 / 1 = none): tap (kh, kw) reads the source pixel kh * dil, kw * dil away (`--dilation`,
                                 models/modeling/backbone.py:117-125: layer4's 3x3 convolutions with dilation 2, pad 2, stride 1).
                                 dil > 1 needs stride 1 and the LDS-DMA tiles (every tile_hint but 1-3). */
} rt_conv_gemm_desc;
int rt_conv_gemm(const rt_conv_gemm_desc* d, rt_stream_t stream);
/* rt_conv_gemm_grouped — n independent rt_conv_gemm problems (HOST array of descriptors).  Dense products with K < 1024 and
 * more than 16 rows (2 <= n <= 12) run as ONE launch whose descriptors travel by value in the kernel arguments; any other
 * mix is issued as n single launches in order.  The results are those of n rt_conv_gemm calls either way.  On the
 * transformer's dependency chains (models/modeling/transformer.py:168-252: q/k and v projections of a layer, the decoder's
 * cross-attention K / V projections of all layers, pairs of backward-data products) a launch costs ~4.5 us whatever it computes. */
int rt_conv_gemm_grouped(const rt_conv_gemm_desc* descs, int n, rt_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * rt_conv_wgrad — bf16 MFMA weight-gradient GEMM, reduction over rows (pixels / tokens):
 *   dw[n][kh][kw][c] += scale[n] * sum_m dy[m, n] * gather(x)[m, (kh,kw,c)]
 * Replaces the conv_backward(weight) / addmm(grad^T, input) calls autograd issues for every trainable
 * Conv2d (layer2-4, backbone.py:87-89; input_proj) and nn.Linear on the path.
 * dy  bf16 [B, DH, DW, N]     x  bf16 [B, SH, SW, SC]     dw  fp32 [N][KH][KW][SC] (accumulated; `overwrite`: assigned)
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
    float*  dbias;        /* optional [N]: dbias[n] += sum_m dy[m, n] (fused bias gradient; no scale applied) */
    int32_t variant;      /* 0 = library's choice; >0 pins a kernel variant (tuning / tests, see rt_wgrad.hip) */
    float*  workspace;    /* optional scratch for the split-M partial tiles (plain stores + one reduction pass instead of
                             fp32 atomics); must not be shared by launches on concurrent streams.  NULL: atomics */
    int64_t workspace_bytes;
    int32_t overwrite;    /* 1: dw = result instead of dw += result (the caller guarantees this is the first contribution to dw since
                             the gradient buffer was last consumed: no pre-zeroed memory is needed and the epilogue skips the read of
                             dw).  dbias always accumulates. */
    int32_t dil;          /* dilation of the taps (0 / 1 = none; > 1 needs stride 1), as in rt_conv_gemm_desc */
    float*  sqacc;        /* optional gradient-norm accumulator (RT_SQ_SLOTS x RT_SQ_STRIDE floats, see rt_sqnorm_finish): the launch adds
                             |dw after|^2 - |dw before|^2, so that the clip norm (engine_vg.py:62-63) needs no pass over dw */
    void*   g16;          /* optional bf16 twin of dw (same shape): every value written to dw is also written there, rounded -- the
                             data-parallel exchange buffer (bf16 on the xGMI links, main_vg.py:290-296's all-reduce), which otherwise
                             costs a 607 MB read + 304 MB write rounding pass per step */
} rt_conv_wgrad_desc;
int rt_conv_wgrad(const rt_conv_wgrad_desc* d, rt_stream_t stream);

/* rt_small_wgrad_grouped — dw_i[N,K] += dy_i[M,N]^T x_i[M,K], dbias_i[N] += colsum(dy_i) for up to any number of independent
 * Linear weight gradients with M <= 16 token rows (decoder / query-encoder / box-head layers, transformer.py:231-252) in one
 * launch per 64 problems; `jobs` is a HOST array, copied into the kernel arguments. */
typedef struct rt_small_wgrad_job {
    const void* dy; const void* x; float* dw; float* dbias;
    int32_t M, N, K, overwrite;      /* overwrite: as in rt_conv_wgrad_desc (dbias accumulates) */
    float* sqacc;                    /* optional gradient-norm accumulator, as in rt_conv_wgrad_desc */
    void*  g16;                      /* optional bf16 twin of dw, as in rt_conv_wgrad_desc */
} rt_small_wgrad_job;
int rt_small_wgrad_grouped(const rt_small_wgrad_job* jobs, int njobs, rt_stream_t stream);

/* rt_conv_wgrad_grouped — `n` independent weight gradients (HOST array of descriptors) queued during backward and launched
 * together: the plain Linear ones (1x1, N and SC >= 128, more than 16 rows, variant = msplit = 0) share ONE launch of the
 * 128x128 kernel per 24 problems (descriptors by value in the kernel arguments) plus one grouped reduction of their split
 * partials, carved out of `workspace`; every other descriptor is forwarded to rt_conv_wgrad unchanged.  Results equal the
 * individual launches.  A transformer layer's weight gradients are 50-250 workgroups each and off the backward-data
 * dependency chain: grouped they fill the 256 CUs instead of serialising ~20 us launches.  Not re-entrant per device. */
int rt_conv_wgrad_grouped(const rt_conv_wgrad_desc* descs, int n, float* workspace, int64_t workspace_bytes, rt_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * rt_layernorm_fwd / rt_layernorm_bwd — nn.LayerNorm over the last axis, fp32 statistics, one wave per row.
 * Replaces: encoder/decoder norm1-3 and decoder.norm (models/modeling/transformer.py:157-158,217-219,
 * 176-180,243-251,131-138), BERT LayerNorms (eps 1e-12), the LayerNorms inside mlp_mapping
 * (models/reftr_transformer.py:14-23) and QueryEncoder.context_out (:36-39).
 * Fused around it: post-norm ReLU (+dropout) of mlp_mapping, bf16 operand copy for the next GEMM,
 * bf16(y + pos) for the q/k input of the next attention (with_pos_embed, transformer.py:165-166,172).
 * Output rows can be re-mapped so that per-image token blocks land inside the [B, S, D] sequence buffer:
 *   out_row = (r / grp_rows) * grp_stride + grp_off + r % grp_rows      (grp_rows = 0: identity)
 * Backward: dx (fp32) plus bf16(dx * dropout2-mask) — the gradient entering the GEMM whose output was
 * dropped-out and added to the residual before this norm; dgamma/dbeta accumulated with atomics.
 * ------------------------------------------------------------------------------------------ */
typedef struct rt_layernorm_desc {
    const float* x;        /* [M, D] */
    const float* gamma;    /* [D] */
    const float* beta;     /* [D] */
    float*       y_f32;    /* row-mapped, or NULL */
    void*        y_bf16;   /* row-mapped, or NULL */
    const float* pos;      /* row-mapped like the outputs, or NULL */
    void*        ypos_bf16;/* bf16(y + pos), or NULL */
    float*       mean;     /* [M] or NULL */
    float*       rstd;     /* [M] or NULL */
    int32_t M, D;
    float   eps;
    int32_t act;           /* RT_ACT_NONE | RT_ACT_RELU (after the affine) */
    float    drop_p;       /* dropout after act, index r*D + c */
    uint32_t drop_seed;
    int32_t grp_rows, grp_stride, grp_off;
    const uint32_t* seed_dev;   /* optional, see rt_conv_gemm_desc */
} rt_layernorm_desc;
int rt_layernorm_fwd(const rt_layernorm_desc* d, rt_stream_t stream);

typedef struct rt_layernorm_bwd_desc {
    const float* dy;       /* row-mapped grad of the LN output */
    const float* dy2;      /* optional second grad (same mapping) added to dy, or NULL */
    const float* x;        /* [M, D] LN input */
    const float* gamma;
    const float* beta;     /* needed when act == RELU */
    const float* mean;
    const float* rstd;
    float*       dx_f32;   /* [M, D] or NULL */
    void*        dx_bf16;  /* [M, D] bf16(dx * mask2 / (1-p2)) or NULL */
    float*       dgamma;   /* [D] accumulated, or NULL */
    float*       dbeta;    /* [D] accumulated, or NULL */
    int32_t M, D;
    int32_t act;
    float    drop_p;  uint32_t drop_seed;    /* the LN's own post-act dropout */
    float    drop2_p; uint32_t drop2_seed;   /* dropout of the producing sub-layer (index r*D + c) */
    int32_t grp_rows, grp_stride, grp_off;
    const uint32_t* seed_dev;   /* optional, applies to both dropout seeds */
    float*   partials;     /* optional [n_blocks][2][D] scratch: per-workgroup dgamma / dbeta partial sums are STORED there
                              (no atomics); the caller reduces them later with rt_ln_param_grad_grouped.  dgamma / dbeta
                              are then not touched.  n_blocks is returned through *n_blocks_out. */
    int32_t* n_blocks_out; /* host pointer, written when partials != NULL */
} rt_layernorm_bwd_desc;
int rt_layernorm_bwd(const rt_layernorm_bwd_desc* d, rt_stream_t stream);

/* rt_ln_param_grad_grouped — dgamma_i[D] += sum_b partials_i[b][0][:], dbeta_i[D] += sum_b partials_i[b][1][:] for any number
 * of LayerNorm backward launches (HOST array, descriptors by value in the kernel arguments).  The LayerNorm parameter
 * gradients are off the backward-data dependency chain: one launch per 64 of them instead of ~2*D contended atomics per
 * workgroup inside every rt_layernorm_bwd. */
typedef struct rt_ln_pg_job {
    const float* partials; float* dgamma; float* dbeta;
    int32_t n_blocks, D;
} rt_ln_pg_job;
int rt_ln_param_grad_grouped(const rt_ln_pg_job* jobs, int n, rt_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * rt_groupnorm_fwd / rt_groupnorm_bwd — nn.GroupNorm(G, C) of input_proj (models/reftr_transformer.py:
 * 121-125,174) on the token-major image x[b][pixel][c]; statistics per (image, group) over HW * C/G values.
 * Output rows go to (b * out_rows_per_img + out_row_off + pixel) so the image tokens land behind the
 * language tokens of the [B, S, C] sequence (models/reftr.py:115-117).  stats/bstats: [B, G, 2] workspaces.
 * The forward statistics are reduced in a fixed order (per-chunk partial sums, then a fixed shuffle tree): no atomics, so
 * the eval-mode forward is bit-reproducible run to run like the reference's.
 * ------------------------------------------------------------------------------------------ */
typedef struct rt_groupnorm_desc {
    const float* x;        /* [B, HW, C] */
    const float* gamma; const float* beta;
    float*       stats;    /* [B, G, 2] workspace: sum, sumsq (kept for backward) */
    float*       y_f32; void* y_bf16;
    const float* pos; void* ypos_bf16;
    int32_t B, HW, C, G;
    float   eps;
    int32_t out_rows_per_img, out_row_off;
    float*  partials;      /* [B, chunks, G, 2] workspace: per-pixel-chunk sums, 1 <= chunks <= 64 */
    int32_t chunks;
} rt_groupnorm_desc;
int rt_groupnorm_fwd(const rt_groupnorm_desc* d, rt_stream_t stream);

typedef struct rt_groupnorm_bwd_desc {
    const float* dy; const float* dy2;   /* sequence-mapped like the forward outputs */
    const float* x; const float* gamma;
    const float* stats;    /* from forward */
    float*       bstats;   /* [B, G, 2] workspace */
    float*       dx_f32; void* dx_bf16;  /* [B, HW, C] */
    float*       dgamma; float* dbeta;
    int32_t B, HW, C, G;
    float   eps;
    int32_t out_rows_per_img, out_row_off;
} rt_groupnorm_bwd_desc;
int rt_groupnorm_bwd(const rt_groupnorm_bwd_desc* d, rt_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * RES head (RefTRSeg, models/reftr_segmentation.py).  The 3x3 / 1x1 convolutions of MaskHeadSmallConv (:216-238) and
 * the k/q projections of MHAttentionMap (:186-187) run through rt_conv_gemm / rt_conv_wgrad with channel counts padded
 * to multiples of 64 (zero weights / activations in the padding); the entry points below are everything around them.
 *
 * rt_gn_nhwc_fwd / _bwd — GroupNorm(8, C) + ReLU (:242-248,256-258,...) over x[b][p][c] (fp32 conv output, row stride
 *   ldx); writes the next convolution's bf16 operand with row stride ldy >= C, padding zero-filled.
 *   stats / bstats: [B, G, 2] workspaces (sum, sumsq) kept for backward.  dy fp32 (row stride lddy), dx bf16 (lddx).
 *   Forward statistics go through `partials` ([B, partial_blocks, G, 2], partial_blocks >= ceil(HW / ceil(16384 / C))) and
 *   are added in a fixed order (no atomics): the eval-mode mask logits are bit-reproducible run to run.
 * rt_upsample_add / _bwd — `cur_fpn + F.interpolate(x, size, mode="nearest")` (:253,262,271).  Backward returns the
 *   pixel-summed gradient of the coarse operand and a bf16 copy of dy (the adapter convolution's output gradient).
 * rt_attn_map_fwd / _bwd — MHAttentionMap.forward (:196-208): softmax over (heads, h, w) jointly of
 *   norm * <q_head, k_head>, padded pixels (mask != 0) -> -inf.  k rows: (b * k_rows_per_img + k_row_off + p) * ldk.
 *   The probabilities are also written (bf16) into columns concat_col.. of the mask head's input rows.
 * rt_seg_concat — torch.cat([img_src_proj, memory_visual, bbox_mask]) of refer_segmentation (:165) / mask_head (:241).
 * rt_mask_loss — CriterionVGOnePhraseSeg.loss_masks (:314-337): bilinear (align_corners=False) upsample of the logits
 *   to the padded target size + sigmoid_focal_loss(alpha .25, gamma 2) + dice_loss (modeling/segmentation.py:178-221).
 *   dpred == NULL: forward (sums [B,4] workspace, losses[2] = {focal, dice}); else backward, accumulating into dpred.
 * ------------------------------------------------------------------------------------------ */
typedef struct rt_gn_nhwc_desc {
    const float* x; const float* gamma; const float* beta;
    float* stats; void* y_bf16;
    int32_t B, HW, C, G, ldx, ldy, act;
    float eps;
    float* partials; int32_t partial_blocks;
} rt_gn_nhwc_desc;
int rt_gn_nhwc_fwd(const rt_gn_nhwc_desc* d, rt_stream_t stream);

typedef struct rt_gn_nhwc_bwd_desc {
    const float* dy; const float* x; const float* gamma; const float* beta;
    const float* stats; float* bstats;
    void* dx_bf16; float* dgamma; float* dbeta;
    int32_t B, HW, C, G, ldx, lddy, lddx, act;
    float eps;
} rt_gn_nhwc_bwd_desc;
int rt_gn_nhwc_bwd(const rt_gn_nhwc_bwd_desc* d, rt_stream_t stream);

typedef struct rt_upsample_add_desc {
    const float* fpn; const void* a_bf16; void* out_bf16;
    int32_t B, H, W, h, w, C, ldf, lda, ldo;
} rt_upsample_add_desc;
int rt_upsample_add(const rt_upsample_add_desc* d, rt_stream_t stream);

typedef struct rt_upsample_add_bwd_desc {
    const float* dy; float* da; void* dy_bf16;
    int32_t B, H, W, h, w, C, lddy, ldda, lddyb;
} rt_upsample_add_bwd_desc;
int rt_upsample_add_bwd(const rt_upsample_add_bwd_desc* d, rt_stream_t stream);

typedef struct rt_attn_map_desc {
    const float* q; const float* k; const uint8_t* mask;
    float* P; void* concat_bf16;
    int32_t B, HW, E, nh, ldk, k_rows_per_img, k_row_off, ld_concat, concat_col;
    float norm;
} rt_attn_map_desc;
int rt_attn_map_fwd(const rt_attn_map_desc* d, rt_stream_t stream);

typedef struct rt_attn_map_bwd_desc {
    const float* q; const float* k; const float* P; const float* dconcat;
    float* dq; float* dk;
    int32_t B, HW, E, nh, ldk, k_rows_per_img, k_row_off, ld_dconcat, concat_col;
    float norm;
} rt_attn_map_bwd_desc;
int rt_attn_map_bwd(const rt_attn_map_bwd_desc* d, rt_stream_t stream);

typedef struct rt_seg_concat_desc {
    const float* src; const float* mem; void* out_bf16;
    int32_t B, HW, E, nh, ldo, mem_rows_per_img, mem_row_off, src_rows_per_img, src_row_off;
} rt_seg_concat_desc;
int rt_seg_concat(const rt_seg_concat_desc* d, rt_stream_t stream);

typedef struct rt_mask_loss_desc {
    const float* pred; const uint8_t* target;
    float* sums; float* losses;
    float* dpred; const float* g_focal; const float* g_dice;
    int32_t B, h, w, Ht, Wt, ldp, lddp;
    float inv_norm;
    float* gbuf;           /* backward only: scratch fp32 [B, Ht, Wt] (the per-target-pixel gradient between the two passes) */
} rt_mask_loss_desc;
int rt_mask_loss(const rt_mask_loss_desc* d, rt_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * rt_attn_fwd / rt_attn_bwd — softmax(scale * Q K^T + key_padding_mask) V per (batch, head), the core of
 * nn.MultiheadAttention (models/modeling/transformer.py:174-175,234-246) and of HF BertSelfAttention.
 * Element (b, i, h, d) of q/k/v/out lives at ptr[(b*S + i)*ld + h*dh + d] so packed projection outputs are
 * consumed in place.  kpm: uint8 [B, Sk], 1 = ignore (-inf); a fully masked row yields NaN like the
 * reference.  Dropout acts on the probabilities (index ((b*H+h)*Sq + i)*Sk + j).  lse = log-sum-exp per row
 * (saved for backward); delta is a [B,H,Sq] fp32 workspace.  dh in {32, 64}; Sq, Sk <= 768.
 * ------------------------------------------------------------------------------------------ */
typedef struct rt_attn_desc {
    const void* q; const void* k; const void* v;
    void*  out;
    float* lse;            /* [B, H, Sq] or NULL */
    const uint8_t* kpm;    /* [B, Sk] or NULL */
    int32_t B, H, Sq, Sk, dh;
    int32_t ldq, ldk, ldv, ldo;
    float    scale;
    float    drop_p; uint32_t drop_seed;
    const uint32_t* seed_dev;   /* optional, see rt_conv_gemm_desc */
} rt_attn_desc;
int rt_attn_fwd(const rt_attn_desc* d, rt_stream_t stream);

typedef struct rt_attn_bwd_desc {
    const void* q; const void* k; const void* v; const void* out; const void* dout;
    const float* lse;
    float* delta;          /* [B, H, Sq] workspace */
    const uint8_t* kpm;
    void* dq; void* dk; void* dv;
    int32_t B, H, Sq, Sk, dh;
    int32_t ldq, ldk, ldv, ldo, lddq, lddk, lddv;
    float    scale;
    float    drop_p; uint32_t drop_seed;
    const uint32_t* seed_dev;
} rt_attn_bwd_desc;
int rt_attn_bwd(const rt_attn_bwd_desc* d, rt_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * Backbone-side kernels (models/modeling/backbone.py:83-125 + torchvision ResNet stem).
 * rt_img_pack        NCHW fp32 [B,3,H,W] -> NHWC4 bf16 [B,Hp,Wp,4] with a 3-pixel zero halo on top/left
 *                    (Hp >= H+6, Wp >= W+6; the 4th channel and the halo are zero)
 * rt_stem_conv       conv1 7x7/2 pad 3 + FrozenBN + ReLU on MFMA: per kernel row one K=32 step
 *                    (8 taps x 4 channels); w bf16 [64][7][8][4] (BN scale folded), bias = BN shift;
 *                    out bf16 NHWC [B,Ho,Wo,64]; needs Wp >= 32*ceil(Wo/16)+6, Hp >= 2*Ho+5
 * rt_maxpool3x3s2    nn.MaxPool2d(3, 2, 1) on NHWC bf16
 * rt_stem_pool       rt_stem_conv + rt_maxpool3x3s2 in one launch (torchvision ResNet.forward: conv1, bn1, relu, maxpool; all
 *                    frozen, backbone.py:87-89): out bf16 NHWC [B,(Ho-1)/2+1,(Wo-1)/2+1,64], bit-identical to the two launches;
 *                    the stem output itself is never written; needs Wp >= 2*Wo+6, Hp >= 2*Ho+5
 * rt_weight_prep     fp32 master weight [N][T][C] (T = KH*KW, channels-last) -> bf16 [N][T][C] (dst) and/or
 *                    bf16 [C][T][N] (dst_t, the backward-data operand), times scale[n] (FrozenBN) if given
 * rt_stem_weight_prep fp32 [64][7][7][3] -> bf16 [64][7][8][4]
 * rt_bn_fold         FrozenBatchNorm2d (backbone.py:70-80): scale = w*rsqrt(rv+eps), shift = b - rm*scale
 * ------------------------------------------------------------------------------------------ */
int rt_img_pack(const float* img, void* out, int B, int H, int W, int Hp, int Wp, rt_stream_t stream);
int rt_stem_conv(const void* xp, const void* w, const float* bias, void* out,
                 int B, int Hp, int Wp, int Ho, int Wo, rt_stream_t stream);
int rt_maxpool3x3s2(const void* x, void* y, int B, int H, int W, int C, int Ho, int Wo, rt_stream_t stream);
int rt_stem_pool(const void* xp, const void* w, const float* bias, void* out,
                 int B, int Hp, int Wp, int Ho, int Wo, rt_stream_t stream);
int rt_weight_prep(const float* src, const float* scale, void* dst, void* dst_t, int N, int T, int C, rt_stream_t stream);

/* rt_bottleneck_fwd — one FROZEN stride-1 bottleneck of layer1 in ONE launch (models/modeling/backbone.py:87-89 freezes conv1 and
 * layer1; torchvision Bottleneck v1.5 with FrozenBatchNorm2d :43-80): conv1 1x1 + BN + ReLU -> conv2 3x3 pad 1 + BN + ReLU ->
 * conv3 1x1 + BN (+ x | + downsample 1x1 + BN) -> ReLU.  The 64-channel intermediates never leave LDS.  NHWC bf16 activations;
 * weights bf16 with the BN scale folded (rt_weight_prep layout [N][T][C]), b* = BN shifts.  planes must be 64;
 * cin = 256 with wd = bd = NULL (identity shortcut) or cin = 64 with the downsample operands.  B*H*W*256 < 2^31. */
typedef struct rt_bottleneck_desc {
    const void*  x;        /* bf16 [B,H,W,cin] */
    const void*  w1;       /* bf16 [64][cin] */
    const void*  w2;       /* bf16 [64][9][64] */
    const void*  w3;       /* bf16 [256][64] */
    const void*  wd;       /* bf16 [256][cin] or NULL */
    const float* b1;       /* [64] */
    const float* b2;       /* [64] */
    const float* b3;       /* [256] */
    const float* bd;       /* [256] or NULL */
    void*        out;      /* bf16 [B,H,W,256] */
    int32_t B, H, W, cin, planes;
    int32_t form;          /* 0 or 1: one 8 x 16 tile per workgroup, weights through an LDS ring (the forms 2 / 3 of round 4 were removed: RT_ERR_UNSUPPORTED) */
} rt_bottleneck_desc;
int rt_bottleneck_fwd(const rt_bottleneck_desc* d, rt_stream_t stream);
/* rt_weight_prep_batched — every per-step operand refresh in ONE launch.  table: DEVICE int64 [njobs][8] =
 * {src, scale|0, dst|0, dst_t|0, N, T, C, first_tile}; a job's tiles are 32(n) x 32(c) per tap, tiles numbered
 * consecutively over jobs (total_tiles = sum).  Both outputs are written coalesced (LDS tile transpose). */
int rt_weight_prep_batched(const int64_t* table, int njobs, int total_tiles, rt_stream_t stream);
int rt_stem_weight_prep(const float* src, const float* scale, void* dst, rt_stream_t stream);
int rt_bn_fold(const float* w, const float* b, const float* rm, const float* rv, float eps,
               float* scale, float* shift, int n, rt_stream_t stream);

/* rt_mask_posenc — pad-mask nearest downsample (backbone.py:107, exact/integer) + PositionEmbeddingSine
 * (models/modeling/position_encoding.py:36-56: normalize, scale 2*pi, T=1e4, eps 1e-6) + add_vec[c]
 * (level_embed[0] + token_type_embeddings[1], models/reftr.py:60,70-73).  mask uint8 [B,H,W] (1 = pad).
 * kpm_out[b*kpm_stride + kpm_off + pix] = downsampled mask; pos_out row = b*pos_rows_per_img + pos_row_off + pix. */
typedef struct rt_mask_posenc_desc {
    const uint8_t* mask;
    uint8_t* kpm_out;
    float*   pos_out;
    const float* add_vec;   /* [C] or NULL */
    int32_t B, H, W, h, w, C;
    int32_t kpm_stride, kpm_off;
    int32_t pos_rows_per_img, pos_row_off;
} rt_mask_posenc_desc;
int rt_mask_posenc(const rt_mask_posenc_desc* d, rt_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * Small fused memory-bound kernels around the GEMMs.
 * rt_colsum          db[n] += sum_m dy[m,n]   (bias gradients of every nn.Linear / input_proj conv)
 * rt_rows_add        out[map_o(r)] (=|+=) alpha * (a[map_a(r)] + b[map_b(r)]), fp32 and/or bf16 out;
 *                    row maps as in rt_layernorm (grp_rows = 0: identity; grp_rows < 0: broadcast
 *                    (r / -grp_rows) * stride + off); accumulate = 2 uses atomics.  Used for with_pos_embed
 *                    (transformer.py:165-166), residual-gradient accumulation, CLS-row gather/scatter.
 * rt_bert_embed_*    HF BertEmbeddings lookup word[ids] + pos[l] + type[0] and its scatter-add backward
 * rt_context_mask    models/reftr_transformer.py:224-248 (bool only): context mask + query mask
 * rt_qenc_attn_*     QueryEncoder CLS-key attention (models/reftr_transformer.py:48-55), fp32
 * ------------------------------------------------------------------------------------------ */
int rt_colsum(const void* dy, int is_bf16, float* db, int M, int N, rt_stream_t stream);

typedef struct rt_rows_add_desc {
    const float* a_f32; const void* a_bf16; const float* b_f32;
    float* out_f32; void* out_bf16;
    int32_t rows, D;
    float   alpha;
    int32_t accumulate;     /* out_f32 += instead of = */
    int32_t a_grp_rows, a_grp_stride, a_grp_off;
    int32_t b_grp_rows, b_grp_stride, b_grp_off;
    int32_t o_grp_rows, o_grp_stride, o_grp_off;
} rt_rows_add_desc;
int rt_rows_add(const rt_rows_add_desc* d, rt_stream_t stream);

int rt_bert_embed_fwd(const int64_t* ids, const float* word, const float* pos, const float* type0,
                      float* out, int rows, int L, int D, const int32_t* pos_ids, rt_stream_t stream);
int rt_bert_embed_bwd(const int64_t* ids, const float* de, float* dword, float* dpos, float* dtype0,
                      int rows, int L, int D, const int32_t* pos_ids, rt_stream_t stream);
/* pos_ids (optional, int32 [rows]): explicit position ids; NULL = row % L (BERT).  rt_roberta_pos_ids fills them as HF
 * RobertaEmbeddings does (create_position_ids_from_input_ids: cumsum(ids != pad) * (ids != pad) + pad), exact integers. */
int rt_roberta_pos_ids(const int64_t* ids, int32_t* pos_ids, int B, int L, int pad_idx, rt_stream_t stream);
int rt_context_mask(const uint8_t* smask, const uint8_t* phrase_mask, const int64_t* pos_l, const int64_t* pos_r,
                    uint8_t* ctx, uint8_t* qmask, int B, int L, int P, int Lp, rt_stream_t stream);
int rt_qenc_attn_fwd(const float* k, const float* qs, const float* vs, const uint8_t* ctx, float* w, float* c,
                     int B, int P, int L, int E, rt_stream_t stream);
int rt_qenc_attn_bwd(const float* k, const float* qs, const float* vs, const float* w, const float* dc,
                     float* dk, float* dqs, float* dvs, int B, int P, int L, int E, rt_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * rt_box_loss — CriterionVGMultiPhrase.loss_boxes for all NL decoder layers at once
 * (models/criterion.py:113-202, util/box_ops.py:17-69).  logits fp32 [NL,B,P,K,4] (pre-sigmoid),
 * valid uint8 [B,P*K] (phrase_mask, 1 = valid), targets fp32 [sum n_b, 4] cxcywh with tgt_off int32 [B+1],
 * num_boxes: DEVICE fp32 scalar (already all-reduced / world, clamp >= 1 applied here).
 * losses[l*2+0] = loss_bbox of layer l, losses[l*2+1] = loss_giou; total = sum_l w_bbox*l1 + w_giou*giou;
 * dlogits = d total / d logits (0 for invalid phrases).  Selection/ordering is exact.
 * Both outputs are cleared by the call (they are accumulated with atomics); total == losses + 2*NL (one buffer) costs one clear.
 * ------------------------------------------------------------------------------------------ */
typedef struct rt_box_loss_desc {
    const float*   logits;
    const uint8_t* valid;
    const float*   targets;
    const int32_t* tgt_off;
    const float*   num_boxes;
    float* losses; float* total; float* dlogits;
    int32_t NL, B, P, K;
    float   w_bbox, w_giou;
    const float* weights;   /* optional DEVICE [NL,2] per-layer (bbox, giou) weights overriding w_bbox / w_giou */
} rt_box_loss_desc;
int rt_box_loss(const rt_box_loss_desc* d, rt_stream_t stream);

/* rt_small_dgrad — backward-data of a Linear with N <= 8 outputs (the 4-wide bbox_embed.layers.2,
 * backbone.py:26-38): dx[m,k] = sum_n dy[m,n] w[n,k], optional ReLU gate (gate > 0), bf16 out.
 * rt_pos_grad    — reduces the gradient of the `pos` sequence (models/reftr.py:60-89) onto
 * lang_pos_embeddings [L,E], token_type_embeddings [2,E] and level_embed [1,E] (accumulated). */
int rt_small_dgrad(const float* dy, const float* w, const void* gate, void* dx, int M, int N, int K, rt_stream_t stream);
int rt_pos_grad(const float* dpos, float* d_lang_pos, float* d_type, float* d_level, int B, int S, int L, int E,
                rt_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * Optimizer step over the flat parameter / gradient buffers (engine_vg.py:62-67, main_vg.py:234-268).
 * rt_sqnorm      out[0] = sum g^2
 * rt_adamw_flat  g' = g * grad_scale; total = sqrt(gnorm_sq) * grad_scale; coef = min(1, max_norm/(total+1e-6))
 *                (clip_grad_norm_, skipped when max_norm <= 0); AdamW with decoupled weight decay, bias
 *                correction at `step` (1-based); per-range lr / wd = the reference's param groups.
 *                span_begin < span_end restricts the pass to that element span (the update of one step may be issued as
 *                several launches on different streams, e.g. the BERT slice concurrently with the ResNet forward);
 *                `active` (DEVICE word, optional): the launch is a no-op while it is 0 (nothing pending under graph
 *                replay); `lr_dev` (DEVICE float[n_ranges], optional) overrides range_lr, so a replayed graph follows
 *                the learning-rate schedule without re-capture.
 * ------------------------------------------------------------------------------------------ */
int rt_sqnorm(const float* g, int64_t n, float* out, rt_stream_t stream);
/* the same over a bf16 gradient buffer (data-parallel runs exchange the gradients in bf16, reftr_amd/parallel.py) */
int rt_sqnorm_bf16(const void* g16, int64_t n, float* out, rt_stream_t stream);

typedef struct rt_adamw_desc {
    float* p; const float* g; float* m; float* v;
    int64_t n;
    const float* gnorm_sq;  /* device scalar from rt_sqnorm, or NULL */
    float* gnorm_out;       /* device scalar: total norm (after grad_scale), or NULL */
    float grad_scale, max_norm, beta1, beta2, eps;
    int32_t step, n_ranges;
    int64_t range_begin[8], range_end[8];
    float   range_lr[8], range_wd[8];
    const int32_t* step_dev;  /* optional DEVICE word overriding `step` (the optimizer step counter under hipGraph replay) */
    const int32_t* active;    /* optional DEVICE word: 0 = skip this launch */
    const float*   lr_dev;    /* optional DEVICE [n_ranges] learning rates overriding range_lr */
    int64_t span_begin, span_end;   /* element span to update (multiples of 4); 0, 0 = the whole buffer */
    const void* g16;          /* optional bf16 gradient buffer read INSTEAD of g (the exchanged gradients of a data-parallel run) */
} rt_adamw_desc;
int rt_adamw_flat(const rt_adamw_desc* d, rt_stream_t stream);
/* rt_sgd_flat — torch.optim.SGD(momentum, weight_decay), the reference's --sgd optimizer (main_vg.py:263-265), over the same flat
 * buffers and with the same descriptor: m = momentum buffer (zero-initialised), beta1 = momentum, v / beta2 / eps / step unused:
 *   g' = clip * grad_scale * g + wd * p;  m = beta1 * m + g';  p -= lr * m.   Clip coefficient and 1/world as in rt_adamw_flat. */
int rt_sgd_flat(const rt_adamw_desc* d, rt_stream_t stream);
/* rt_zero_chunks — clears `n` chunks of an fp32 buffer in one launch: table (DEVICE, static) = n x {int64 element offset, int64
 * element count (<= 16384)}.  Used for the gradient tensors that are accumulated with atomics (biases, norm parameters,
 * embeddings) when the weight matrices are produced in overwrite mode and the full clear of the gradient buffer is skipped. */
int rt_zero_chunks(float* base, const int64_t* table, int n, rt_stream_t stream);
/* rt_sqnorm_finish — the clip norm without a pass over the weight gradients.  Every weight-gradient launch given `sqacc` has added
 * |dw after|^2 - |dw before|^2 of what it wrote to the accumulator's slots (RT_SQ_SLOTS words, RT_SQ_STRIDE floats apart: the caller
 * clears the RT_SQ_SLOTS * RT_SQ_STRIDE floats when it clears / re-arms the gradient buffer).  This call adds the squared norm of
 * the tensors no such launch produces -- biases, norm parameters, embeddings, accumulated with atomics: the chunks of `table`
 * (DEVICE int64 [nchunks][2] = {element offset in `base`, count <= 16384}) -- and writes out[0] = sum of the slots (+ extra[0]). */
#define RT_SQ_SLOTS 256
#define RT_SQ_STRIDE 32
int rt_sqnorm_finish(const float* base, const int64_t* table, int nchunks, float* slots, const float* extra, float* out, rt_stream_t stream);
/* rt_round_chunks — twin[i] = bf16(base[i]) over `n` chunks {element offset, count <= 16384} (DEVICE int64 table): the tensors of a
 * gradient slice that no weight-gradient launch writes (biases, norm parameters, embeddings) on their way into the bf16 exchange
 * buffer; the weight matrices arrive there through rt_conv_wgrad_desc.g16. */
int rt_round_chunks(const float* base, void* twin, const int64_t* table, int n, rt_stream_t stream);
/* rt_counter_add — *ctr += inc on the device (step / dropout-seed counters that must advance inside a captured graph). */
int rt_counter_add(int32_t* ctr, int32_t inc, rt_stream_t stream);
/* rt_counter_add_if_zero — the same, but only while the DEVICE word *cond is 0; otherwise *ctr is left alone (reset_else = 0) or
 * cleared (reset_else = 1).  The engine advances the optimizer step counter (left alone) and the "update pending" word (cleared) of
 * an iteration through it with cond = the cooperative decoder's failure word: an iteration whose launches reported a hand-off
 * timeout never arms its AdamW update (engine_vg.py:55-58 of the reference stops BEFORE the update when an iteration is bad; here
 * the decision is taken on the device, inside the captured graph). */
int rt_counter_add_if_zero(int32_t* ctr, int32_t inc, const uint32_t* cond, int reset_else, rt_stream_t stream);
/* rt_finish_step — both counters of an iteration's end in ONE launch, with the reference's second stop condition carried to the
 * device as well: *step += 1 and *active += 1 unless (*cond != 0, cond optional) or (*loss is not finite, loss optional: the
 * iteration's weighted total); otherwise *active = 0 and *step is left alone.  engine_vg.py:53-58 stops BEFORE the update when the
 * loss is not finite; with the decision on the device the host may launch iteration i + 1 before it has read iteration i's numbers
 * (a pipelined loop): a bad iteration's update is never applied, the host stops one read later. */
int rt_finish_step(int32_t* step, int32_t* active, const uint32_t* cond, const float* loss, rt_stream_t stream);
/* rt_finish_stats — rt_finish_step plus everything else the loop reads from an iteration, in the same one-thread launch: the total
 * gradient norm grad_norm[0] = sqrt(sq[0]) * norm_scale (clip_grad_norm_'s return value, engine_vg.py:62-66) and the iteration's stats
 * vector stats = [*src[0] ... *src[n_src-1] | float(*cond) if cond_in_stats | grad_norm] -- the unweighted losses the meters log
 * (engine_vg.py:46-53, util/misc.py:156-160) as device scalars wherever the criterion left them.  Replaces four launches at the end of
 * every replayed step (counters, sqrt, a dtype conversion, a concatenation). */
#define RT_STATS_MAX 40
typedef struct {
    int32_t *step, *active;  const uint32_t* cond;  const float* loss;          /* as rt_finish_step */
    const float* sq;  float norm_scale;  float* grad_norm;                      /* optional (sq = NULL: grad_norm untouched) */
    const float* src[RT_STATS_MAX];  int n_src;  int cond_in_stats;  float* stats;   /* optional (stats = NULL) */
} rt_finish_desc;
int rt_finish_stats(const rt_finish_desc* d, rt_stream_t stream);
/* rt_stamp — buf[idx] = the device's constant 100 MHz wall clock (s_memrealtime) when the stream reaches this point: a one-thread
 * kernel the measurement tools capture into the step's graph at phase boundaries of every stream (tools/concurrent_timeline.py),
 * because rocprofv3 serialises the graph's concurrent streams. */
int rt_stamp(uint64_t* buf, int idx, rt_stream_t stream);
/* rt_adamw_mat / rt_adamw_chunks — rt_adamw_flat's update (same descriptor, same arithmetic and rounding, same `active` / step /
 * learning-rate device words), split into (a) the weight MATRICES, walked in tiles of 32 rows x 256 columns of [N][T*C] so that the kernel also writes the bf16
 * GEMM operands of the NEXT forward / backward from the registers that hold the new fp32 values -- what rt_weight_prep_batched
 * produced in a second pass over the masters -- and (b) everything else (biases, norm parameters, embeddings), in chunks.
 *   mat table (DEVICE int64 [njobs][8], static): {element offset of the matrix [N][T][C] in p / g / m / v, scale pointer | 0
 *     (FrozenBN scale[n] folded into the bf16 copies only), dst bf16 [N][T][C] | 0, dst_t bf16 [C][T][N] | 0, N, T, C, first tile};
 *     a job has ceil(N/32) * ceil(T*C/256) tiles; total_tiles = their sum; a matrix lies inside ONE learning-rate range.
 *     A job with dst = dst_t = 0 (an embedding table) may carry, in the scale slot, a DEVICE uint8 array [N][ceil(T*C/256)] of
 *     state bytes (0 = "m and v of this 256-column piece of row n are all zero"; start it at 1 = unknown; the kernel maintains
 *     it; reset it to 1 whenever m / v are written by anyone else).  A piece with byte 0 AND an all-zero gradient is skipped
 *     without reading p / m / v whenever (1 - lr * wd) rounds to 1.0f: with g = m = v = 0 the update is exactly that factor
 *     (torch.optim.AdamW's own fp32 no-op at the reference's lr_bert * weight_decay = 1e-9), so results are bit-identical.
 *   chunk table (DEVICE int64 [nchunks][2], static): {element offset, element count <= 16384}, both multiples of 4.
 * d->span_* are ignored (the tables say what is updated).  The caller makes jobs and chunks tile the parameters exactly once. */
int rt_adamw_mat(const rt_adamw_desc* d, const int64_t* table, int njobs, int total_tiles, rt_stream_t stream);
int rt_adamw_chunks(const rt_adamw_desc* d, const int64_t* table, int nchunks, rt_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * Host input pipeline on the device — integer / byte work, bit-exact.
 * rt_resample_u8 — one pass (axis 0 = vertical, 1 = horizontal) of Pillow's 8-bit ImagingResample, the resampler behind
 *   datasets/transforms.py:81-116 (torchvision F.resize of a PIL image, bilinear + antialias): src uint8 [n0][n1][C];
 *   out[o] = clip8((2^21 + sum_t src[lo_o + t] * k_o[t]) >> 22) with bounds[o] = {lo, taps} and coeffs[o][ksize] the 22-bit
 *   fixed-point taps (computed by the caller in double precision as Pillow does, reftr_amd/data/resample.py).
 * rt_img_collate_norm — ToTensor + Normalize (transforms.py:233-235,247-250) + nested_tensor_from_tensor_list
 *   (util/collate_fn.py:24-41): table[b] = {device pointer of image b (uint8 [h][w][3]), h, w} ->
 *   out fp32 [B,3,H,W] = (u8/255 - mean)/std zero-padded, mask uint8 [B,H,W] (1 = padding).
 * ------------------------------------------------------------------------------------------ */
int rt_resample_u8(const void* src, void* dst, const int32_t* bounds, const int32_t* coeffs, int n0, int n1, int C,
                   int out_len, int ksize, int axis, rt_stream_t stream);
int rt_img_collate_norm(const int64_t* table, float* out, uint8_t* mask, int B, int H, int W, const float* mean3,
                        const float* std3, rt_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * rt_cem_fwd / rt_cem_bwd — the CEM block of RefTRSeg (--ablation cem_loss, models/reftr_segmentation.py:16-41) for the one
 * query per image of RES: u_b = c3(hs_b) (hs bf16 [B, E] = last decoder output, w3 [16, E], b3 [16]), res bf16 [B*HW, ld] =
 * MaskHeadSmallConv's last feature map (16 channels in front of the padding), a_p = res_p . w2 + b2,
 *   energy_b = sum_p softmax_p(a)_p * clamp((cos(u_b, res_p) + 1) / 2, 1e-6, 1 - 1e-6),  loss = -sum_b log(energy_b + 1e-6) / B
 * (the softmax over the single query, c1, is the constant 1).  One workgroup per image, online softmax (one pass over the pixels).
 * Backward (g = d loss): dres fp32 [B*HW, lddr] += d res, dhs [B, E] = d hs, dw3 / db3 / dw2 += (d c2.bias = d c1 = 0 exactly).
 * ------------------------------------------------------------------------------------------ */
typedef struct rt_cem_desc {
    const void*  hs;        /* bf16 [B, E] */
    const float* w3;        /* [16, E] */
    const float* b3;        /* [16] */
    const void*  res;
    const float* w2;        /* [16] */
    const float* b2;        /* [1] */
    float* u;               /* [B, 16]: written by the forward, read by the backward */
    float* energy;          /* [B] */
    float* stats;           /* [B, 2] = {max_p a_p, sum_p exp(a_p - max)} */
    float* loss;            /* [1] */
    const float* g;         /* backward: device scalar d loss */
    float* dres; float* dhs; float* dw3; float* db3; float* dw2;
    int32_t B, HW, ld, lddr, E, reserved;
} rt_cem_desc;
int rt_cem_fwd(const rt_cem_desc* d, rt_stream_t stream);
int rt_cem_bwd(const rt_cem_desc* d, rt_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * Evaluation post-processing, exact decisions / selections.
 * rt_mask_postprocess — PostProcessSegm.forward (models/reftr_segmentation.py:282-302): pred fp32 [B*Q, h, w] mask logits ->
 *   masks uint8 [B, Q, max_h, max_w] = sigmoid(bilinear(pred -> max_h x max_w, align_corners=False)) > threshold inside the
 *   image's own (sizes[b] = {img_h, img_w}) top-left corner, 0 outside it (the reference's crop = the [:img_h, :img_w] view);
 *   masks_origin (optional) uint8, image b's block at origin_off[b] with shape [Q, orig_h, orig_w] (orig[b] = {orig_h, orig_w}) =
 *   nearest resize of the cropped mask (torch 'nearest': src = min(floor(dst * in / out), in - 1)); max_origin = max_b orig_h*orig_w.
 * rt_box_postprocess  — PostProcessVGMultiPhrase.forward (models/post_process.py:45-83): boxes fp32 [B,P,K,4] cxcywh, valid uint8
 *   [B,P,K] (phrase_mask) -> out fp32 [B,P,4]: row r of image b = xyxy of prediction 0 of its r-th VALID phrase (phrase order
 *   kept), times {w,h,w,h} when sizes (fp32 [B,2] = {img_h, img_w}) is given; counts int32 [B] = valid phrases per image.
 * ------------------------------------------------------------------------------------------ */
typedef struct rt_mask_post_desc {
    const float*   pred;
    const int32_t* sizes;
    const int32_t* orig;
    const int64_t* origin_off;
    uint8_t* masks;
    uint8_t* masks_origin;
    int32_t B, Q, h, w, max_h, max_w;
    int64_t max_origin;
    float   threshold;
} rt_mask_post_desc;
int rt_mask_postprocess(const rt_mask_post_desc* d, rt_stream_t stream);
typedef struct rt_box_post_desc {
    const float*   boxes;
    const uint8_t* valid;
    const float*   sizes;
    float*   out;
    int32_t* counts;
    int32_t B, P, K;
} rt_box_post_desc;
int rt_box_postprocess(const rt_box_post_desc* d, rt_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * rt_comm_* — the data-parallel gradient exchange: what the reference gets from
 * torch.nn.parallel.DistributedDataParallel (main_vg.py:290-296) after util/misc.py:392-431 set up one process per GPU.
 * RCCL over xGMI, bound at run time (dlopen: the copy the process already holds -- torch.distributed's -- or the system's);
 * RT_ERR_UNSUPPORTED when no RCCL can be loaded, RT_ERR_COMM (+ a line on stderr) when RCCL reports an error.
 *   rt_comm_unique_id  rank 0 fills 128 opaque bytes; the caller carries them to every rank (any out-of-band channel);
 *   rt_comm_init       collective over all ranks (one process per GPU, the device current at the call is the rank's);
 *   rt_comm_allreduce  in-place SUM all-reduce of `n` buffers (device pointers, element counts) of one dtype, issued as one
 *                      group on `stream` -- asynchronous: ordered behind the work already enqueued on `stream`, the caller
 *                      joins its compute stream with an event;
 *   rt_comm_destroy    releases the communicator.
 * Thread-compatible: one caller thread per device.
 * ------------------------------------------------------------------------------------------ */
typedef void* rt_comm_t;
enum { RT_COMM_F32 = 0, RT_COMM_BF16 = 1 };
int rt_comm_unique_id(void* id128);
int rt_comm_init(const void* id128, int rank, int world, rt_comm_t* out);
int rt_comm_allreduce(rt_comm_t comm, void* const* bufs, const int64_t* counts, int n, int dtype, rt_stream_t stream);
int rt_comm_destroy(rt_comm_t comm);

/* --------------------------------------------------------------------------------------------
 * rt_decoder_fwd — the TransformerDecoder stack of the referring queries as ONE cooperative launch
 * (models/modeling/transformer.py:105-143 the stack, :231-252 one layer, forward_post; one query per image: M = B rows, the
 * self-attention over a single key is the V projection with its per-head probability dropout).  Equivalent, operation for
 * operation, to the launched chain  rt_conv_gemm (self_attn.v, out_proj) -> rt_layernorm_fwd -> rt_conv_gemm (multihead_attn.q) ->
 * rt_attn_fwd -> rt_conv_gemm (out_proj) -> rt_layernorm_fwd -> rt_conv_gemm (linear1, linear2) -> rt_layernorm_fwd  per layer:
 * the saved tensors it writes are the chain's (bit-identical), so the backward is unchanged.  Workgroups stay resident and hand
 * the row block from stage to stage through global memory as tagged 8-byte units (csrc/rt_decoder.hip).
 *   handoff: RT_DEC_HANDOFF_BYTES of device memory, zeroed ONCE by the caller and then owned by the launches (word 0 = launch
 *            epoch, advanced by every launch; word 1 is set to 1 if a consumer gave up waiting for its rows).  Launches that
 *            share a buffer must be ordered on one stream.
 *   K / V of layer l: bf16 [B * S, ldkv] (multihead_attn.k / .v of memory + pos / memory, computed beforehand for every layer).
 * Supported: width 256, 8 heads, F = 2048, M <= 16, S <= 768, n_layers <= RT_DEC_MAX_LAYERS; else RT_ERR_UNSUPPORTED.
 * ------------------------------------------------------------------------------------------ */
#define RT_DEC_MAX_LAYERS 8
#define RT_DEC_HANDOFF_BYTES (256 + 16 * 256 * 36 + 16 * 2048 * 4)
typedef struct rt_decoder_layer_fwd {
    const void *Wv, *Wo, *Wq, *Wo2, *W1, *W2;            /* bf16 [N][K]: self_attn v / out_proj, multihead_attn q / out_proj, linear1 / 2 */
    const float *bv, *bo, *bq, *bo2, *b1, *b2;
    const float *g1, *be1, *g2, *be2, *g3, *be3;         /* norm1..3 weight / bias */
    const void *k2, *v2;
    void *o, *t1q16, *q2, *o2, *t2_16, *hdn, *t3_16;      /* bf16 saved tensors: [M,256] (hdn: [M,F]) */
    float *u, *u2, *u3;                                   /* fp32 pre-norm sums [M,256] */
    float *mean1, *rstd1, *mean2, *rstd2, *mean3, *rstd3; /* [M] */
    float *lse2;                                          /* [B, H] */
    float *t3_f32;                                        /* norm3 output rows, fp32 [M,256] (the shared decoder norm's input) */
    uint32_t seed_ad, seed_d1, seed_ad2, seed_d2, seed_dh, seed_d3;   /* dropout sites, in the chain's order */
} rt_decoder_layer_fwd;
typedef struct rt_decoder_fwd_desc {
    rt_decoder_layer_fwd layer[RT_DEC_MAX_LAYERS];
    const float*   t32;        /* [M,256] the stack's input (tgt) */
    const void*    t16;        /* bf16 copy */
    const float*   qpos;       /* [M,256] query_pos */
    const uint8_t* kpm;        /* [B, S] key padding mask (1 = ignore) or NULL */
    uint32_t*      handoff;
    const uint32_t* seed_dev;  /* optional, see rt_conv_gemm_desc */
    int32_t n_layers, M, H, S, F, ldkv;
    float   drop_p, eps, scale;
} rt_decoder_fwd_desc;
int rt_decoder_fwd(const rt_decoder_fwd_desc* d, rt_stream_t stream);
/* rt_decoder_bwd -- the backward of the same stack as one cooperative launch: per layer LayerNorm backward (norm3, norm2, norm1,
 * recomputed by every consumer), the six M <= 16 backward-data products, the one-query attention backward; the launched chain
 * rt_layernorm_bwd -> rt_conv_gemm (linear2^T gate, linear1^T + res) -> rt_layernorm_bwd -> rt_conv_gemm (out_proj^T) -> rt_attn_bwd
 * -> rt_conv_gemm (q^T) -> rt_layernorm_bwd -> rt_conv_gemm (out_proj^T head mask, v^T + res) of transformer.py:231-252's backward.
 * It writes what the chain writes for the launches that stay outside: the bf16 dy operands of the (grouped) weight gradients, the
 * LayerNorm parameter-gradient partials (rt_ln_param_grad_grouped's format, (M+3)/4 blocks), dK / dV of the cross-attention (the
 * M = B*S memory-gradient products and weight gradients), the gradient w.r.t. the stack's input and query_pos.  Same hand-off buffer
 * and limits as rt_decoder_fwd. */
typedef struct rt_decoder_layer_bwd {
    const void *WT2, *WT1, *WTo2, *WTq, *WTo, *WTv;      /* bf16 backward-data operands [K_in][N_out]: linear2, linear1, multihead_attn.out_proj / q, self_attn.out_proj / v */
    const float *g1, *g2, *g3;                            /* norm1..3 weight */
    const float *u, *u2, *u3, *mean1, *rstd1, *mean2, *rstd2, *mean3, *rstd3;      /* saved by the forward */
    const void *hdn, *q2, *k2, *v2, *o2;
    const float *lse2;
    const float *dnorm;                                   /* [M,256]: gradient w.r.t. this layer's output through the shared decoder norm */
    void *du3b, *dhdn, *du2b, *dq2, *dub, *dv;            /* bf16 [M,256] (dhdn [M,F]): dy operands of the weight gradients */
    void *dk2, *dv2;                                      /* bf16 [B*S, ldkv] */
    void *dk2p, *dv2p;                                    /* optional second copy with row stride ldkvp (all layers packed side by side
                                                             for ONE K-concatenated memory-gradient product), or NULL */
    float *part1, *part2, *part3;                         /* [(M+3)/4][2][256] */
    uint32_t seed_ad, seed_d1, seed_ad2, seed_d2, seed_d3, reserved;
} rt_decoder_layer_bwd;
typedef struct rt_decoder_bwd_desc {
    rt_decoder_layer_bwd layer[RT_DEC_MAX_LAYERS];
    float*          dta;       /* out [M,256]: gradient w.r.t. the stack's input (tgt) */
    float*          dqpos;     /* [M,256], accumulated (+=): gradient w.r.t. query_pos */
    const uint8_t*  kpm;
    uint32_t*       handoff;
    const uint32_t* seed_dev;
    int32_t n_layers, M, H, S, F, ldkv;
    float   drop_p, scale, gate_scale;                    /* gate_scale = 1 / (1 - p) of linear1's dropout */
    int32_t ldkvp;
} rt_decoder_bwd_desc;
int rt_decoder_bwd(const rt_decoder_bwd_desc* d, rt_stream_t stream);
/* REFTR_DEC_TRACE=1 only: 1024 wall-clock stamps (100 MHz) of the last launch's stage boundaries, host buffer */
int rt_decoder_trace(uint32_t* out1024);
/* rt_decoder_supported — RT_OK when rt_decoder_fwd / rt_decoder_bwd may be used on the CURRENT device: their G spin-waiting
 * workgroups must be co-resident, which is checked with the occupancy query (>= 1 workgroup of each kernel per compute unit at its
 * LDS size) and a 2x margin of compute units over G; RT_ERR_UNSUPPORTED otherwise (the launchers answer the same, and the caller
 * keeps the launched chain).  rt_decoder_set_spin — polls a consumer makes before it gives up and raises the failure word
 * (<= 0: the default, 2^20 / REFTR_DEC_SPIN); tests force a timeout with 1. */
int rt_decoder_supported(int F);
int rt_decoder_set_spin(int spin);


/* --------------------------------------------------------------------------------------------
 * The few-row region between the encoder's forward and backward as three launches (round 5, csrc/rt_qregion.hip).
 * Rows: N = B * P phrase rows (P <= 16 phrases per image), L <= 96 language tokens, E = 256.  Every workgroup works alone (no
 * grid hand-off); the tensors listed as outputs are exactly what the launched chain (rt_conv_gemm + rt_layernorm_* + rt_rows_add +
 * rt_qenc_attn_* + rt_box_loss + rt_small_dgrad) leaves behind, in the same formats, so forward / backward may mix fused and
 * launched halves and the weight gradients stay with rt_small_wgrad_grouped / rt_conv_wgrad_grouped.
 *
 * rt_qenc_fwd  -- QueryEncoder.forward (models/reftr_transformer.py:41-66) + the query / query_pos split (:60-66), one workgroup
 *                 per image.  cat16 [N, 2E]: columns E.. hold map_phrase's output on entry, columns ..E are written here.
 * rt_head_loss -- decoder.norm on every layer's output (models/modeling/transformer.py:131-141), bbox_embed (models/modeling/
 *                 backbone.py:26-38, models/reftr_transformer.py:287), CriterionVGMultiPhrase's box losses for all layers
 *                 (models/criterion.py:113-153) with d total / d logits for the per-layer weights `weights` [NL][2], and the
 *                 backward-data of the MLP and of the norm: one workgroup per decoder layer; losses [NL][2] are WRITTEN (no clear
 *                 needed, no atomics: run-to-run reproducible).  Nothing is accumulated: the launch may run before the gradient clear.
 * rt_qenc_bwd  -- backward-data of rt_qenc_fwd: from ga (+ gb) = d tgt and dqpos = d query_pos down to d memory (language rows and
 *                 the CLS row of every image are accumulated in place; nothing else writes them while the launch runs), d
 *                 query_embed (accumulated, n_q = 1), the bf16 dy operands of the six Linear weight gradients, and per-workgroup
 *                 LayerNorm parameter-gradient partials [B][2][E] for rt_ln_param_grad_grouped.
 * ------------------------------------------------------------------------------------------ */
typedef struct rt_qenc_fwd_desc {
    const void*  mem16;  const float* mem32;  const uint8_t* ctx;     /* memory bf16 / fp32 [B*S, E]; context mask [B, P, L] */
    const void *W1, *W2, *W3, *Wc, *Wf0, *Wf4;                          /* bf16 [E][E] (Wf0: [E][2E]) */
    const float *b1, *b2, *b3, *bc, *bf0, *bf4;
    const float *gc, *betc, *g1, *bet1, *g5, *bet5;                     /* context_out.1, fuse_encoder_query.1 / .5 */
    const float* qembed;                                                /* query_embed.weight [nq, 2E] */
    const uint32_t* seed_dev;
    void *cls16, *lang16;  float *kq, *qs, *vs, *qw;                    /* saved: bf16 [B,E], [B*L,E]; fp32 [B,E], [B*L,E] x 2, [B,P,L] */
    void* c16;  float *co, *cmean, *crstd;                              /* bf16 [N,E]; fp32 [N,E], [N], [N] */
    void* cat16;                                                        /* bf16 [N, 2E] */
    float *t1, *m1, *r1;  void* a16;  float *t2, *m2, *r2;
    float* tgt32;  void* tgt16;  float* qpos;  void* tgtq16;            /* [N*nq, E] */
    int32_t B, S, L, P, nq, E;
    float   eps, drop_p;
    uint32_t drop_seed;  int32_t reserved;
} rt_qenc_fwd_desc;
int rt_qenc_fwd(const rt_qenc_fwd_desc* d, rt_stream_t stream);

typedef struct rt_head_loss_desc {
    const float* t3;                                                    /* fp32 [NL*N, E]: every decoder layer's output */
    const float *gn, *betn;                                             /* decoder.norm */
    const void *W0, *W1, *W2, *W0T, *W1T;                               /* bf16 [E][E], [E][E], [4][E]; backward-data operands [E][E] */
    const float *b0, *b1, *b2, *w2_f32;                                 /* fp32 biases; fp32 master of the last Linear [4][E] */
    const uint8_t* valid;  const float* targets;  const int32_t* tgt_off;  const float* num_boxes;  const float* weights;   /* as rt_box_loss */
    void *hs16, *y1, *y2;  float *hmean, *hrstd;                        /* saved: bf16 [NL*N, E] x 3; fp32 [NL*N] x 2 */
    float *logits, *losses, *dlogits;  void* dl16;                      /* fp32 [NL*N, 4], [NL][2], [NL*N, 4]; bf16 [NL*N, 4] */
    void *dy2, *dy1;  float *dhs, *dnorm;                               /* bf16 [NL*N, E] x 2; fp32 [NL*N, E] x 2 */
    float* db2_part;                                                    /* optional: [NL][2][4], row 0 of a pair = the layer's share of d bias of the last Linear (rt_ln_param_grad_grouped adds the pairs up) */
    float* part_n;                                                      /* [NL][2][E] */
    float* total;  int32_t* ticket;                                     /* optional: weighted total [1], written by the workgroup that draws the last ticket; *ticket must be 0 at the first launch and is left 0 */
    int32_t NL, B, P, K, E;
    float   eps;
    int32_t invert_valid;  int32_t reserved;                            /* invert_valid: `valid` holds 1 = ignore (the model's query mask) */
} rt_head_loss_desc;
int rt_head_loss(const rt_head_loss_desc* d, rt_stream_t stream);

typedef struct rt_qenc_bwd_desc {
    const float *ga, *gb, *dqpos;                                       /* fp32 [N, E]; gb optional */
    const float *t2, *m2, *r2, *g5, *bet5;
    const float *t1, *m1, *r1, *g1, *bet1;
    const float *co, *cmean, *crstd, *gc, *betc;
    const float *kq, *qs, *vs, *qw;
    const void *Wf4T, *Wf0T, *WcT, *W1T, *W2T, *W3T;                    /* bf16 backward-data operands [K][N] */
    const uint32_t* seed_dev;
    void *dt2b, *dt1b, *dcob, *dk16, *dqs16, *dvs16;                    /* bf16 dy operands of the weight gradients */
    float *da, *dcat, *dc;                                              /* fp32 scratch / outputs: [N,E], [N,2E], [N,E] */
    float* dmem;                                                        /* fp32 [B*S, E], accumulated */
    float* dqembed;                                                     /* optional: fp32 [2, E], accumulated */
    float *part5, *part1, *partc;                                       /* [B][2][E] each */
    int32_t B, S, L, P, E;
    float   drop_p;
    uint32_t drop_seed;  int32_t reserved;
} rt_qenc_bwd_desc;
int rt_qenc_bwd(const rt_qenc_bwd_desc* d, rt_stream_t stream);
/* Measurement aid: the stage stamps (100 MHz wall clock) workgroup 0 of the last rt_qenc_fwd / rt_head_loss / rt_qenc_bwd launch left,
 * [3][24] values copied to HOST memory (synchronises with the device). */
int rt_qregion_trace(unsigned long long* host_out);

#ifdef __cplusplus
}
#endif
#endif /* REFTR_HIP_H */
