"""ctypes binding of libreftr_hip.so (include/reftr_hip.h) + tensor-level op wrappers.

PyTorch is used here only as the owner of device memory and of the HIP stream the kernels are enqueued
on (torch.cuda.current_stream()).  There is NO fallback: if the library is missing or a kernel reports
an error, a RuntimeError is raised.
"""
import contextlib
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_int64, c_uint32, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libreftr_hip_lab.so" if os.environ.get("REFTR_LAB", "0") == "1" else "libreftr_hip.so")   # _build.py
ABI_VERSION = 33

ACT_NONE, ACT_RELU, ACT_GELU, ACT_TANH = 0, 1, 2, 3
_c_float_p = POINTER(c_float)


class ConvGemmDesc(Structure):
    _fields_ = [
        ("src", c_void_p), ("wgt", c_void_p), ("out_bf16", c_void_p), ("out_f32", c_void_p),
        ("bias", c_void_p), ("res_f32", c_void_p), ("res_bf16", c_void_p), ("gate", c_void_p),
        ("preact", c_void_p),
        ("B", c_int32), ("SH", c_int32), ("SW", c_int32), ("SC", c_int32),
        ("DH", c_int32), ("DW", c_int32), ("N", c_int32),
        ("KH", c_int32), ("KW", c_int32), ("stride", c_int32), ("pad", c_int32),
        ("transposed", c_int32), ("act", c_int32),
        ("gate_scale", c_float), ("drop_p", c_float), ("drop_seed", c_uint32), ("tile_hint", c_int32),
        ("out_preact", c_void_p), ("dtanh", c_void_p), ("res_first", c_int32), ("seed_dev", c_void_p),
        ("drop_shift", c_int32), ("acc2_f32", c_void_p), ("dil", c_int32),
    ]


class ConvWgradDesc(Structure):
    _fields_ = [
        ("dy", c_void_p), ("x", c_void_p), ("dw", c_void_p), ("scale", c_void_p),
        ("B", c_int32), ("SH", c_int32), ("SW", c_int32), ("SC", c_int32),
        ("DH", c_int32), ("DW", c_int32), ("N", c_int32),
        ("KH", c_int32), ("KW", c_int32), ("stride", c_int32), ("pad", c_int32),
        ("msplit", c_int32), ("dbias", c_void_p), ("variant", c_int32),
        ("workspace", c_void_p), ("workspace_bytes", ctypes.c_int64), ("overwrite", c_int32), ("dil", c_int32),
        ("sqacc", c_void_p), ("g16", c_void_p),
    ]


class LayerNormDesc(Structure):
    _fields_ = [
        ("x", c_void_p), ("gamma", c_void_p), ("beta", c_void_p), ("y_f32", c_void_p), ("y_bf16", c_void_p),
        ("pos", c_void_p), ("ypos_bf16", c_void_p), ("mean", c_void_p), ("rstd", c_void_p),
        ("M", c_int32), ("D", c_int32), ("eps", c_float), ("act", c_int32),
        ("drop_p", c_float), ("drop_seed", c_uint32),
        ("grp_rows", c_int32), ("grp_stride", c_int32), ("grp_off", c_int32), ("seed_dev", c_void_p),
    ]


class LayerNormBwdDesc(Structure):
    _fields_ = [
        ("dy", c_void_p), ("dy2", c_void_p), ("x", c_void_p), ("gamma", c_void_p), ("beta", c_void_p),
        ("mean", c_void_p), ("rstd", c_void_p), ("dx_f32", c_void_p), ("dx_bf16", c_void_p),
        ("dgamma", c_void_p), ("dbeta", c_void_p),
        ("M", c_int32), ("D", c_int32), ("act", c_int32),
        ("drop_p", c_float), ("drop_seed", c_uint32), ("drop2_p", c_float), ("drop2_seed", c_uint32),
        ("grp_rows", c_int32), ("grp_stride", c_int32), ("grp_off", c_int32), ("seed_dev", c_void_p),
        ("partials", c_void_p), ("n_blocks_out", POINTER(c_int32)),
    ]


class LnPgJob(Structure):
    _fields_ = [("partials", c_void_p), ("dgamma", c_void_p), ("dbeta", c_void_p), ("n_blocks", c_int32), ("D", c_int32)]


class GroupNormDesc(Structure):
    _fields_ = [
        ("x", c_void_p), ("gamma", c_void_p), ("beta", c_void_p), ("stats", c_void_p),
        ("y_f32", c_void_p), ("y_bf16", c_void_p), ("pos", c_void_p), ("ypos_bf16", c_void_p),
        ("B", c_int32), ("HW", c_int32), ("C", c_int32), ("G", c_int32), ("eps", c_float),
        ("out_rows_per_img", c_int32), ("out_row_off", c_int32), ("partials", c_void_p), ("chunks", c_int32),
    ]


class GroupNormBwdDesc(Structure):
    _fields_ = [
        ("dy", c_void_p), ("dy2", c_void_p), ("x", c_void_p), ("gamma", c_void_p), ("stats", c_void_p),
        ("bstats", c_void_p), ("dx_f32", c_void_p), ("dx_bf16", c_void_p), ("dgamma", c_void_p), ("dbeta", c_void_p),
        ("B", c_int32), ("HW", c_int32), ("C", c_int32), ("G", c_int32), ("eps", c_float),
        ("out_rows_per_img", c_int32), ("out_row_off", c_int32),
    ]


class AttnDesc(Structure):
    _fields_ = [
        ("q", c_void_p), ("k", c_void_p), ("v", c_void_p), ("out", c_void_p), ("lse", c_void_p), ("kpm", c_void_p),
        ("B", c_int32), ("H", c_int32), ("Sq", c_int32), ("Sk", c_int32), ("dh", c_int32),
        ("ldq", c_int32), ("ldk", c_int32), ("ldv", c_int32), ("ldo", c_int32),
        ("scale", c_float), ("drop_p", c_float), ("drop_seed", c_uint32), ("seed_dev", c_void_p),
    ]


class AttnBwdDesc(Structure):
    _fields_ = [
        ("q", c_void_p), ("k", c_void_p), ("v", c_void_p), ("out", c_void_p), ("dout", c_void_p),
        ("lse", c_void_p), ("delta", c_void_p), ("kpm", c_void_p), ("dq", c_void_p), ("dk", c_void_p), ("dv", c_void_p),
        ("B", c_int32), ("H", c_int32), ("Sq", c_int32), ("Sk", c_int32), ("dh", c_int32),
        ("ldq", c_int32), ("ldk", c_int32), ("ldv", c_int32), ("ldo", c_int32),
        ("lddq", c_int32), ("lddk", c_int32), ("lddv", c_int32),
        ("scale", c_float), ("drop_p", c_float), ("drop_seed", c_uint32), ("seed_dev", c_void_p),
    ]


class MaskPosencDesc(Structure):
    _fields_ = [
        ("mask", c_void_p), ("kpm_out", c_void_p), ("pos_out", c_void_p), ("add_vec", c_void_p),
        ("B", c_int32), ("H", c_int32), ("W", c_int32), ("h", c_int32), ("w", c_int32), ("C", c_int32),
        ("kpm_stride", c_int32), ("kpm_off", c_int32), ("pos_rows_per_img", c_int32), ("pos_row_off", c_int32),
    ]


class BottleneckDesc(Structure):
    _fields_ = [
        ("x", c_void_p), ("w1", c_void_p), ("w2", c_void_p), ("w3", c_void_p), ("wd", c_void_p),
        ("b1", c_void_p), ("b2", c_void_p), ("b3", c_void_p), ("bd", c_void_p), ("out", c_void_p),
        ("B", c_int32), ("H", c_int32), ("W", c_int32), ("cin", c_int32), ("planes", c_int32), ("form", c_int32),
    ]


class RowsAddDesc(Structure):
    _fields_ = [
        ("a_f32", c_void_p), ("a_bf16", c_void_p), ("b_f32", c_void_p), ("out_f32", c_void_p), ("out_bf16", c_void_p),
        ("rows", c_int32), ("D", c_int32), ("alpha", c_float), ("accumulate", c_int32),
        ("a_grp_rows", c_int32), ("a_grp_stride", c_int32), ("a_grp_off", c_int32),
        ("b_grp_rows", c_int32), ("b_grp_stride", c_int32), ("b_grp_off", c_int32),
        ("o_grp_rows", c_int32), ("o_grp_stride", c_int32), ("o_grp_off", c_int32),
    ]


class BoxLossDesc(Structure):
    _fields_ = [
        ("logits", c_void_p), ("valid", c_void_p), ("targets", c_void_p), ("tgt_off", c_void_p),
        ("num_boxes", c_void_p), ("losses", c_void_p), ("total", c_void_p), ("dlogits", c_void_p),
        ("NL", c_int32), ("B", c_int32), ("P", c_int32), ("K", c_int32), ("w_bbox", c_float), ("w_giou", c_float),
        ("weights", c_void_p),
    ]


class MaskPostDesc(Structure):
    _fields_ = [
        ("pred", c_void_p), ("sizes", c_void_p), ("orig", c_void_p), ("origin_off", c_void_p),
        ("masks", c_void_p), ("masks_origin", c_void_p),
        ("B", c_int32), ("Q", c_int32), ("h", c_int32), ("w", c_int32), ("max_h", c_int32), ("max_w", c_int32),
        ("max_origin", c_int64), ("threshold", c_float),
    ]


class BoxPostDesc(Structure):
    _fields_ = [
        ("boxes", c_void_p), ("valid", c_void_p), ("sizes", c_void_p), ("out", c_void_p), ("counts", c_void_p),
        ("B", c_int32), ("P", c_int32), ("K", c_int32),
    ]


class CemDesc(Structure):
    _fields_ = [("hs", c_void_p), ("w3", c_void_p), ("b3", c_void_p), ("res", c_void_p), ("w2", c_void_p), ("b2", c_void_p),
                ("u", c_void_p), ("energy", c_void_p), ("stats", c_void_p), ("loss", c_void_p), ("g", c_void_p),
                ("dres", c_void_p), ("dhs", c_void_p), ("dw3", c_void_p), ("db3", c_void_p), ("dw2", c_void_p),
                ("B", c_int32), ("HW", c_int32), ("ld", c_int32), ("lddr", c_int32), ("E", c_int32), ("reserved", c_int32)]


class AdamWDesc(Structure):
    _fields_ = [
        ("p", c_void_p), ("g", c_void_p), ("m", c_void_p), ("v", c_void_p), ("n", c_int64),
        ("gnorm_sq", c_void_p), ("gnorm_out", c_void_p),
        ("grad_scale", c_float), ("max_norm", c_float), ("beta1", c_float), ("beta2", c_float), ("eps", c_float),
        ("step", c_int32), ("n_ranges", c_int32),
        ("range_begin", c_int64 * 8), ("range_end", c_int64 * 8), ("range_lr", c_float * 8), ("range_wd", c_float * 8),
        ("step_dev", c_void_p), ("active", c_void_p), ("lr_dev", c_void_p), ("span_begin", c_int64), ("span_end", c_int64),
        ("g16", c_void_p),
    ]


# name -> (restype, argtypes); every symbol include/reftr_hip.h declares must be listed here
# (tests/test_abi.py cross-checks this table against the header).
class SmallWgradJob(Structure):
    _fields_ = [("dy", c_void_p), ("x", c_void_p), ("dw", c_void_p), ("dbias", c_void_p),
                ("M", c_int32), ("N", c_int32), ("K", c_int32), ("overwrite", c_int32), ("sqacc", c_void_p), ("g16", c_void_p)]


class GnNhwcDesc(Structure):
    _fields_ = [("x", c_void_p), ("gamma", c_void_p), ("beta", c_void_p), ("stats", c_void_p), ("y_bf16", c_void_p),
                ("B", c_int32), ("HW", c_int32), ("C", c_int32), ("G", c_int32), ("ldx", c_int32), ("ldy", c_int32),
                ("act", c_int32), ("eps", c_float), ("partials", c_void_p), ("partial_blocks", c_int32)]


class GnNhwcBwdDesc(Structure):
    _fields_ = [("dy", c_void_p), ("x", c_void_p), ("gamma", c_void_p), ("beta", c_void_p), ("stats", c_void_p),
                ("bstats", c_void_p), ("dx_bf16", c_void_p), ("dgamma", c_void_p), ("dbeta", c_void_p),
                ("B", c_int32), ("HW", c_int32), ("C", c_int32), ("G", c_int32), ("ldx", c_int32), ("lddy", c_int32),
                ("lddx", c_int32), ("act", c_int32), ("eps", c_float)]


class UpsampleAddDesc(Structure):
    _fields_ = [("fpn", c_void_p), ("a_bf16", c_void_p), ("out_bf16", c_void_p),
                ("B", c_int32), ("H", c_int32), ("W", c_int32), ("h", c_int32), ("w", c_int32), ("C", c_int32),
                ("ldf", c_int32), ("lda", c_int32), ("ldo", c_int32)]


class UpsampleAddBwdDesc(Structure):
    _fields_ = [("dy", c_void_p), ("da", c_void_p), ("dy_bf16", c_void_p),
                ("B", c_int32), ("H", c_int32), ("W", c_int32), ("h", c_int32), ("w", c_int32), ("C", c_int32),
                ("lddy", c_int32), ("ldda", c_int32), ("lddyb", c_int32)]


class AttnMapDesc(Structure):
    _fields_ = [("q", c_void_p), ("k", c_void_p), ("mask", c_void_p), ("P", c_void_p), ("concat_bf16", c_void_p),
                ("B", c_int32), ("HW", c_int32), ("E", c_int32), ("nh", c_int32), ("ldk", c_int32),
                ("k_rows_per_img", c_int32), ("k_row_off", c_int32), ("ld_concat", c_int32), ("concat_col", c_int32),
                ("norm", c_float)]


class AttnMapBwdDesc(Structure):
    _fields_ = [("q", c_void_p), ("k", c_void_p), ("P", c_void_p), ("dconcat", c_void_p), ("dq", c_void_p), ("dk", c_void_p),
                ("B", c_int32), ("HW", c_int32), ("E", c_int32), ("nh", c_int32), ("ldk", c_int32),
                ("k_rows_per_img", c_int32), ("k_row_off", c_int32), ("ld_dconcat", c_int32), ("concat_col", c_int32),
                ("norm", c_float)]


class SegConcatDesc(Structure):
    _fields_ = [("src", c_void_p), ("mem", c_void_p), ("out_bf16", c_void_p),
                ("B", c_int32), ("HW", c_int32), ("E", c_int32), ("nh", c_int32), ("ldo", c_int32),
                ("mem_rows_per_img", c_int32), ("mem_row_off", c_int32), ("src_rows_per_img", c_int32), ("src_row_off", c_int32)]


class MaskLossDesc(Structure):
    _fields_ = [("pred", c_void_p), ("target", c_void_p), ("sums", c_void_p), ("losses", c_void_p),
                ("dpred", c_void_p), ("g_focal", c_void_p), ("g_dice", c_void_p),
                ("B", c_int32), ("h", c_int32), ("w", c_int32), ("Ht", c_int32), ("Wt", c_int32), ("ldp", c_int32),
                ("lddp", c_int32), ("inv_norm", c_float), ("gbuf", c_void_p)]


DEC_MAX_LAYERS = 8


class DecoderLayerFwd(Structure):
    _PTRS = ("Wv", "Wo", "Wq", "Wo2", "W1", "W2", "bv", "bo", "bq", "bo2", "b1", "b2", "g1", "be1", "g2", "be2", "g3", "be3",
             "k2", "v2", "o", "t1q16", "q2", "o2", "t2_16", "hdn", "t3_16", "u", "u2", "u3",
             "mean1", "rstd1", "mean2", "rstd2", "mean3", "rstd3", "lse2", "t3_f32")
    _SEEDS = ("seed_ad", "seed_d1", "seed_ad2", "seed_d2", "seed_dh", "seed_d3")
    _fields_ = [(n, c_void_p) for n in _PTRS] + [(n, c_uint32) for n in _SEEDS]


class DecoderLayerBwd(Structure):
    _PTRS = ("WT2", "WT1", "WTo2", "WTq", "WTo", "WTv", "g1", "g2", "g3", "u", "u2", "u3",
             "mean1", "rstd1", "mean2", "rstd2", "mean3", "rstd3", "hdn", "q2", "k2", "v2", "o2", "lse2", "dnorm",
             "du3b", "dhdn", "du2b", "dq2", "dub", "dv", "dk2", "dv2", "dk2p", "dv2p", "part1", "part2", "part3")
    _SEEDS = ("seed_ad", "seed_d1", "seed_ad2", "seed_d2", "seed_d3", "reserved")
    _fields_ = [(n, c_void_p) for n in _PTRS] + [(n, c_uint32) for n in _SEEDS]


class DecoderBwdDesc(Structure):
    _fields_ = [("layer", DecoderLayerBwd * DEC_MAX_LAYERS),
                ("dta", c_void_p), ("dqpos", c_void_p), ("kpm", c_void_p), ("handoff", c_void_p), ("seed_dev", c_void_p),
                ("n_layers", c_int32), ("M", c_int32), ("H", c_int32), ("S", c_int32), ("F", c_int32), ("ldkv", c_int32),
                ("drop_p", c_float), ("scale", c_float), ("gate_scale", c_float), ("ldkvp", c_int32)]


class DecoderFwdDesc(Structure):
    _fields_ = [("layer", DecoderLayerFwd * DEC_MAX_LAYERS),
                ("t32", c_void_p), ("t16", c_void_p), ("qpos", c_void_p), ("kpm", c_void_p), ("handoff", c_void_p),
                ("seed_dev", c_void_p),
                ("n_layers", c_int32), ("M", c_int32), ("H", c_int32), ("S", c_int32), ("F", c_int32), ("ldkv", c_int32),
                ("drop_p", c_float), ("eps", c_float), ("scale", c_float)]


class QencFwdDesc(Structure):
    _PTRS = ("mem16", "mem32", "ctx", "W1", "W2", "W3", "Wc", "Wf0", "Wf4", "b1", "b2", "b3", "bc", "bf0", "bf4",
             "gc", "betc", "g1", "bet1", "g5", "bet5", "qembed", "seed_dev",
             "cls16", "lang16", "kq", "qs", "vs", "qw", "c16", "co", "cmean", "crstd", "cat16",
             "t1", "m1", "r1", "a16", "t2", "m2", "r2", "tgt32", "tgt16", "qpos", "tgtq16")
    _fields_ = [(n, c_void_p) for n in _PTRS] + [("B", c_int32), ("S", c_int32), ("L", c_int32), ("P", c_int32), ("nq", c_int32),
                                                 ("E", c_int32), ("eps", c_float), ("drop_p", c_float), ("drop_seed", c_uint32),
                                                 ("reserved", c_int32)]


class HeadLossDesc(Structure):
    _PTRS = ("t3", "gn", "betn", "W0", "W1", "W2", "W0T", "W1T", "b0", "b1", "b2", "w2_f32",
             "valid", "targets", "tgt_off", "num_boxes", "weights",
             "hs16", "y1", "y2", "hmean", "hrstd", "logits", "losses", "dlogits", "dl16", "dy2", "dy1", "dhs", "dnorm", "db2_part", "part_n", "total", "ticket")
    _fields_ = [(n, c_void_p) for n in _PTRS] + [("NL", c_int32), ("B", c_int32), ("P", c_int32), ("K", c_int32), ("E", c_int32),
                                                 ("eps", c_float), ("invert_valid", c_int32), ("reserved", c_int32)]


class QencBwdDesc(Structure):
    _PTRS = ("ga", "gb", "dqpos", "t2", "m2", "r2", "g5", "bet5", "t1", "m1", "r1", "g1", "bet1", "co", "cmean", "crstd", "gc", "betc",
             "kq", "qs", "vs", "qw", "Wf4T", "Wf0T", "WcT", "W1T", "W2T", "W3T", "seed_dev",
             "dt2b", "dt1b", "dcob", "dk16", "dqs16", "dvs16", "da", "dcat", "dc", "dmem", "dqembed", "part5", "part1", "partc")
    _fields_ = [(n, c_void_p) for n in _PTRS] + [("B", c_int32), ("S", c_int32), ("L", c_int32), ("P", c_int32), ("E", c_int32),
                                                 ("drop_p", c_float), ("drop_seed", c_uint32), ("reserved", c_int32)]


_SIGNATURES = {
    "rt_abi_version": (c_int, []),
    "rt_device_arch": (c_int, [c_int, c_char_p, c_int]),
    "rt_conv_gemm": (c_int, [POINTER(ConvGemmDesc), c_void_p]),
    "rt_conv_gemm_grouped": (c_int, [c_void_p, c_int, c_void_p]),
    "rt_conv_wgrad": (c_int, [POINTER(ConvWgradDesc), c_void_p]),
    "rt_layernorm_fwd": (c_int, [POINTER(LayerNormDesc), c_void_p]),
    "rt_layernorm_bwd": (c_int, [POINTER(LayerNormBwdDesc), c_void_p]),
    "rt_groupnorm_fwd": (c_int, [POINTER(GroupNormDesc), c_void_p]),
    "rt_groupnorm_bwd": (c_int, [POINTER(GroupNormBwdDesc), c_void_p]),
    "rt_attn_fwd": (c_int, [POINTER(AttnDesc), c_void_p]),
    "rt_attn_bwd": (c_int, [POINTER(AttnBwdDesc), c_void_p]),
    "rt_img_pack": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "rt_stem_conv": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "rt_maxpool3x3s2": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "rt_stem_pool": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "rt_bottleneck_fwd": (c_int, [POINTER(BottleneckDesc), c_void_p]),
    "rt_weight_prep": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "rt_weight_prep_batched": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "rt_stem_weight_prep": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "rt_bn_fold": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_int, c_void_p]),
    "rt_mask_posenc": (c_int, [POINTER(MaskPosencDesc), c_void_p]),
    "rt_colsum": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_void_p]),
    "rt_rows_add": (c_int, [POINTER(RowsAddDesc), c_void_p]),
    "rt_bert_embed_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "rt_bert_embed_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "rt_roberta_pos_ids": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "rt_context_mask": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "rt_qenc_attn_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "rt_qenc_attn_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "rt_box_loss": (c_int, [POINTER(BoxLossDesc), c_void_p]),
    "rt_cem_fwd": (c_int, [POINTER(CemDesc), c_void_p]),
    "rt_cem_bwd": (c_int, [POINTER(CemDesc), c_void_p]),
    "rt_mask_postprocess": (c_int, [POINTER(MaskPostDesc), c_void_p]),
    "rt_box_postprocess": (c_int, [POINTER(BoxPostDesc), c_void_p]),
    "rt_small_dgrad": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "rt_pos_grad": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "rt_sqnorm": (c_int, [c_void_p, c_int64, c_void_p, c_void_p]),
    "rt_sqnorm_bf16": (c_int, [c_void_p, c_int64, c_void_p, c_void_p]),
    "rt_sqnorm_finish": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "rt_round_chunks": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "rt_adamw_flat": (c_int, [POINTER(AdamWDesc), c_void_p]),
    "rt_sgd_flat": (c_int, [POINTER(AdamWDesc), c_void_p]),
    "rt_zero_chunks": (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    "rt_counter_add": (c_int, [c_void_p, c_int32, c_void_p]),
    "rt_ln_param_grad_grouped": (c_int, [POINTER(LnPgJob), c_int, c_void_p]),
    "rt_conv_wgrad_grouped": (c_int, [POINTER(ConvWgradDesc), c_int, c_void_p, ctypes.c_int64, c_void_p]),
    "rt_small_wgrad_grouped": (c_int, [POINTER(SmallWgradJob), c_int, c_void_p]),
    "rt_resample_u8": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "rt_img_collate_norm": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, POINTER(c_float), POINTER(c_float), c_void_p]),
    "rt_gn_nhwc_fwd": (c_int, [POINTER(GnNhwcDesc), c_void_p]),
    "rt_gn_nhwc_bwd": (c_int, [POINTER(GnNhwcBwdDesc), c_void_p]),
    "rt_upsample_add": (c_int, [POINTER(UpsampleAddDesc), c_void_p]),
    "rt_upsample_add_bwd": (c_int, [POINTER(UpsampleAddBwdDesc), c_void_p]),
    "rt_attn_map_fwd": (c_int, [POINTER(AttnMapDesc), c_void_p]),
    "rt_attn_map_bwd": (c_int, [POINTER(AttnMapBwdDesc), c_void_p]),
    "rt_seg_concat": (c_int, [POINTER(SegConcatDesc), c_void_p]),
    "rt_mask_loss": (c_int, [POINTER(MaskLossDesc), c_void_p]),
    "rt_comm_unique_id": (c_int, [c_void_p]),
    "rt_comm_init": (c_int, [c_void_p, c_int, c_int, POINTER(c_void_p)]),
    "rt_comm_allreduce": (c_int, [c_void_p, POINTER(c_void_p), POINTER(ctypes.c_int64), c_int, c_int, c_void_p]),
    "rt_comm_destroy": (c_int, [c_void_p]),
    "rt_decoder_fwd": (c_int, [POINTER(DecoderFwdDesc), c_void_p]),
    "rt_decoder_bwd": (c_int, [POINTER(DecoderBwdDesc), c_void_p]),
    "rt_decoder_trace": (c_int, [c_void_p]),
    "rt_decoder_supported": (c_int, [c_int]),
    "rt_decoder_set_spin": (c_int, [c_int]),
    "rt_counter_add_if_zero": (c_int, [c_void_p, c_int32, c_void_p, c_int, c_void_p]),
    "rt_stamp": (c_int, [c_void_p, c_int, c_void_p]),
    "rt_finish_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "rt_finish_stats": (c_int, [c_void_p, c_void_p]),
    "rt_adamw_mat": (c_int, [POINTER(AdamWDesc), c_void_p, c_int, c_int, c_void_p]),
    "rt_adamw_chunks": (c_int, [POINTER(AdamWDesc), c_void_p, c_int, c_void_p]),
    "rt_qenc_fwd": (c_int, [POINTER(QencFwdDesc), c_void_p]),
    "rt_head_loss": (c_int, [POINTER(HeadLossDesc), c_void_p]),
    "rt_qenc_bwd": (c_int, [POINTER(QencBwdDesc), c_void_p]),
    "rt_qregion_trace": (c_int, [c_void_p]),
}

_lib = None
LAB_LIB_PATH = os.path.join(_HERE, "libreftr_hip_lab.so")
# rt_conv_gemm tile hints the PRODUCT library instantiates (what its heuristics can choose; csrc/rt_gemm.hip); every other measured
# variant lives in the lab library only
PRODUCT_TILE_HINTS = frozenset({0, 21, 31, 33, 51, 233, 252, 262, 281, 285})


def _load(path):
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} not found: the HIP kernel library is not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc; no CPU fallback exists).")
    L = ctypes.CDLL(path)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    v = L.rt_abi_version()
    if v != ABI_VERSION:
        raise RuntimeError(f"{os.path.basename(path)} ABI {v} != binding ABI {ABI_VERSION}; rebuild")
    return L


def lib():
    """Load libreftr_hip.so (once).  Fails loudly when it has not been built."""
    global _lib
    if _lib is None:
        _lib = _load(LIB_PATH)
    return _lib


_lab_lib = None


@contextlib.contextmanager
def lab_library():
    """Route this module's calls to the LAB build (libreftr_hip_lab.so, the same sources with -DRT_LAB) inside the block: tests and
    sweeps of the rt_conv_gemm variants the product library does not instantiate.  Never used by the product path."""
    global _lib, _lab_lib
    if _lab_lib is None:
        _lab_lib = _load(LAB_LIB_PATH)
    prev = lib()
    _lib = _lab_lib
    try:
        yield
    finally:
        _lib = prev


def exported_symbols():
    return sorted(_SIGNATURES)


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed with code {rc} "
                           f"({'RT_ERR' if rc < 0 else 'hipError'})")


def _p(t):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


_TIMER = None
_SEED_DEV = None


def set_seed_dev(t):
    """Device uint32/int32 word mixed into every dropout seed (lets a captured hipGraph draw fresh masks per replay)."""
    global _SEED_DEV
    _SEED_DEV = t


def _seedp(drop_p):
    return _p(_SEED_DEV) if (drop_p > 0 and _SEED_DEV is not None) else None



def set_launch_timer(records):
    """bench.py instrumentation: when `records` is a list, every rt_conv_gemm / rt_conv_wgrad launch is
    bracketed by HIP events on the launch stream and appended as {kind, flops, start, end}."""
    global _TIMER
    _TIMER = records


def _algo_bytes(tag):
    """Compulsory HBM bytes of one GEMM-family launch: each operand read once (bf16), the result written once
    (bf16 activations / fp32 weight gradients)."""
    kind, B, SH, SW, SC, DH, DW, N, KH, KW = tag[:10]
    M = B * DH * DW
    if kind == "W":
        return 2.0 * (M * N + B * SH * SW * SC) + 4.0 * N * KH * KW * SC
    return 2.0 * (B * SH * SW * SC + N * KH * KW * SC + M * N)


def _timed(kind, flops, fn, tag=None, nbytes=None, epi_bytes=0.0):
    if _TIMER is None:
        return fn()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    r = fn()
    e1.record()
    _TIMER.append({"kind": kind, "flops": float(flops), "start": e0, "end": e1, "tag": tag,
                   "bytes": nbytes if nbytes is not None else (_algo_bytes(tag) if tag is not None else 0.0),
                   "epi_bytes": float(epi_bytes)})
    return r


def as_u8(t):
    """A mask as the uint8 bytes the kernels read: bool tensors are re-typed in place (same bytes, no launch -- `.to(torch.uint8)` is a
    converting copy: 12 us for the [8, 640, 640] padding mask at the head of every step), everything else is converted."""
    t = t.contiguous()
    if t.dtype == torch.bool:
        return t.view(torch.uint8)
    return t if t.dtype == torch.uint8 else t.to(torch.uint8)


def _req(t, dtype, name):
    if t is None:
        return
    if not t.is_cuda:
        raise RuntimeError(f"{name}: HIP kernels need a device tensor (got {t.device}); no CPU path exists")
    if t.dtype != dtype:
        raise RuntimeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise RuntimeError(f"{name}: tensor must be contiguous")


# --------------------------------------------------------------------------------------------
# op wrappers
# --------------------------------------------------------------------------------------------
def conv_gemm(src, wgt, *, geom, bias=None, res_f32=None, res_bf16=None, gate=None, gate_scale=1.0,
              preact=None, dtanh=None, res_first=False, act=ACT_NONE, drop_p=0.0, drop_seed=0, transposed=False,
              out_bf16=True, out_f32=False, out_preact=False, tile_hint=0, group=None, drop_shift=0, acc2_f32=None, dil=1):
    """out[B,DH,DW,N] = epilogue(implicit_gemm(src[B,SH,SW,SC], wgt[N,KH,KW,SC])).

    geom = (B, SH, SW, SC, DH, DW, N, KH, KW, stride, pad).  Returns (out_bf16 | None, out_f32 | None)
    (+ the bf16 pre-activation as a third value when out_preact=True).  out_bf16 / out_f32 may be True (allocate
    [M, N]) or an existing tensor to write into (in-place accumulation when it is also `res_f32`).
    """
    B, SH, SW, SC, DH, DW, N, KH, KW, stride, pad = geom
    _req(src, torch.bfloat16, "src"); _req(wgt, torch.bfloat16, "wgt")
    _req(bias, torch.float32, "bias"); _req(res_f32, torch.float32, "res_f32")
    _req(res_bf16, torch.bfloat16, "res_bf16"); _req(gate, torch.bfloat16, "gate")
    _req(preact, torch.bfloat16, "preact")
    assert src.numel() == B * SH * SW * SC, (src.shape, geom)
    assert wgt.numel() == N * KH * KW * SC, (wgt.shape, geom)
    M = B * DH * DW
    _req(dtanh, torch.bfloat16, "dtanh")
    ob = out_bf16 if torch.is_tensor(out_bf16) else (torch.empty((M, N), dtype=torch.bfloat16, device=src.device) if out_bf16 else None)
    of = out_f32 if torch.is_tensor(out_f32) else (torch.empty((M, N), dtype=torch.float32, device=src.device) if out_f32 else None)
    _req(ob, torch.bfloat16, "out_bf16"); _req(of, torch.float32, "out_f32"); _req(acc2_f32, torch.float32, "acc2_f32")
    assert acc2_f32 is None or acc2_f32.numel() == M * N
    assert (ob is None or ob.numel() == M * N) and (of is None or of.numel() == M * N)
    d = ConvGemmDesc(_p(src), _p(wgt), _p(ob), _p(of), _p(bias), _p(res_f32), _p(res_bf16), _p(gate),
                     _p(preact), B, SH, SW, SC, DH, DW, N, KH, KW, stride, pad,
                     1 if transposed else 0, act, gate_scale, drop_p, drop_seed & 0xFFFFFFFF, tile_hint, None, _p(dtanh), 1 if res_first else 0, _seedp(drop_p),
                     drop_shift, _p(acc2_f32), int(dil))
    op = None
    if out_preact:
        op = torch.empty((M, N), dtype=torch.bfloat16, device=src.device)
        d.out_preact = _p(op)
    if transposed:     # algorithmic FLOPs of backward-data = those of the forward conv it differentiates
        flops = 2.0 * B * SH * SW * SC * N * KH * KW
    else:
        flops = 2.0 * M * N * KH * KW * SC
    # bytes of the elementwise operations fused into the epilogue (NOT part of the operand-only "algorithmic" figure): bf16
    # residual / gate / GELU-pre-activation / tanh reads, fp32 residual read, read-modify-write of the second destination, and
    # every output beyond the first bf16 one
    epi = M * N * (2.0 * sum(t is not None for t in (res_bf16, gate, preact, dtanh)) + 4.0 * (res_f32 is not None)
                   + 8.0 * (acc2_f32 is not None) + (4.0 if of is not None else 0.0) + (2.0 if ob is not None else 0.0) - 2.0
                   + (2.0 if out_preact else 0.0))
    if group is not None:      # queued: GemmGroup.run() launches every queued product at once
        group.add(d, flops, ("T" if transposed else "F",) + tuple(geom), (src, wgt, ob, of, op, bias, res_f32, res_bf16, gate, preact, dtanh), epi)
    else:
        _timed("conv_gemm", flops, lambda: _check(lib().rt_conv_gemm(ctypes.byref(d), _stream()), "rt_conv_gemm"),
               tag=("T" if transposed else "F",) + tuple(geom), epi_bytes=epi)
    if out_preact:
        return ob, of, op
    return ob, of


class GemmGroup:
    """Independent GEMMs queued with `conv_gemm(..., group=g)` / `linear(..., group=g)` (outputs are allocated at queue time)
    and launched together by `run()` (rt_conv_gemm_grouped: one launch when the products allow it)."""

    def __init__(self):
        self.descs, self.keep, self.flops, self.nbytes, self.epi = [], [], 0.0, 0.0, 0.0

    def add(self, d, flops, tag, keep, epi=0.0):
        self.descs.append(d); self.keep.append(keep); self.flops += flops; self.nbytes += _algo_bytes(tag); self.epi += epi

    def run(self):
        if not self.descs:
            return
        n = len(self.descs)
        for i in range(0, n, 12):
            chunk = self.descs[i:i + 12]
            arr = (ConvGemmDesc * len(chunk))(*chunk)
            share = len(chunk) / n
            _timed("conv_gemm", self.flops * share,
                   lambda arr=arr, m=len(chunk): _check(lib().rt_conv_gemm_grouped(arr, m, _stream()), "rt_conv_gemm_grouped"),
                   nbytes=self.nbytes * share, epi_bytes=self.epi * share)
        self.descs, self.keep, self.flops, self.nbytes, self.epi = [], [], 0.0, 0.0, 0.0


def linear(x, w, bias=None, **kw):
    """y[M,N] = epilogue(x[M,K] @ w[N,K]^T): a 1x1 'conv' over M rows."""
    M, K = x.shape
    N = w.shape[0]
    return conv_gemm(x, w, geom=(M, 1, 1, K, 1, 1, N, 1, 1, 1, 0), bias=bias, **kw)


_WGRAD_WS = {}
WGRAD_WS_BYTES = 64 << 20


def _wgrad_workspace(dev):
    """Per-(device, stream) scratch for rt_conv_wgrad's split partials: launches on concurrent streams must not
    share it, launches on one stream reuse it in order."""
    key = (dev.index, torch.cuda.current_stream().cuda_stream)
    ws = _WGRAD_WS.get(key)
    if ws is None:
        ws = _WGRAD_WS[key] = torch.empty(WGRAD_WS_BYTES // 4, dtype=torch.float32, device=dev)
    return ws


# Gradient-norm accumulators: {data_ptr of a registered weight-gradient matrix: its model's slot buffer} (ParamStore registers its
# overwritable matrices here).  Every weight-gradient launch into such a matrix hands the slots to the library (`sqacc`), which adds
# |dw after|^2 - |dw before|^2 -- the clip norm then needs no pass over the gradient buffer (rt_sqnorm_finish).
SQ_SLOTS, SQ_STRIDE = 256, 32
_SQACC_MAP = {}


def _owned(table, dw):
    """Entry of a {data_ptr: (weakref to the owning ParamStore, value)} map, valid only while the owner is alive (its gradient
    buffer still occupies that address: nothing else can); entries of dead owners are dropped."""
    ent = table.get(dw.data_ptr())
    if ent is None:
        return None
    if ent[0]() is None:
        del table[dw.data_ptr()]
        return None
    return ent[1]


def _sqacc(dw):
    t = _owned(_SQACC_MAP, dw)
    return t.data_ptr() if t is not None else None


# bf16 exchange twins (data parallel): {data_ptr of a registered weight-gradient matrix: data_ptr of its bf16 twin} -- the launches
# that write the matrix write the rounded copy as well (rt_conv_wgrad_desc.g16); reftr_amd.parallel fills the map
_G16_MAP = {}


def _g16(dw):
    return _owned(_G16_MAP, dw)


def round_chunks(base, twin, table, n):
    """twin[i] = bf16(base[i]) over n chunks {offset, count} of a static device table (rt_round_chunks)."""
    _req(base, torch.float32, "base"); _req(twin, torch.bfloat16, "twin"); _req(table, torch.int64, "table")
    if n > 0:
        _check(lib().rt_round_chunks(_p(base), _p(twin), _p(table), int(n), _stream()), "rt_round_chunks")


def sqnorm_finish(flat_g, table, nchunks, slots, out, extra=None):
    """out[0] = sum of the accumulator slots + |the chunks of flat_g listed in `table`|^2 (+ extra[0])."""
    _req(flat_g, torch.float32, "flat_g"); _req(slots, torch.float32, "slots"); _req(out, torch.float32, "out")
    assert slots.numel() == SQ_SLOTS * SQ_STRIDE
    _check(lib().rt_sqnorm_finish(_p(flat_g), _p(table), int(nchunks), _p(slots), _p(extra), _p(out), _stream()), "rt_sqnorm_finish")


def conv_wgrad(dy, x, dw, *, geom, scale=None, msplit=0, dbias=None, variant=0, workspace=True, overwrite=False, dil=1):
    """dw[N,KH,KW,SC] (fp32) += (overwrite: =) scale[n] * sum_m dy[m,n] * gather(x)[m,(kh,kw,c)]."""
    B, SH, SW, SC, DH, DW, N, KH, KW, stride, pad = geom
    _req(dy, torch.bfloat16, "dy"); _req(x, torch.bfloat16, "x"); _req(dw, torch.float32, "dw")
    _req(scale, torch.float32, "scale")
    assert dy.numel() == B * DH * DW * N and x.numel() == B * SH * SW * SC and dw.numel() == N * KH * KW * SC
    _req(dbias, torch.float32, "dbias")
    ws = _wgrad_workspace(dy.device) if workspace else None
    d = ConvWgradDesc(_p(dy), _p(x), _p(dw), _p(scale), B, SH, SW, SC, DH, DW, N, KH, KW, stride, pad, msplit, _p(dbias), variant,
                      _p(ws), WGRAD_WS_BYTES if ws is not None else 0, int(bool(overwrite)), int(dil), _sqacc(dw), _g16(dw))
    _timed("conv_wgrad", 2.0 * B * DH * DW * N * KH * KW * SC,
           lambda: _check(lib().rt_conv_wgrad(ctypes.byref(d), _stream()), "rt_conv_wgrad"), tag=("W",) + tuple(geom))
    return dw


def linear_wgrad(dy, x, dw, **kw):
    M, N = dy.shape
    K = x.shape[1]
    return conv_wgrad(dy, x, dw, geom=(M, 1, 1, K, 1, 1, N, 1, 1, 1, 0), **kw)


# --------------------------------------------------------------------------------------------
# norms
# --------------------------------------------------------------------------------------------
def _new(shape, dtype, like):
    return torch.empty(shape, dtype=dtype, device=like.device)


def layernorm_fwd(x, gamma, beta, eps=1e-5, *, act=ACT_NONE, drop_p=0.0, drop_seed=0, pos=None,
                  y_f32=None, y_bf16=None, ypos_bf16=None, want_f32=True, want_bf16=True,
                  rowmap=(0, 0, 0), save_stats=True):
    """y = [drop][relu](LN(x)); returns (y_f32, y_bf16, ypos_bf16, mean, rstd).  Output buffers may be passed
    in (row-mapped writes into a larger sequence buffer) or are allocated [M, D]."""
    M, D = x.shape
    _req(x, torch.float32, "x"); _req(gamma, torch.float32, "gamma"); _req(beta, torch.float32, "beta")
    _req(pos, torch.float32, "pos")
    if y_f32 is None and want_f32:
        y_f32 = _new((M, D), torch.float32, x)
    if y_bf16 is None and want_bf16:
        y_bf16 = _new((M, D), torch.bfloat16, x)
    if pos is not None and ypos_bf16 is None:
        ypos_bf16 = _new((M, D), torch.bfloat16, x)
    mean = _new((M,), torch.float32, x) if save_stats else None
    rstd = _new((M,), torch.float32, x) if save_stats else None
    d = LayerNormDesc(_p(x), _p(gamma), _p(beta), _p(y_f32), _p(y_bf16), _p(pos), _p(ypos_bf16), _p(mean), _p(rstd),
                      M, D, eps, act, drop_p, drop_seed & 0xFFFFFFFF, *rowmap, _seedp(drop_p))
    _check(lib().rt_layernorm_fwd(ctypes.byref(d), _stream()), "rt_layernorm_fwd")
    return y_f32, y_bf16, ypos_bf16, mean, rstd


class LnGradBatch:
    """LayerNorm parameter gradients of many rt_layernorm_bwd launches, reduced together (rt_ln_param_grad_grouped)."""

    def __init__(self):
        self.jobs, self.keep = [], []

    def run(self):
        if not self.jobs:
            return
        arr = (LnPgJob * len(self.jobs))(*self.jobs)
        _check(lib().rt_ln_param_grad_grouped(arr, len(self.jobs), _stream()), "rt_ln_param_grad_grouped")
        self.jobs, self.keep = [], []


def layernorm_bwd(dy, x, gamma, beta, mean, rstd, dgamma, dbeta, *, dy2=None, act=ACT_NONE, drop_p=0.0,
                  drop_seed=0, drop2_p=0.0, drop2_seed=0, rowmap=(0, 0, 0), want_f32=True, want_bf16=True, pg_batch=None):
    """Returns (dx_f32 [M,D], dx_bf16 [M,D] = bf16(dx * dropout2-mask)); dgamma/dbeta accumulated in place, or -- with
    `pg_batch` (LnGradBatch) -- queued as per-workgroup partial sums that pg_batch.run() reduces later."""
    M, D = x.shape
    _req(dy, torch.float32, "dy"); _req(dy2, torch.float32, "dy2"); _req(x, torch.float32, "x")
    dx_f32 = _new((M, D), torch.float32, x) if want_f32 else None
    dx_bf16 = _new((M, D), torch.bfloat16, x) if want_bf16 else None
    part, nb = None, c_int32(0)
    if pg_batch is not None and (dgamma is not None or dbeta is not None):
        part = _new((min((M + 3) // 4, 1024), 2, D), torch.float32, x)
    d = LayerNormBwdDesc(_p(dy), _p(dy2), _p(x), _p(gamma), _p(beta), _p(mean), _p(rstd), _p(dx_f32), _p(dx_bf16),
                         _p(dgamma), _p(dbeta), M, D, act, drop_p, drop_seed & 0xFFFFFFFF, drop2_p,
                         drop2_seed & 0xFFFFFFFF, *rowmap, _seedp(max(drop_p, drop2_p)), _p(part), ctypes.pointer(nb))
    _check(lib().rt_layernorm_bwd(ctypes.byref(d), _stream()), "rt_layernorm_bwd")
    if part is not None:
        pg_batch.jobs.append(LnPgJob(_p(part), _p(dgamma), _p(dbeta), nb.value, D))
        pg_batch.keep.append(part)
    return dx_f32, dx_bf16


def groupnorm_fwd(x, gamma, beta, G, eps, *, y_f32=None, y_bf16=None, pos=None, ypos_bf16=None,
                  rows_per_img=None, row_off=0):
    """x fp32 [B, HW, C]; outputs are sequence buffers [B*rows_per_img, C] written at row_off + pixel."""
    B, HW, C = x.shape
    _req(x, torch.float32, "x")
    rows_per_img = HW if rows_per_img is None else rows_per_img
    stats = _new((B, G, 2), torch.float32, x)
    chunks = min(64, (HW + 7) // 8)
    partials = _new((B, chunks, G, 2), torch.float32, x)
    d = GroupNormDesc(_p(x), _p(gamma), _p(beta), _p(stats), _p(y_f32), _p(y_bf16), _p(pos), _p(ypos_bf16),
                      B, HW, C, G, eps, rows_per_img, row_off, _p(partials), chunks)
    _check(lib().rt_groupnorm_fwd(ctypes.byref(d), _stream()), "rt_groupnorm_fwd")
    return stats


def groupnorm_bwd(dy, x, gamma, stats, dgamma, dbeta, G, eps, *, dy2=None, rows_per_img=None, row_off=0,
                  want_f32=False, want_bf16=True):
    B, HW, C = x.shape
    rows_per_img = HW if rows_per_img is None else rows_per_img
    bstats = _new((B, G, 2), torch.float32, x)
    dx_f32 = _new((B, HW, C), torch.float32, x) if want_f32 else None
    dx_bf16 = _new((B, HW, C), torch.bfloat16, x) if want_bf16 else None
    d = GroupNormBwdDesc(_p(dy), _p(dy2), _p(x), _p(gamma), _p(stats), _p(bstats), _p(dx_f32), _p(dx_bf16),
                         _p(dgamma), _p(dbeta), B, HW, C, G, eps, rows_per_img, row_off)
    _check(lib().rt_groupnorm_bwd(ctypes.byref(d), _stream()), "rt_groupnorm_bwd")
    return dx_f32, dx_bf16


# --------------------------------------------------------------------------------------------
# attention
# --------------------------------------------------------------------------------------------
def _ld(t):
    assert t.dim() == 2 and t.stride(1) == 1, "attention operands are 2-D row views with unit inner stride"
    return t.stride(0)


def attn_fwd(q, k, v, kpm, *, B, H, Sq, Sk, dh, scale, drop_p=0.0, drop_seed=0, out=None):
    """q [B*Sq, >=H*dh], k/v [B*Sk, >=H*dh] bf16 row views (head h at columns h*dh..); kpm uint8 [B, Sk] or None.
    Returns (out bf16 [B*Sq, H*dh], lse fp32 [B, H, Sq])."""
    if out is None:
        out = _new((B * Sq, H * dh), torch.bfloat16, q)
    lse = _new((B, H, Sq), torch.float32, q)
    d = AttnDesc(_p(q), _p(k), _p(v), _p(out), _p(lse), _p(kpm), B, H, Sq, Sk, dh, _ld(q), _ld(k), _ld(v), _ld(out),
                 scale, drop_p, drop_seed & 0xFFFFFFFF, _seedp(drop_p))
    _check(lib().rt_attn_fwd(ctypes.byref(d), _stream()), "rt_attn_fwd")
    return out, lse


def attn_bwd(q, k, v, out, dout, lse, kpm, *, B, H, Sq, Sk, dh, scale, drop_p=0.0, drop_seed=0,
             dq=None, dk=None, dv=None):
    """Returns (dq, dk, dv) bf16; output row views may be supplied (e.g. halves of one packed buffer)."""
    if dq is None:
        dq = _new((B * Sq, H * dh), torch.bfloat16, q)
    if dk is None:
        dk = _new((B * Sk, H * dh), torch.bfloat16, q)
    if dv is None:
        dv = _new((B * Sk, H * dh), torch.bfloat16, q)
    delta = _new((B, H, Sq), torch.float32, q)
    d = AttnBwdDesc(_p(q), _p(k), _p(v), _p(out), _p(dout), _p(lse), _p(delta), _p(kpm), _p(dq), _p(dk), _p(dv),
                    B, H, Sq, Sk, dh, _ld(q), _ld(k), _ld(v), _ld(out), _ld(dq), _ld(dk), _ld(dv),
                    scale, drop_p, drop_seed & 0xFFFFFFFF, _seedp(drop_p))
    assert _ld(dout) == _ld(out)
    _check(lib().rt_attn_bwd(ctypes.byref(d), _stream()), "rt_attn_bwd")
    return dq, dk, dv


DEC_HANDOFF_BYTES = 256 + 16 * 256 * 36 + 16 * 2048 * 4


def decoder_handoff(dev):
    """A hand-off buffer for rt_decoder_fwd / rt_decoder_bwd (zeroed once; word 0 = launch epoch, word 1 = failure flag).  Its owner
    (one model) keeps it for all its launches: they are ordered on the model's stream."""
    return torch.zeros(DEC_HANDOFF_BYTES // 4, dtype=torch.int32, device=dev)


def decoder_fwd(layers, t32, t16, qpos, kpm, *, H, S, F, drop_p, scale, handoff, eps=1e-5):
    """The decoder stack of one-query-per-image inputs as one cooperative launch (rt_decoder_fwd).  `layers`: one dict per
    layer, keys = DecoderLayerFwd._PTRS (tensors) + DecoderLayerFwd._SEEDS (ints); every output tensor is allocated by the
    caller (they are the launched chain's saved tensors).  Returns the hand-off buffer's header [epoch, failure flag]."""
    M = t32.shape[0]
    d = DecoderFwdDesc()
    assert 1 <= len(layers) <= DEC_MAX_LAYERS
    for i, lay in enumerate(layers):
        L = d.layer[i]
        for n in DecoderLayerFwd._PTRS:
            setattr(L, n, _p(lay[n]))
        for n in DecoderLayerFwd._SEEDS:
            setattr(L, n, lay[n] & 0xFFFFFFFF)
    d.t32, d.t16, d.qpos, d.kpm, d.handoff = _p(t32), _p(t16), _p(qpos), _p(kpm), _p(handoff)
    d.seed_dev = _seedp(drop_p)
    d.n_layers, d.M, d.H, d.S, d.F, d.ldkv = len(layers), M, H, S, F, _ld(layers[0]["k2"])
    d.drop_p, d.eps, d.scale = drop_p, eps, scale
    _check(lib().rt_decoder_fwd(ctypes.byref(d), _stream()), "rt_decoder_fwd")
    return handoff[:2]


def decoder_bwd(layers, dta, dqpos, kpm, *, H, S, F, drop_p, scale, gate_scale, handoff, ldkvp=0):
    """Backward of the cooperative decoder stack (rt_decoder_bwd).  `layers`: one dict per layer (first to last), keys =
    DecoderLayerBwd._PTRS (tensors, all allocated by the caller) + the dropout seeds of the forward."""
    M = dta.shape[0]
    d = DecoderBwdDesc()
    assert 1 <= len(layers) <= DEC_MAX_LAYERS
    for i, lay in enumerate(layers):
        L = d.layer[i]
        for n in DecoderLayerBwd._PTRS:
            setattr(L, n, _p(lay.get(n)))
        for n in DecoderLayerBwd._SEEDS[:-1]:
            setattr(L, n, lay[n] & 0xFFFFFFFF)
    d.dta, d.dqpos, d.kpm, d.handoff, d.seed_dev = _p(dta), _p(dqpos), _p(kpm), _p(handoff), _seedp(drop_p)
    d.n_layers, d.M, d.H, d.S, d.F, d.ldkv = len(layers), M, H, S, F, _ld(layers[0]["k2"])
    d.drop_p, d.scale, d.gate_scale, d.ldkvp = drop_p, scale, gate_scale, ldkvp
    _check(lib().rt_decoder_bwd(ctypes.byref(d), _stream()), "rt_decoder_bwd")
    return handoff[:2]


def decoder_trace(readback=True):
    """REFTR_DEC_TRACE=1: (workgroup 0 stamps, last workgroup stamps) of the last rt_decoder_fwd launch, 100 MHz ticks."""
    buf = (c_uint32 * 1024)()
    _check(lib().rt_decoder_trace(ctypes.cast(buf, c_void_p) if readback else None), "rt_decoder_trace")
    if not readback:
        return None
    a = list(buf)
    return a[:512], a[512:]


# --------------------------------------------------------------------------------------------
# backbone-side
# --------------------------------------------------------------------------------------------
def stem_geometry(H, W):
    """conv1 7x7/2 pad 3 output size and the padded NHWC4 input size rt_stem_conv needs."""
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    Hp = max(H + 6, 2 * Ho + 5)
    Wp = max(W + 6, 32 * ((Wo + 15) // 16) + 6)
    return Ho, Wo, Hp, Wp


def img_pack(img):
    B, C, H, W = img.shape
    assert C == 3
    _req(img, torch.float32, "img")
    _, _, Hp, Wp = stem_geometry(H, W)
    out = _new((B, Hp, Wp, 4), torch.bfloat16, img)
    _check(lib().rt_img_pack(_p(img), _p(out), B, H, W, Hp, Wp, _stream()), "rt_img_pack")
    return out


def stem_conv(xp, w, bias, Ho, Wo):
    B, Hp, Wp, _ = xp.shape
    out = _new((B, Ho, Wo, 64), torch.bfloat16, xp)
    _check(lib().rt_stem_conv(_p(xp), _p(w), _p(bias), _p(out), B, Hp, Wp, Ho, Wo, _stream()), "rt_stem_conv")
    return out


def maxpool3x3s2(x):
    B, H, W, C = x.shape
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    y = _new((B, Ho, Wo, C), torch.bfloat16, x)
    _check(lib().rt_maxpool3x3s2(_p(x), _p(y), B, H, W, C, Ho, Wo, _stream()), "rt_maxpool3x3s2")
    return y


def stem_pool(xp, w, bias, Ho, Wo):
    """conv1 7x7/2 + FrozenBN + ReLU + MaxPool2d(3, 2, 1) in one launch: the stem output is never written."""
    B, Hp, Wp, _ = xp.shape
    out = _new((B, (Ho - 1) // 2 + 1, (Wo - 1) // 2 + 1, 64), torch.bfloat16, xp)
    _check(lib().rt_stem_pool(_p(xp), _p(w), _p(bias), _p(out), B, Hp, Wp, Ho, Wo, _stream()), "rt_stem_pool")
    return out


def bottleneck_fwd(x, w1, b1, w2, b2, w3, b3, wd=None, bd=None, out=None, form=0):
    """One frozen stride-1 layer1 bottleneck (planes 64) in one launch: x bf16 [B,H,W,cin] -> bf16 [B,H,W,256]."""
    _req(x, torch.bfloat16, "x")
    B, Hh, Ww, cin = x.shape
    if out is None:
        out = _new((B, Hh, Ww, 256), torch.bfloat16, x)
    d = BottleneckDesc(_p(x), _p(w1), _p(w2), _p(w3), _p(wd), _p(b1), _p(b2), _p(b3), _p(bd), _p(out), B, Hh, Ww, cin, w1.shape[0], form)
    # a member of the GEMM family of bench.py's roofline: the algorithmic FLOPs of the convolutions it replaces (the conv1 halo
    # recomputation is not counted), compulsory bytes = block input + output + weights once
    M = B * Hh * Ww
    macs = cin * 64 + 576 * 64 + 64 * 256 + (cin * 256 if wd is not None else 0)
    nbytes = 2.0 * (M * (cin + 256) + macs)
    _timed("bottleneck_fwd", 2.0 * M * macs, lambda: _check(lib().rt_bottleneck_fwd(ctypes.byref(d), _stream()), "rt_bottleneck_fwd"),
           tag=("BNK", B, Hh, Ww, cin, Hh, Ww, 256, 3, 3, 1, 1), nbytes=nbytes)
    return out


def weight_prep(src, N, T, C, *, scale=None, dst=None, dst_t=None):
    """fp32 [N][T][C] master -> bf16 dst [N][T][C] and/or dst_t [C][T][N] (both * scale[n])."""
    _req(src, torch.float32, "src")
    _check(lib().rt_weight_prep(_p(src), _p(scale), _p(dst), _p(dst_t), N, T, C, _stream()), "rt_weight_prep")


def stem_weight_prep(src, scale, dst):
    _check(lib().rt_stem_weight_prep(_p(src), _p(scale), _p(dst), _stream()), "rt_stem_weight_prep")


def bn_fold(w, b, rm, rv, eps, scale, shift):
    _check(lib().rt_bn_fold(_p(w), _p(b), _p(rm), _p(rv), eps, _p(scale), _p(shift), w.numel(), _stream()), "rt_bn_fold")


def mask_posenc(mask_u8, h, w, C, add_vec, kpm_out, kpm_off, pos_out, rows_per_img, row_off):
    """mask uint8 [B,H,W]; writes kpm_out[b, kpm_off + pix] and pos_out[b*rows_per_img + row_off + pix, :]."""
    B, H, W = mask_u8.shape
    d = MaskPosencDesc(_p(mask_u8), _p(kpm_out), _p(pos_out), _p(add_vec), B, H, W, h, w, C,
                       kpm_out.shape[1], kpm_off, rows_per_img, row_off)
    _check(lib().rt_mask_posenc(ctypes.byref(d), _stream()), "rt_mask_posenc")


# --------------------------------------------------------------------------------------------
# small fused ops
# --------------------------------------------------------------------------------------------
def colsum(dy, db):
    """db[n] += sum_m dy[m, n] (dy bf16 or fp32, 2-D contiguous)."""
    M, N = dy.shape
    _check(lib().rt_colsum(_p(dy), 1 if dy.dtype == torch.bfloat16 else 0, _p(db), M, N, _stream()), "rt_colsum")


def rows_add(rows, D, *, a_f32=None, a_bf16=None, b_f32=None, out_f32=None, out_bf16=None, alpha=1.0,
             accumulate=False, a_map=(0, 0, 0), b_map=(0, 0, 0), o_map=(0, 0, 0)):
    d = RowsAddDesc(_p(a_f32), _p(a_bf16), _p(b_f32), _p(out_f32), _p(out_bf16), rows, D, alpha,
                    int(accumulate), *a_map, *b_map, *o_map)
    _check(lib().rt_rows_add(ctypes.byref(d), _stream()), "rt_rows_add")


def roberta_pos_ids(ids, pad_idx):
    B, L = ids.shape
    out = torch.empty((B, L), dtype=torch.int32, device=ids.device)
    _check(lib().rt_roberta_pos_ids(_p(ids), _p(out), B, L, pad_idx, _stream()), "rt_roberta_pos_ids")
    return out


def bert_embed_fwd(ids, word, pos, type_emb, L, pos_ids=None):
    rows = ids.numel()
    D = word.shape[1]
    out = _new((rows, D), torch.float32, word)
    _check(lib().rt_bert_embed_fwd(_p(ids), _p(word), _p(pos), _p(type_emb), _p(out), rows, L, D, _p(pos_ids), _stream()),
           "rt_bert_embed_fwd")
    return out


def bert_embed_bwd(ids, de, dword, dpos, dtype_emb, L, pos_ids=None):
    rows, D = de.shape
    _check(lib().rt_bert_embed_bwd(_p(ids), _p(de), _p(dword), _p(dpos), _p(dtype_emb), rows, L, D, _p(pos_ids), _stream()),
           "rt_bert_embed_bwd")


def context_mask(smask_u8, phrase_mask_u8=None, pos_l=None, pos_r=None):
    """Returns (ctx uint8 [B,P,L] 1 = ignore, qmask uint8 [B,P] 1 = ignore) — models/reftr_transformer.py:224-248."""
    B, L = smask_u8.shape
    if phrase_mask_u8 is None:
        P, Lp = 1, 0
    else:
        _, P, Lp = phrase_mask_u8.shape
    ctx = _new((B, P, L), torch.uint8, smask_u8)
    qmask = _new((B, P), torch.uint8, smask_u8)
    _check(lib().rt_context_mask(_p(smask_u8), _p(phrase_mask_u8), _p(pos_l), _p(pos_r), _p(ctx), _p(qmask),
                                 B, L, P, Lp, _stream()), "rt_context_mask")
    return ctx, qmask


def qenc_attn_fwd(k, qs, vs, ctx):
    """k [B,E], qs/vs [B,L,E] fp32, ctx uint8 [B,P,L] -> (w [B,P,L], c [B,P,E])."""
    B, L, E = qs.shape
    P = ctx.shape[1]
    w = _new((B, P, L), torch.float32, qs)
    c = _new((B, P, E), torch.float32, qs)
    _check(lib().rt_qenc_attn_fwd(_p(k), _p(qs), _p(vs), _p(ctx), _p(w), _p(c), B, P, L, E, _stream()), "rt_qenc_attn_fwd")
    return w, c


def qenc_attn_bwd(k, qs, vs, w, dc):
    B, L, E = qs.shape
    P = w.shape[1]
    zz = torch.zeros(B * E + 2 * B * L * E, dtype=torch.float32, device=qs.device)       # one clear for the three outputs
    dk, dqs, dvs = zz[:B * E].view(B, E), zz[B * E:B * E + B * L * E].view(B, L, E), zz[B * E + B * L * E:].view(B, L, E)
    _check(lib().rt_qenc_attn_bwd(_p(k), _p(qs), _p(vs), _p(w), _p(dc), _p(dk), _p(dqs), _p(dvs), B, P, L, E, _stream()),
           "rt_qenc_attn_bwd")
    return dk, dqs, dvs


def _fill(desc_cls, ptrs, **scalars):
    """A descriptor whose pointer fields are named tensors (None -> NULL); unknown names are an error."""
    d = desc_cls()
    names = set(desc_cls._PTRS)
    for k, v in ptrs.items():
        if k not in names:
            raise KeyError(f"{desc_cls.__name__}: no pointer field {k!r}")
        setattr(d, k, _p(v))
    for k, v in scalars.items():
        setattr(d, k, v)
    return d


def qenc_fwd(*, B, S, L, P, nq, E, eps=1e-5, drop_p=0.0, drop_seed=0, **t):
    """QueryEncoder forward + query / query_pos split as ONE launch (rt_qenc_fwd).  `t`: the named tensors of rt_qenc_fwd_desc."""
    d = _fill(QencFwdDesc, dict(t, seed_dev=_SEED_DEV if drop_p > 0 else None), B=B, S=S, L=L, P=P, nq=nq, E=E, eps=eps,
              drop_p=drop_p, drop_seed=drop_seed & 0xFFFFFFFF, reserved=0)
    _check(lib().rt_qenc_fwd(ctypes.byref(d), _stream()), "rt_qenc_fwd")


def head_loss(*, NL, B, P, K, E, eps=1e-5, invert_valid=False, **t):
    """decoder.norm + bbox MLP + box losses + d total / d logits + their backward-data as ONE launch (rt_head_loss)."""
    d = _fill(HeadLossDesc, t, NL=NL, B=B, P=P, K=K, E=E, eps=eps, invert_valid=int(bool(invert_valid)), reserved=0)
    _check(lib().rt_head_loss(ctypes.byref(d), _stream()), "rt_head_loss")


def qenc_bwd(*, B, S, L, P, E, drop_p=0.0, drop_seed=0, **t):
    """Backward-data of rt_qenc_fwd as ONE launch (rt_qenc_bwd)."""
    d = _fill(QencBwdDesc, dict(t, seed_dev=_SEED_DEV if drop_p > 0 else None), B=B, S=S, L=L, P=P, E=E, drop_p=drop_p,
              drop_seed=drop_seed & 0xFFFFFFFF, reserved=0)
    _check(lib().rt_qenc_bwd(ctypes.byref(d), _stream()), "rt_qenc_bwd")


def qregion_trace():
    """Stage stamps [3][24] (10 ns units) of workgroup 0 of the last rt_qenc_fwd / rt_head_loss / rt_qenc_bwd launches."""
    buf = (ctypes.c_uint64 * 72)()
    _check(lib().rt_qregion_trace(ctypes.cast(buf, c_void_p)), "rt_qregion_trace")
    return [list(buf[i * 24:(i + 1) * 24]) for i in range(3)]


def box_loss(logits, valid_u8, targets, tgt_off, num_boxes, w_bbox=1.0, w_giou=1.0, want_grad=True, weights=None):
    """logits fp32 [NL,B,P,K,4]; returns (losses [NL,2], total [1], dlogits | None).  `weights` (device fp32
    [NL,2]) overrides the scalar weights per layer."""
    NL, B, P, K, _ = logits.shape
    buf = _new((NL * 2 + 1,), torch.float32, logits)        # [NL][2] losses | total: one clear inside rt_box_loss
    losses, total = buf[:NL * 2].view(NL, 2), buf[NL * 2:]
    dl = _new(tuple(logits.shape), torch.float32, logits) if want_grad else None
    d = BoxLossDesc(_p(logits), _p(valid_u8), _p(targets), _p(tgt_off), _p(num_boxes), _p(losses), _p(total), _p(dl),
                    NL, B, P, K, w_bbox, w_giou, _p(weights))
    _check(lib().rt_box_loss(ctypes.byref(d), _stream()), "rt_box_loss")
    return losses, total, dl


def cem_fwd(hs16, w3, b3, res16, w2, b2, B, HW):
    """CEM block forward (rt_cem_fwd): returns (loss [1], saved = (u [B,16], energy [B], stats [B,2]))."""
    _req(hs16, torch.bfloat16, "hs"); _req(res16, torch.bfloat16, "res")
    for t, n in ((w3, "w3"), (b3, "b3"), (w2, "w2"), (b2, "b2")):
        _req(t, torch.float32, n)
    E = hs16.shape[-1]
    u = _new((B, 16), torch.float32, w3); energy = _new((B,), torch.float32, w3); stats = _new((B, 2), torch.float32, w3)
    loss = _new((1,), torch.float32, w3)
    d = CemDesc(_p(hs16), _p(w3), _p(b3), _p(res16), _p(w2), _p(b2), _p(u), _p(energy), _p(stats), _p(loss), None,
                None, None, None, None, None, B, HW, res16.shape[-1], 0, E, 0)
    _check(lib().rt_cem_fwd(ctypes.byref(d), _stream()), "rt_cem_fwd")
    return loss, (u, energy, stats)


def cem_bwd(hs16, w3, b3, res16, w2, b2, saved, g, dres, dw3, db3, dw2, B, HW):
    """CEM block backward: dres (fp32 [B*HW, ld]) += d res; dw3 / db3 / dw2 accumulate; returns d hs fp32 [B, E]."""
    _req(g, torch.float32, "g"); _req(dres, torch.float32, "dres")
    for t, n in ((dw3, "dw3"), (db3, "db3"), (dw2, "dw2")):
        _req(t, torch.float32, n)
    u, energy, stats = saved
    E = hs16.shape[-1]
    dhs = _new((B, E), torch.float32, w3)
    d = CemDesc(_p(hs16), _p(w3), _p(b3), _p(res16), _p(w2), _p(b2), _p(u), _p(energy), _p(stats), None, _p(g),
                _p(dres), _p(dhs), _p(dw3), _p(db3), _p(dw2), B, HW, res16.shape[-1], dres.shape[-1], E, 0)
    _check(lib().rt_cem_bwd(ctypes.byref(d), _stream()), "rt_cem_bwd")
    return dhs


def mask_postprocess(pred, sizes_i32, max_hw, threshold=0.5, orig_i32=None, origin_off=None, origin_total=0, max_origin=0):
    """pred fp32 [B, Q, h, w] mask logits; sizes_i32 / orig_i32 device int32 [B, 2]; origin_off device int64 [B + 1].
    Returns (masks uint8 [B, Q, max_h, max_w], masks_origin uint8 [origin_total] | None)."""
    B, Q, h, w = pred.shape
    _req(pred, torch.float32, "pred"); _req(sizes_i32, torch.int32, "sizes")
    _req(orig_i32, torch.int32, "orig"); _req(origin_off, torch.int64, "origin_off")
    max_h, max_w = int(max_hw[0]), int(max_hw[1])
    masks = _new((B, Q, max_h, max_w), torch.uint8, pred)
    mo = _new((int(origin_total),), torch.uint8, pred) if orig_i32 is not None else None
    d = MaskPostDesc(_p(pred), _p(sizes_i32), _p(orig_i32), _p(origin_off), _p(masks), _p(mo), B, Q, h, w, max_h, max_w,
                     int(max_origin), float(threshold))
    _check(lib().rt_mask_postprocess(ctypes.byref(d), _stream()), "rt_mask_postprocess")
    return masks, mo


def box_postprocess(boxes, valid_u8, sizes_f32=None):
    """boxes fp32 [B, P, K, 4] cxcywh, valid uint8 [B, P, K]; sizes fp32 [B, 2] = (img_h, img_w) or None.
    Returns (xyxy fp32 [B, P, 4] with every image's valid phrases compacted to the front, counts int32 [B])."""
    B, P, K, _ = boxes.shape
    _req(boxes, torch.float32, "boxes"); _req(valid_u8, torch.uint8, "valid"); _req(sizes_f32, torch.float32, "sizes")
    out = torch.zeros(B, P, 4, dtype=torch.float32, device=boxes.device)
    counts = _new((B,), torch.int32, boxes)
    d = BoxPostDesc(_p(boxes), _p(valid_u8), _p(sizes_f32), _p(out), _p(counts), B, P, K)
    _check(lib().rt_box_postprocess(ctypes.byref(d), _stream()), "rt_box_postprocess")
    return out, counts


def zero_chunks(base, table, n):
    """Clears n chunks {element offset, count} (static device int64 table) of the fp32 buffer `base` in one launch."""
    _req(base, torch.float32, "base"); _req(table, torch.int64, "table")
    _check(lib().rt_zero_chunks(_p(base), _p(table), int(n), _stream()), "rt_zero_chunks")


def sqnorm(g, out):
    if g.dtype == torch.bfloat16:
        _check(lib().rt_sqnorm_bf16(_p(g), g.numel(), _p(out), _stream()), "rt_sqnorm_bf16")
    else:
        _check(lib().rt_sqnorm(_p(g), g.numel(), _p(out), _stream()), "rt_sqnorm")


def adamw_flat(p, g, m, v, *, step, ranges, gnorm_sq=None, gnorm_out=None, grad_scale=1.0, max_norm=0.0,
               beta1=0.9, beta2=0.999, eps=1e-8, step_dev=None, active=None, lr_dev=None, span=None, g16=None, sgd=False,
               mat=None, chunks=None):
    """ranges = [(begin, end, lr, wd), ...] element ranges of the flat buffers (multiples of 4); span = (begin, end)
    restricts the launch to that element span; active / lr_dev: device words (see rt_adamw_desc).  sgd=True: rt_sgd_flat
    (m = momentum buffer, beta1 = momentum, v unused).  mat = (device job table, njobs, total tiles) and / or chunks = (device
    chunk table, nchunks): the matrix-aware form (rt_adamw_mat + rt_adamw_chunks) that also emits the bf16 operands; the
    tables say what is updated and `span` is ignored."""
    d = AdamWDesc()
    d.p, d.g, d.m, d.v, d.n = _p(p), _p(g), _p(m), _p(v), p.numel()
    d.gnorm_sq, d.gnorm_out = _p(gnorm_sq), _p(gnorm_out)
    d.grad_scale, d.max_norm, d.beta1, d.beta2, d.eps = grad_scale, max_norm, beta1, beta2, eps
    d.step, d.n_ranges = step, len(ranges)
    d.step_dev, d.active, d.lr_dev = _p(step_dev), _p(active), _p(lr_dev)
    d.span_begin, d.span_end = (0, 0) if span is None else span
    d.g16 = _p(g16)
    for i, (b, e, lr, wd) in enumerate(ranges):
        d.range_begin[i], d.range_end[i], d.range_lr[i], d.range_wd[i] = b, e, lr, wd
    if mat is not None or chunks is not None:
        assert not sgd
        if mat is not None and mat[1] > 0:
            _check(lib().rt_adamw_mat(ctypes.byref(d), _p(mat[0]), mat[1], mat[2], _stream()), "rt_adamw_mat")
        if chunks is not None and chunks[1] > 0:
            _check(lib().rt_adamw_chunks(ctypes.byref(d), _p(chunks[0]), chunks[1], _stream()), "rt_adamw_chunks")
    elif sgd:
        _check(lib().rt_sgd_flat(ctypes.byref(d), _stream()), "rt_sgd_flat")
    else:
        _check(lib().rt_adamw_flat(ctypes.byref(d), _stream()), "rt_adamw_flat")


def small_dgrad(dy_f32, w_f32, gate=None):
    """dx bf16 [M,K] = gate(dy[M,N<=8] @ w[N,K])."""
    M, N = dy_f32.shape
    K = w_f32.shape[1]
    dx = _new((M, K), torch.bfloat16, dy_f32)
    _check(lib().rt_small_dgrad(_p(dy_f32), _p(w_f32), _p(gate), _p(dx), M, N, K, _stream()), "rt_small_dgrad")
    return dx


def pos_grad(dpos, d_lang_pos, d_type, d_level, B, S, L):
    E = dpos.shape[1]
    _check(lib().rt_pos_grad(_p(dpos), _p(d_lang_pos), _p(d_type), _p(d_level), B, S, L, E, _stream()), "rt_pos_grad")


def counter_add(ctr, inc=1, unless=None, reset_else=False):
    """*ctr += inc on the device; `unless` (a device word): only while it is 0, else *ctr is left alone or -- reset_else --
    cleared (rt_counter_add_if_zero)."""
    if unless is not None:
        _check(lib().rt_counter_add_if_zero(_p(ctr), inc, _p(unless), int(bool(reset_else)), _stream()), "rt_counter_add_if_zero")
    else:
        _check(lib().rt_counter_add(_p(ctr), inc, _stream()), "rt_counter_add")


def finish_step(step_dev, active, veto=None, loss=None):
    """*step_dev += 1, *active += 1 unless the veto word is set or the loss is not finite; else *active = 0 (rt_finish_step)."""
    _req(loss, torch.float32, "loss")
    _check(lib().rt_finish_step(_p(step_dev), _p(active), _p(veto), _p(loss), _stream()), "rt_finish_step")


RT_STATS_MAX = 40


class FinishDesc(Structure):
    _fields_ = [("step", c_void_p), ("active", c_void_p), ("cond", c_void_p), ("loss", c_void_p),
                ("sq", c_void_p), ("norm_scale", c_float), ("grad_norm", c_void_p),
                ("src", c_void_p * RT_STATS_MAX), ("n_src", c_int), ("cond_in_stats", c_int), ("stats", c_void_p)]


def finish_stats(step_dev, active, veto, loss, sq, norm_scale, grad_norm, srcs=(), cond_in_stats=False, stats=None):
    """rt_finish_stats: the iteration's counters (as finish_step), grad_norm = sqrt(sq) * norm_scale and the stats vector
    [*srcs | float(veto word) | grad_norm] in one launch.  `srcs`: fp32 device scalars (tensors of one element)."""
    assert len(srcs) <= RT_STATS_MAX
    for t in list(srcs) + [loss, sq, grad_norm, stats]:
        _req(t, torch.float32, "finish_stats operand")
    d = FinishDesc()
    d.step, d.active, d.cond, d.loss = _p(step_dev), _p(active), _p(veto), _p(loss)
    d.sq, d.norm_scale, d.grad_norm = _p(sq), float(norm_scale), _p(grad_norm)
    for i, t in enumerate(srcs):
        d.src[i] = _p(t)
    d.n_src, d.cond_in_stats, d.stats = len(srcs), int(bool(cond_in_stats)), _p(stats)
    if stats is not None:
        assert stats.numel() == len(srcs) + int(bool(cond_in_stats)) + 1
    _check(lib().rt_finish_stats(ctypes.byref(d), _stream()), "rt_finish_stats")


def stamp(buf, idx):
    """buf[idx] (int64 device tensor) = the device's 100 MHz wall clock when the current stream gets here (rt_stamp)."""
    assert buf.dtype == torch.int64 and 0 <= idx < buf.numel()
    _check(lib().rt_stamp(_p(buf), idx, _stream()), "rt_stamp")


# Measurement aid (tools/concurrent_timeline.py): wall-clock stamps at named points of a step, on whatever stream reaches them.
# rocprofv3 serialises the concurrent streams of a replayed graph, so the step's real timeline is taken from inside the graph: with
# stamping enabled BEFORE the step is captured, every `mark` is a one-thread kernel node writing the device's 100 MHz clock.
_MARKS = None


def enable_marks(device, capacity=256):
    """Turns `mark` on: returns {'buf': int64 device tensor, 'names': {name: (slot, stream)}}.  Call before capturing the step."""
    global _MARKS
    _MARKS = {"buf": torch.zeros(capacity, dtype=torch.int64, device=device), "names": {}}
    return _MARKS


def disable_marks():
    global _MARKS
    _MARKS = None


def mark(name):
    """No-op unless enable_marks() was called (one global read)."""
    m = _MARKS
    if m is None:
        return
    ent = m["names"].get(name)
    if ent is None:
        ent = m["names"][name] = (len(m["names"]), torch.cuda.current_stream().cuda_stream)
    stamp(m["buf"], ent[0])


def decoder_supported(F=2048):
    """True when the cooperative decoder launches may be used on the current device (co-residency check, rt_decoder_supported)."""
    return lib().rt_decoder_supported(int(F)) == 0


def decoder_set_spin(spin):
    _check(lib().rt_decoder_set_spin(int(spin)), "rt_decoder_set_spin")


class WgradBatch:
    """Linear weight gradients queued during backward (they are off the backward-data dependency chain) and launched as
    one group (rt_conv_wgrad_grouped).  `workspace_mb`: scratch for the split partials of the whole group."""

    _WS = {}

    def __init__(self, workspace_mb=512):
        self.descs, self.keep, self.ws_bytes = [], [], workspace_mb << 20
        self.flops = self.nbytes = 0.0

    def _account(self, geom):
        self.flops += 2.0 * geom[0] * geom[4] * geom[5] * geom[6] * geom[7] * geom[8] * geom[3]
        self.nbytes += _algo_bytes(("W",) + tuple(geom))

    def add(self, dy, x, dw, dbias=None, overwrite=False):
        _req(dy, torch.bfloat16, "dy"); _req(x, torch.bfloat16, "x"); _req(dw, torch.float32, "dw"); _req(dbias, torch.float32, "dbias")
        M, N = dy.shape
        K = x.shape[1]
        assert x.shape[0] == M and dw.numel() == N * K
        self.descs.append(ConvWgradDesc(_p(dy), _p(x), _p(dw), None, M, 1, 1, K, 1, 1, N, 1, 1, 1, 0, 0, _p(dbias), 0, None, 0,
                                        int(bool(overwrite)), 0, _sqacc(dw), _g16(dw)))
        self.keep.append((dy, x))
        self._account((M, 1, 1, K, 1, 1, N, 1, 1, 1, 0))

    def add_conv(self, dy, x, dw, geom, scale=None, overwrite=False, dil=1):
        """A convolution weight gradient (any geometry: the non-groupable ones are forwarded to rt_conv_wgrad at run())."""
        B, SH, SW, SC, DH, DW, N, KH, KW, stride, pad = geom
        _req(dy, torch.bfloat16, "dy"); _req(x, torch.bfloat16, "x"); _req(dw, torch.float32, "dw"); _req(scale, torch.float32, "scale")
        self.descs.append(ConvWgradDesc(_p(dy), _p(x), _p(dw), _p(scale), B, SH, SW, SC, DH, DW, N, KH, KW, stride, pad, 0, None, 0,
                                        None, 0, int(bool(overwrite)), int(dil), _sqacc(dw), _g16(dw)))
        self.keep.append((dy, x, scale))
        self._account(geom)

    def run(self):
        if not self.descs:
            return
        dev = self.keep[0][0].device
        key = (dev.index, torch.cuda.current_stream().cuda_stream, self.ws_bytes)
        ws = WgradBatch._WS.get(key)
        if ws is None:
            ws = WgradBatch._WS[key] = torch.empty(self.ws_bytes // 4, dtype=torch.float32, device=dev)
        single = _wgrad_workspace(dev)          # for the descriptors that are forwarded to rt_conv_wgrad (this stream's)
        for d in self.descs:
            d.workspace, d.workspace_bytes = _p(single), WGRAD_WS_BYTES
        arr = (ConvWgradDesc * len(self.descs))(*self.descs)
        n = len(self.descs)
        _timed("conv_wgrad_grouped", self.flops,
               lambda: _check(lib().rt_conv_wgrad_grouped(arr, n, _p(ws), self.ws_bytes, _stream()), "rt_conv_wgrad_grouped"),
               nbytes=self.nbytes)
        self.descs, self.keep = [], []
        self.flops = self.nbytes = 0.0


class SmallWgradBatch:
    """Weight gradients of Linears over <= 16 token rows, queued during backward and launched together
    (rt_small_wgrad_grouped): they are off the backward-data dependency chain, so ~50 launch latencies become one."""

    def __init__(self):
        self.jobs, self.keep = [], []
        self.flops = self.nbytes = 0.0

    def add(self, dy, x, dw, dbias=None, overwrite=False):
        _req(dy, torch.bfloat16, "dy"); _req(x, torch.bfloat16, "x"); _req(dw, torch.float32, "dw"); _req(dbias, torch.float32, "dbias")
        M, N = dy.shape
        K = x.shape[1]
        assert M <= 16 and x.shape[0] == M and dw.numel() == N * K and K % 4 == 0
        self.jobs.append(SmallWgradJob(_p(dy), _p(x), _p(dw), _p(dbias), M, N, K, int(bool(overwrite)), _sqacc(dw), _g16(dw)))
        self.keep.append((dy, x))
        self.flops += 2.0 * M * N * K; self.nbytes += 2.0 * M * (N + K) + 4.0 * N * K

    def run(self):
        if not self.jobs:
            return
        arr = (SmallWgradJob * len(self.jobs))(*self.jobs)
        n = len(self.jobs)
        _timed("small_wgrad_grouped", self.flops,
               lambda: _check(lib().rt_small_wgrad_grouped(arr, n, _stream()), "rt_small_wgrad_grouped"), nbytes=self.nbytes)
        self.jobs, self.keep = [], []
        self.flops = self.nbytes = 0.0


class WeightPrepBatch:
    """All per-step operand refreshes as ONE launch: jobs are (src fp32 [N,T,C], scale|None, dst|None, dst_t|None)."""

    def __init__(self, device):
        self.jobs, self.device, self.table, self.tiles = [], device, None, 0
        self._keep = []

    def add(self, src, N, T, C, scale=None, dst=None, dst_t=None):
        assert src.is_contiguous() and src.numel() == N * T * C
        self.jobs.append([_p(src), _p(scale) or 0, _p(dst) or 0, _p(dst_t) or 0, N, T, C, self.tiles])
        self.tiles += ((N + 63) // 64) * ((C + 63) // 64) * T
        self._keep.append((src, scale, dst, dst_t))
        self.table = None

    def run(self):
        if not self.jobs:
            return
        if self.table is None:
            self.table = torch.tensor(self.jobs, dtype=torch.int64).to(self.device)
        _check(lib().rt_weight_prep_batched(_p(self.table), len(self.jobs), self.tiles, _stream()), "rt_weight_prep_batched")


# --------------------------------------------------------------------------------------------
# RES head (RefTRSeg) kernels
# --------------------------------------------------------------------------------------------
def gn_nhwc_fwd(x, gamma, beta, B, HW, C, G=8, ldy=None, act=ACT_RELU, eps=1e-5):
    """y = [relu](GroupNorm(G, C)(x)) as a bf16 [B*HW, ldy] operand (padding zero); returns (y_bf16, stats)."""
    _req(x, torch.float32, "x"); _req(gamma, torch.float32, "gamma"); _req(beta, torch.float32, "beta")
    ldx = x.shape[-1]
    ldy = ldy or ldx
    assert x.numel() == B * HW * ldx and gamma.numel() >= C
    y = torch.empty((B * HW, ldy), dtype=torch.bfloat16, device=x.device)
    stats = torch.empty((B, G, 2), dtype=torch.float32, device=x.device)
    ppb = max(1, (4096 + C - 1) // C)          # rt_gn_nhwc_fwd cuts the pixels the same way
    nblk = (HW + ppb - 1) // ppb
    partials = torch.empty((B, nblk, G, 2), dtype=torch.float32, device=x.device)
    d = GnNhwcDesc(_p(x), _p(gamma), _p(beta), _p(stats), _p(y), B, HW, C, G, ldx, ldy, act, eps, _p(partials), nblk)
    _check(lib().rt_gn_nhwc_fwd(ctypes.byref(d), _stream()), "rt_gn_nhwc_fwd")
    return y, stats


def gn_nhwc_bwd(dy, x, gamma, beta, stats, dgamma, dbeta, B, HW, C, G=8, lddx=None, act=ACT_RELU, eps=1e-5):
    """dx (bf16 [B*HW, lddx], the producing convolution's output gradient); dgamma / dbeta accumulated."""
    _req(dy, torch.float32, "dy"); _req(x, torch.float32, "x")
    ldx, lddy = x.shape[-1], dy.shape[-1]
    lddx = lddx or ldx
    dx = torch.empty((B * HW, lddx), dtype=torch.bfloat16, device=x.device)
    bst = torch.empty((B, G, 2), dtype=torch.float32, device=x.device)
    d = GnNhwcBwdDesc(_p(dy), _p(x), _p(gamma), _p(beta), _p(stats), _p(bst), _p(dx), _p(dgamma), _p(dbeta),
                      B, HW, C, G, ldx, lddy, lddx, act, eps)
    _check(lib().rt_gn_nhwc_bwd(ctypes.byref(d), _stream()), "rt_gn_nhwc_bwd")
    return dx


def upsample_add(fpn, a, B, H, W, h, w, C, ldo=None):
    """bf16(fpn + nearest_upsample(a)): fpn fp32 [B*H*W, ldf], a bf16 [B*h*w, lda] -> bf16 [B*H*W, ldo]."""
    _req(fpn, torch.float32, "fpn"); _req(a, torch.bfloat16, "a")
    ldf, lda = fpn.shape[-1], a.shape[-1]
    ldo = ldo or ldf
    out = torch.empty((B * H * W, ldo), dtype=torch.bfloat16, device=fpn.device)
    d = UpsampleAddDesc(_p(fpn), _p(a), _p(out), B, H, W, h, w, C, ldf, lda, ldo)
    _check(lib().rt_upsample_add(ctypes.byref(d), _stream()), "rt_upsample_add")
    return out


def upsample_add_bwd(dy, B, H, W, h, w, C, ldda=None, lddyb=None):
    """dy fp32 [B*H*W, lddy] -> (da fp32 [B*h*w, ldda] (padding zero), bf16 copy of dy [B*H*W, lddyb])."""
    _req(dy, torch.float32, "dy")
    lddy = dy.shape[-1]
    ldda = ldda or C; lddyb = lddyb or lddy
    da = torch.zeros((B * h * w, ldda), dtype=torch.float32, device=dy.device) if ldda > C else \
        torch.empty((B * h * w, ldda), dtype=torch.float32, device=dy.device)
    dyb = torch.empty((B * H * W, lddyb), dtype=torch.bfloat16, device=dy.device)
    d = UpsampleAddBwdDesc(_p(dy), _p(da), _p(dyb), B, H, W, h, w, C, lddy, ldda, lddyb)
    _check(lib().rt_upsample_add_bwd(ctypes.byref(d), _stream()), "rt_upsample_add_bwd")
    return da, dyb


def attn_map_fwd(q, k, mask_u8, B, HW, E, nh, k_rows_per_img, k_row_off, concat=None, concat_col=0):
    _req(q, torch.float32, "q"); _req(k, torch.float32, "k"); _req(mask_u8, torch.uint8, "mask")
    P = torch.empty((B, nh, HW), dtype=torch.float32, device=q.device)
    d = AttnMapDesc(_p(q), _p(k), _p(mask_u8), _p(P), _p(concat), B, HW, E, nh, k.shape[-1], k_rows_per_img, k_row_off,
                    concat.shape[-1] if concat is not None else 0, concat_col, float(E / nh) ** -0.5)
    _check(lib().rt_attn_map_fwd(ctypes.byref(d), _stream()), "rt_attn_map_fwd")
    return P


def attn_map_bwd(q, k, P, dconcat, B, HW, E, nh, k_rows_per_img, k_row_off, concat_col):
    """Returns (dq fp32 [B,E], dk fp32 shaped like k: rows outside the image block are zero)."""
    _req(dconcat, torch.float32, "dconcat")
    dq = torch.empty((B, E), dtype=torch.float32, device=q.device)
    dk = torch.zeros_like(k)
    d = AttnMapBwdDesc(_p(q), _p(k), _p(P), _p(dconcat), _p(dq), _p(dk), B, HW, E, nh, k.shape[-1], k_rows_per_img,
                       k_row_off, dconcat.shape[-1], concat_col, float(E / nh) ** -0.5)
    _check(lib().rt_attn_map_bwd(ctypes.byref(d), _stream()), "rt_attn_map_bwd")
    return dq, dk


def seg_concat(src, mem, out, B, HW, E, nh, rows_per_img, row_off, src_dense=False):
    """src / mem rows of image b, pixel p: (b * rows_per_img + row_off + p); src_dense: src is a dense [B*HW, E]."""
    _req(src, torch.float32, "src"); _req(mem, torch.float32, "mem"); _req(out, torch.bfloat16, "out")
    d = SegConcatDesc(_p(src), _p(mem), _p(out), B, HW, E, nh, out.shape[-1], rows_per_img, row_off,
                      HW if src_dense else rows_per_img, 0 if src_dense else row_off)
    _check(lib().rt_seg_concat(ctypes.byref(d), _stream()), "rt_seg_concat")
    return out


def mask_loss(pred, target_u8, B, h, w, Ht, Wt, ldp, norm, sums=None, dpred=None, g_focal=None, g_dice=None):
    """Forward (dpred None): returns (losses[2] = {focal, dice}, sums [B,4]).  Backward: accumulates into dpred."""
    _req(pred, torch.float32, "pred"); _req(target_u8, torch.uint8, "target")
    assert target_u8.numel() == B * Ht * Wt
    if dpred is None:
        sums = torch.empty((B, 4), dtype=torch.float32, device=pred.device)
        losses = torch.empty(2, dtype=torch.float32, device=pred.device)
        d = MaskLossDesc(_p(pred), _p(target_u8), _p(sums), _p(losses), None, None, None, B, h, w, Ht, Wt, ldp, 0, 1.0 / norm, None)
        _check(lib().rt_mask_loss(ctypes.byref(d), _stream()), "rt_mask_loss")
        return losses, sums
    _req(dpred, torch.float32, "dpred"); _req(g_focal, torch.float32, "g_focal"); _req(g_dice, torch.float32, "g_dice")
    gbuf = torch.empty(B * Ht * Wt, dtype=torch.float32, device=pred.device)
    d = MaskLossDesc(_p(pred), _p(target_u8), _p(sums), None, _p(dpred), _p(g_focal), _p(g_dice), B, h, w, Ht, Wt, ldp,
                     dpred.shape[-1], 1.0 / norm, _p(gbuf))
    _check(lib().rt_mask_loss(ctypes.byref(d), _stream()), "rt_mask_loss")
    return dpred


# --------------------------------------------------------------------------------------------
# input pipeline kernels
# --------------------------------------------------------------------------------------------
def resample_u8(src, bounds, coeffs, out_len, axis):
    """One Pillow-compatible resampling pass of a uint8 [n0, n1, C] image along `axis` (0 vertical / 1 horizontal)."""
    _req(src, torch.uint8, "src"); _req(bounds, torch.int32, "bounds"); _req(coeffs, torch.int32, "coeffs")
    n0, n1, C = src.shape
    shape = (out_len, n1, C) if axis == 0 else (n0, out_len, C)
    dst = torch.empty(shape, dtype=torch.uint8, device=src.device)
    _check(lib().rt_resample_u8(_p(src), _p(dst), _p(bounds), _p(coeffs), n0, n1, C, out_len, coeffs.shape[1], axis, _stream()),
           "rt_resample_u8")
    return dst


def img_collate_norm(images, H, W, mean, std):
    """images: list of uint8 [h, w, 3] device tensors -> (fp32 [B,3,H,W] normalised + zero padded, uint8 mask [B,H,W])."""
    B = len(images)
    for im in images:
        _req(im, torch.uint8, "image")
        assert im.dim() == 3 and im.shape[2] == 3 and im.shape[0] <= H and im.shape[1] <= W
    dev = images[0].device
    tab = torch.tensor([[im.data_ptr(), im.shape[0], im.shape[1]] for im in images], dtype=torch.int64).to(dev, non_blocking=True)
    out = torch.empty((B, 3, H, W), dtype=torch.float32, device=dev)
    mask = torch.empty((B, H, W), dtype=torch.uint8, device=dev)
    m = (c_float * 3)(*[float(v) for v in mean]); s_ = (c_float * 3)(*[float(v) for v in std])
    _check(lib().rt_img_collate_norm(_p(tab), _p(out), _p(mask), B, H, W, m, s_, _stream()), "rt_img_collate_norm")
    return out, mask


# --------------------------------------------------------------------------------------------
# gradient exchange behind the C ABI (rt_comm_*: RCCL bound at run time)
# --------------------------------------------------------------------------------------------
class Comm:
    """One RCCL communicator of this process (one process per GPU) through rt_comm_*.  `uid` travels from rank 0 to the other
    ranks by whatever channel the caller has (reftr_amd.parallel uses the torch.distributed rendezvous that main_vg.py set up)."""

    @staticmethod
    def unique_id():
        buf = ctypes.create_string_buffer(128)
        _check(lib().rt_comm_unique_id(buf), "rt_comm_unique_id")
        return bytes(buf.raw)

    def __init__(self, uid, rank, world):
        assert len(uid) == 128
        h = c_void_p()
        _check(lib().rt_comm_init(ctypes.create_string_buffer(uid, 128), int(rank), int(world), ctypes.byref(h)), "rt_comm_init")
        self.handle, self.rank, self.world = h, rank, world

    def allreduce(self, tensors, stream=None):
        """In-place SUM all-reduce of contiguous device tensors of one dtype (fp32 or bf16), one RCCL group, asynchronous on
        `stream` (default: the current stream)."""
        tensors = [t for t in tensors if t.numel()]
        if not tensors:
            return
        dt = tensors[0].dtype
        assert dt in (torch.float32, torch.bfloat16) and all(t.dtype == dt and t.is_cuda and t.is_contiguous() for t in tensors)
        n = len(tensors)
        ptrs = (c_void_p * n)(*[t.data_ptr() for t in tensors])
        cnts = (ctypes.c_int64 * n)(*[t.numel() for t in tensors])
        s = stream.cuda_stream if stream is not None else _stream()
        _check(lib().rt_comm_allreduce(self.handle, ptrs, cnts, n, 0 if dt == torch.float32 else 1, s), "rt_comm_allreduce")

    def destroy(self):
        if self.handle:
            _check(lib().rt_comm_destroy(self.handle), "rt_comm_destroy")
            self.handle = None


class SideStream:
    """A second HIP stream for work that is off the critical path (the BERT branch, queued weight gradients).
    `run` forks it from the current stream (event edge) -- or, with defer=True, only queues the closure until
    `flush` forks ONCE and launches the whole queue (per-launch fork edges cost more than they give inside a
    hipGraph); `join` makes the current stream wait for it.  Tensors the side work reads are kept referenced until
    the join so the caching allocator cannot hand their memory out early.  Under hipGraph capture the fork / join
    events become graph edges and the branches replay concurrently."""

    def __init__(self, enabled=True, defer=False):
        self.enabled = enabled and torch.cuda.is_available()
        self.stream = torch.cuda.Stream() if self.enabled else None
        self.defer = defer
        self.jobs = []
        self.keep = []
        self.dirty = False

    def run(self, fn, *keep):
        if not self.enabled:
            return fn()
        self.keep.extend(keep)
        if self.defer:
            self.jobs.append(fn)
            return None
        self.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            r = fn()
        self.dirty = True
        return r

    def flush(self):
        if not self.enabled or not self.jobs:
            return
        self.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            for fn in self.jobs:
                fn()
        self.jobs = []
        self.dirty = True

    def join(self):
        self.flush()
        if self.enabled and self.dirty:
            torch.cuda.current_stream().wait_stream(self.stream)
        self.keep.clear()
        self.dirty = False
