"""ctypes binding of libreftr_hip.so (include/reftr_hip.h) + tensor-level op wrappers.

PyTorch is used here only as the owner of device memory and of the HIP stream the kernels are enqueued
on (torch.cuda.current_stream()).  There is NO fallback: if the library is missing or a kernel reports
an error, a RuntimeError is raised.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_uint32, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libreftr_hip.so")
ABI_VERSION = 1

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
    ]


class ConvWgradDesc(Structure):
    _fields_ = [
        ("dy", c_void_p), ("x", c_void_p), ("dw", c_void_p), ("scale", c_void_p),
        ("B", c_int32), ("SH", c_int32), ("SW", c_int32), ("SC", c_int32),
        ("DH", c_int32), ("DW", c_int32), ("N", c_int32),
        ("KH", c_int32), ("KW", c_int32), ("stride", c_int32), ("pad", c_int32),
        ("msplit", c_int32),
    ]


# name -> (restype, argtypes); every symbol include/reftr_hip.h declares must be listed here
# (tests/test_abi.py cross-checks this table against the header).
_SIGNATURES = {
    "rt_abi_version": (c_int, []),
    "rt_device_arch": (c_int, [c_int, c_char_p, c_int]),
    "rt_conv_gemm": (c_int, [POINTER(ConvGemmDesc), c_void_p]),
    "rt_conv_wgrad": (c_int, [POINTER(ConvWgradDesc), c_void_p]),
}

_lib = None


def lib():
    """Load libreftr_hip.so (once).  Fails loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: the HIP kernel library is not built. Run "
                "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc; no CPU fallback exists).")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        v = L.rt_abi_version()
        if v != ABI_VERSION:
            raise RuntimeError(f"libreftr_hip.so ABI {v} != binding ABI {ABI_VERSION}; rebuild")
        _lib = L
    return _lib


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
              preact=None, act=ACT_NONE, drop_p=0.0, drop_seed=0, transposed=False,
              out_bf16=True, out_f32=False, tile_hint=0):
    """out[B,DH,DW,N] = epilogue(implicit_gemm(src[B,SH,SW,SC], wgt[N,KH,KW,SC])).

    geom = (B, SH, SW, SC, DH, DW, N, KH, KW, stride, pad).  Returns (out_bf16 | None, out_f32 | None).
    """
    B, SH, SW, SC, DH, DW, N, KH, KW, stride, pad = geom
    _req(src, torch.bfloat16, "src"); _req(wgt, torch.bfloat16, "wgt")
    _req(bias, torch.float32, "bias"); _req(res_f32, torch.float32, "res_f32")
    _req(res_bf16, torch.bfloat16, "res_bf16"); _req(gate, torch.bfloat16, "gate")
    _req(preact, torch.bfloat16, "preact")
    assert src.numel() == B * SH * SW * SC, (src.shape, geom)
    assert wgt.numel() == N * KH * KW * SC, (wgt.shape, geom)
    M = B * DH * DW
    ob = torch.empty((M, N), dtype=torch.bfloat16, device=src.device) if out_bf16 else None
    of = torch.empty((M, N), dtype=torch.float32, device=src.device) if out_f32 else None
    d = ConvGemmDesc(_p(src), _p(wgt), _p(ob), _p(of), _p(bias), _p(res_f32), _p(res_bf16), _p(gate),
                     _p(preact), B, SH, SW, SC, DH, DW, N, KH, KW, stride, pad,
                     1 if transposed else 0, act, gate_scale, drop_p, drop_seed & 0xFFFFFFFF, tile_hint)
    _check(lib().rt_conv_gemm(ctypes.byref(d), _stream()), "rt_conv_gemm")
    return ob, of


def linear(x, w, bias=None, **kw):
    """y[M,N] = epilogue(x[M,K] @ w[N,K]^T): a 1x1 'conv' over M rows."""
    M, K = x.shape
    N = w.shape[0]
    return conv_gemm(x, w, geom=(M, 1, 1, K, 1, 1, N, 1, 1, 1, 0), bias=bias, **kw)


def conv_wgrad(dy, x, dw, *, geom, scale=None, msplit=0):
    """dw[N,KH,KW,SC] (fp32, accumulated) += scale[n] * sum_m dy[m,n] * gather(x)[m,(kh,kw,c)]."""
    B, SH, SW, SC, DH, DW, N, KH, KW, stride, pad = geom
    _req(dy, torch.bfloat16, "dy"); _req(x, torch.bfloat16, "x"); _req(dw, torch.float32, "dw")
    _req(scale, torch.float32, "scale")
    assert dy.numel() == B * DH * DW * N and x.numel() == B * SH * SW * SC and dw.numel() == N * KH * KW * SC
    d = ConvWgradDesc(_p(dy), _p(x), _p(dw), _p(scale), B, SH, SW, SC, DH, DW, N, KH, KW, stride, pad, msplit)
    _check(lib().rt_conv_wgrad(ctypes.byref(d), _stream()), "rt_conv_wgrad")
    return dw


def linear_wgrad(dy, x, dw, **kw):
    M, N = dy.shape
    K = x.shape[1]
    return conv_wgrad(dy, x, dw, geom=(M, 1, 1, K, 1, 1, N, 1, 1, 1, 0), **kw)
