"""GPU parity of rt_conv_gemm / rt_conv_wgrad against a plain torch fp32 reference of the same op
(floating-point kernels: inputs are bf16-representable, accumulation fp32 -> fp32 outputs must agree to
~1e-5 rel-L2; bf16 outputs to bf16 rounding)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL_F32 = 2e-5      # fp32-accumulated result vs fp32 CPU reference (summation order only)
TOL_BF16 = 3e-3     # result rounded to bf16 (2^-9 relative per element)


# The product library instantiates only the tile variants its heuristics can choose (hip.PRODUCT_TILE_HINTS); every other measured variant
# lives in the lab library (same sources, -DRT_LAB).  Those cases run in a CHILD pytest process whose one library IS the lab library
# (REFTR_LAB=1; test_lab_variants_in_child_process below): loading a second copy of every kernel into the suite's own process made later
# hipGraph replays of the model tests segfault inside the runtime (round 6, reproducibly, in different tests from run to run).
LAB_CHILD = os.environ.get("REFTR_GEMM_LAB_CHILD", "0") == "1"


def lib_for(hip, hint):
    import contextlib
    if hint not in hip.PRODUCT_TILE_HINTS and not LAB_CHILD:
        pytest.skip("lab-library variant: runs in the REFTR_LAB=1 child process (test_lab_variants_in_child_process)")
    return contextlib.nullcontext()


def test_lab_variants_in_child_process(suite_note):
    """Every tile / stage / schedule variant of rt_conv_gemm that is not in the product library, against torch fp32, in a process of its own."""
    import subprocess, sys
    if LAB_CHILD:
        return                                  # (this IS the child)
    env = dict(os.environ, REFTR_LAB="1", REFTR_GEMM_LAB_CHILD="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider"],
                       env=env, capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    tail = "\n".join(r.stdout.strip().splitlines()[-5:])
    print("\n[lab-library child] " + tail)
    suite_note("[tests/test_gemm_gpu.py, lab-library child process (REFTR_LAB=1): the variants skipped above] " + r.stdout.strip().splitlines()[-1])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in tail and "skipped" not in tail, tail     # nothing is skipped under the lab library


def rel(a, b):
    a = a.float().cpu()
    b = b.float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def bf(x):
    return x.bfloat16()


def nhwc(x):  # NCHW -> NHWC contiguous
    return x.permute(0, 2, 3, 1).contiguous()


def hash_keep(seed, idx, p):
    """numpy restatement of rt_hash32 / rt_drop_thresh (csrc/rt_common.h)."""
    x = (idx.astype(np.uint64) * np.uint64(0x9E3779B1)) & np.uint64(0xFFFFFFFF)
    x = x ^ np.uint64(seed)
    x ^= x >> np.uint64(16); x = (x * np.uint64(0x85EBCA6B)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(13); x = (x * np.uint64(0xC2B2AE35)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16)
    thresh = np.uint64(int(float(np.float32(p)) * 4294967296.0))
    return x >= thresh


@pytest.mark.parametrize("M,K,N,hint", [
    (200, 256, 256, 0), (200, 256, 256, 1), (200, 256, 256, 2), (200, 256, 256, 3),
    (3520, 256, 2048, 0), (3520, 2048, 256, 0), (320, 768, 3072, 0), (8, 256, 256, 0),
    (48, 256, 4, 0), (129, 64, 68, 0),
    # LDS-DMA variants (tile / stages): ragged M and N, K of one tile and of many
    (200, 256, 256, 11), (200, 256, 256, 12), (200, 256, 256, 13), (333, 64, 68, 21), (333, 192, 68, 22),
    (129, 64, 68, 31), (3520, 2048, 256, 32), (320, 768, 3072, 33), (3520, 256, 2048, 13),
    # 8-wave workgroups
    (200, 256, 256, 51), (3520, 2048, 256, 52), (333, 192, 72, 53), (12800, 256, 1024, 54), (129, 64, 68, 51),
    (700, 256, 264, 61), (3520, 2048, 256, 62), (333, 192, 328, 63),
    # software-pipelined LDS-DMA variants (fragments of tile kt+1 read under the MFMAs of tile kt): one K tile, odd and even counts
    (129, 64, 68, 231), (200, 256, 256, 231), (3520, 2048, 256, 233), (320, 768, 3072, 233), (333, 192, 72, 221), (333, 320, 72, 221),
    (320, 768, 3072, 81), (320, 3072, 768, 281), (333, 192, 72, 281), (129, 64, 68, 282), (320, 768, 768, 282), (333, 320, 72, 283),
    (320, 3072, 768, 284), (50, 128, 40, 281),
    # deep-stage forms: fewer K tiles than stages (64, 192), as many, and many more
    (320, 3072, 768, 285), (333, 192, 72, 285), (50, 64, 40, 285), (320, 384, 768, 285), (320, 768, 768, 286), (129, 64, 68, 286),
    (333, 320, 72, 287), (320, 2304, 768, 288), (3520, 2048, 256, 234), (129, 64, 68, 234), (3520, 2048, 256, 236), (200, 256, 256, 236),
    (700, 256, 264, 261), (3520, 2048, 256, 262),
    (200, 256, 256, 211), (700, 320, 264, 251), (3520, 2048, 256, 252), (129, 64, 68, 251), (12800, 256, 1024, 251),
    # K-parity ping-pong forms (two 4-wave groups alternate memory / MFMA segments, 4 stages, 3 tiles in flight): 1, 2, 3, 4, 5, 6
    # K tiles (prologue-only, first steady iteration, odd and even counts, the empty trailing iteration of group 0) and many;
    # ragged M and N (N a multiple of 8: the groups' partial sums meet in the LDS-staged epilogue)
    (129, 64, 72, 351), (200, 128, 256, 351), (333, 192, 264, 351), (200, 256, 256, 351), (700, 320, 264, 351), (333, 384, 136, 351),
    (3520, 2048, 256, 351), (12800, 256, 1024, 351), (320, 3072, 768, 351), (1000, 2304, 256, 351),
    (129, 64, 72, 331), (333, 192, 72, 331), (200, 256, 256, 331), (320, 768, 3072, 331), (3520, 2048, 256, 331), (50, 320, 40, 331),
    (129, 64, 72, 321), (333, 320, 72, 321), (3520, 2048, 256, 321), (700, 128, 264, 321),
    (129, 64, 72, 323), (333, 320, 72, 323), (3520, 2048, 256, 323), (700, 128, 264, 323),
    # activation-stationary form (rt_gemm_astat.hip: K = 128 / 256 only): ragged rows / columns, one column tile, split column ranges
    (129, 128, 72, 501), (333, 256, 264, 501), (64, 128, 512, 501), (1000, 256, 1024, 501), (12800, 256, 1024, 501), (3520, 256, 2048, 501),
    (51200, 128, 512, 501), (200, 256, 256, 501), (50, 256, 40, 501),
    # three-stage ping-pong forms (the issuing group waits behind its MFMA segment)
    (129, 64, 72, 352), (200, 128, 256, 352), (333, 192, 264, 352), (200, 256, 256, 352), (700, 320, 264, 352), (333, 384, 136, 352),
    (3520, 2048, 256, 352), (12800, 256, 1024, 352), (320, 3072, 768, 352), (1000, 2304, 256, 352),
    (129, 64, 72, 332), (333, 192, 72, 332), (320, 768, 3072, 332), (3520, 2048, 256, 332), (50, 320, 40, 332),
    (129, 64, 72, 322), (333, 320, 72, 322), (3520, 2048, 256, 322), (700, 128, 264, 322),
])
def test_linear_fwd(hip, M, K, N, hint):
    g = torch.Generator().manual_seed(M * 7 + N)
    x = bf(torch.randn(M, K, generator=g))
    w = bf(torch.randn(N, K, generator=g) / K ** 0.5)
    b = torch.randn(N, generator=g)
    ref = x.float() @ w.float().T + b
    with lib_for(hip, hint):
        ob, of = hip.linear(x.cuda(), w.cuda(), bias=b.cuda(), out_bf16=True, out_f32=True, tile_hint=hint)
    assert rel(of, ref) < TOL_F32
    assert rel(ob, ref) < TOL_BF16
    # asymmetric check against transposition: single row/col spot values
    assert torch.allclose(of.cpu()[M - 1, N - 1], ref[M - 1, N - 1], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("hint", [0, 501, 351, 331, 321, 352, 332])
def test_linear_epilogue(hip, hint):
    with lib_for(hip, hint):
        _linear_epilogue(hip, hint)


def _linear_epilogue(hip, hint):
    g = torch.Generator().manual_seed(5)
    M, K, N = 333, 128, 192
    x = bf(torch.randn(M, K, generator=g)); w = bf(torch.randn(N, K, generator=g) / K ** 0.5)
    b = torch.randn(N, generator=g)
    rf = torch.randn(M, N, generator=g); rb = bf(torch.randn(M, N, generator=g))
    gate = bf(torch.randn(M, N, generator=g)); pre = bf(torch.randn(M, N, generator=g))
    lin = x.float() @ w.float().T + b
    xc, wc, bc = x.cuda(), w.cuda(), b.cuda()
    for act, fn in [(hip.ACT_RELU, torch.relu), (hip.ACT_GELU, F.gelu), (hip.ACT_TANH, torch.tanh)]:
        _, of = hip.linear(xc, wc, bias=bc, act=act, out_bf16=False, out_f32=True, tile_hint=hint)
        assert rel(of, fn(lin)) < 1e-4, act
    _, of = hip.linear(xc, wc, bias=bc, res_f32=rf.cuda(), res_bf16=rb.cuda(), out_bf16=False, out_f32=True, tile_hint=hint)
    assert rel(of, lin + rf + rb.float()) < TOL_F32
    _, of = hip.linear(xc, wc, bias=bc, gate=gate.cuda(), gate_scale=1.25, out_bf16=False, out_f32=True, tile_hint=hint)
    assert rel(of, lin * (gate.float() > 0) * 1.25) < TOL_F32
    _, of = hip.linear(xc, wc, bias=bc, preact=pre.cuda(), out_bf16=False, out_f32=True, tile_hint=hint)
    u = pre.float().requires_grad_(True)
    F.gelu(u).sum().backward()
    assert rel(of, lin * u.grad) < 1e-4
    # dropout: mask must be the documented hash of (seed, m*N+n)
    p, seed = 0.1, 1234
    _, of = hip.linear(xc, wc, bias=bc, drop_p=p, drop_seed=seed, out_bf16=False, out_f32=True, tile_hint=hint)
    keep = torch.from_numpy(hash_keep(seed, np.arange(M * N, dtype=np.uint64), p).reshape(M, N))
    assert rel(of, lin * keep / (1 - np.float32(p))) < TOL_F32
    assert abs(float(keep.float().mean()) - 0.9) < 0.01


CONV_CASES = [
    # B, H, W, Cin, Cout, k, stride, pad
    (2, 20, 20, 64, 64, 3, 1, 1),
    (2, 20, 24, 128, 128, 3, 2, 1),
    (2, 20, 20, 256, 512, 1, 2, 0),
    (2, 10, 14, 512, 128, 1, 1, 0),
    (1, 13, 17, 64, 128, 3, 2, 1),
    (3, 9, 9, 64, 64, 3, 1, 1),
]


@pytest.mark.parametrize("hint", [11, 12, 13, 21, 22, 31, 32, 33, 51, 52, 53, 54, 61, 62, 63, 211, 221, 231, 233, 251, 252, 351, 321, 323, 331, 352, 322, 332])
@pytest.mark.parametrize("B,H,W,Ci,Co,k,s,p", [CONV_CASES[1], CONV_CASES[4], CONV_CASES[2]])
def test_conv_dma_variants(hip, hint, B, H, W, Ci, Co, k, s, p):
    """The LDS-DMA tile variants against torch fp32: forward gather and transposed (backward-data) gather."""
    with lib_for(hip, hint):
        _conv_dma_variants(hip, hint, B, H, W, Ci, Co, k, s, p)


def _conv_dma_variants(hip, hint, B, H, W, Ci, Co, k, s, p):
    g = torch.Generator().manual_seed(hint * 1000 + H)
    x = bf(torch.randn(B, Ci, H, W, generator=g)).float().requires_grad_(True)
    w = bf(torch.randn(Co, Ci, k, k, generator=g) / (Ci * k * k) ** 0.5).float().requires_grad_(True)
    bias = torch.randn(Co, generator=g)
    y = F.conv2d(x, w, bias, stride=s, padding=p)
    Ho, Wo = y.shape[-2:]
    dy = bf(torch.randn(y.shape, generator=g))
    y.backward(dy.float())
    x_nhwc = nhwc(x.detach()).bfloat16().cuda()
    w_k = w.detach().permute(0, 2, 3, 1).contiguous().bfloat16().cuda()
    _, of = hip.conv_gemm(x_nhwc, w_k, geom=(B, H, W, Ci, Ho, Wo, Co, k, k, s, p), bias=bias.cuda(), out_bf16=False,
                          out_f32=True, tile_hint=hint)
    assert rel(of, nhwc(y.detach()).reshape(-1, Co)) < TOL_F32
    w_t = w.detach().permute(1, 2, 3, 0).contiguous().bfloat16().cuda()
    _, dxf = hip.conv_gemm(nhwc(dy).cuda(), w_t, geom=(B, Ho, Wo, Co, H, W, Ci, k, k, s, p), transposed=True,
                           out_bf16=False, out_f32=True, tile_hint=hint)
    assert rel(dxf, nhwc(x.grad).reshape(-1, Ci)) < TOL_F32


@pytest.mark.parametrize("B,H,W,Ci,Co,k,s,p", CONV_CASES)
def test_conv_fwd_dgrad_wgrad(hip, B, H, W, Ci, Co, k, s, p):
    g = torch.Generator().manual_seed(B * 100 + H + Ci)
    x = bf(torch.randn(B, Ci, H, W, generator=g)).float().requires_grad_(True)
    w = bf(torch.randn(Co, Ci, k, k, generator=g) / (Ci * k * k) ** 0.5).float().requires_grad_(True)
    bias = torch.randn(Co, generator=g)
    y = F.conv2d(x, w, bias, stride=s, padding=p)
    Ho, Wo = y.shape[-2:]
    dy = bf(torch.randn(y.shape, generator=g))
    y.backward(dy.float())
    x_nhwc = nhwc(x.detach()).bfloat16().cuda()
    w_k = w.detach().permute(0, 2, 3, 1).contiguous().bfloat16().cuda()        # [Co][kh][kw][Ci]
    geom = (B, H, W, Ci, Ho, Wo, Co, k, k, s, p)
    ob, of = hip.conv_gemm(x_nhwc, w_k, geom=geom, bias=bias.cuda(), out_bf16=True, out_f32=True)
    ref = nhwc(y.detach()).reshape(-1, Co)
    assert rel(of, ref) < TOL_F32
    assert rel(ob, ref) < TOL_BF16
    # backward-data: transposed gather over dy with [Ci][kh][kw][Co] weights
    w_t = w.detach().permute(1, 2, 3, 0).contiguous().bfloat16().cuda()
    dy_nhwc = nhwc(dy).cuda()
    geom_t = (B, Ho, Wo, Co, H, W, Ci, k, k, s, p)
    _, dxf = hip.conv_gemm(dy_nhwc, w_t, geom=geom_t, transposed=True, out_bf16=False, out_f32=True)
    assert rel(dxf, nhwc(x.grad).reshape(-1, Ci)) < TOL_F32
    # weight gradient (+ per-channel scale)
    scale = torch.rand(Co, generator=g) + 0.5
    dw = torch.zeros(Co, k, k, Ci, device="cuda")
    hip.conv_wgrad(dy_nhwc, x_nhwc, dw, geom=geom, scale=scale.cuda())
    ref_dw = w.grad.permute(0, 2, 3, 1) * scale.view(-1, 1, 1, 1)
    assert rel(dw, ref_dw) < TOL_F32
    # accumulation semantics + explicit split
    hip.conv_wgrad(dy_nhwc, x_nhwc, dw, geom=geom, scale=scale.cuda(), msplit=3)
    assert rel(dw, 2 * ref_dw) < TOL_F32


@pytest.mark.parametrize("M,K,N", [(200, 256, 256), (3520, 256, 2048), (3520, 2048, 256), (8, 256, 256),
                                   (48, 256, 4), (320, 768, 768), (77, 64, 68)])
def test_linear_wgrad(hip, M, K, N):
    g = torch.Generator().manual_seed(M + N)
    x = bf(torch.randn(M, K, generator=g)); dy = bf(torch.randn(M, N, generator=g))
    dw = torch.zeros(N, K, device="cuda")
    hip.linear_wgrad(dy.cuda(), x.cuda(), dw)
    assert rel(dw, dy.float().T @ x.float()) < TOL_F32


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 5, 9])
@pytest.mark.parametrize("B,H,W,Ci,Co,k,s,p", [(2, 20, 24, 128, 128, 3, 2, 1), (3, 9, 9, 64, 64, 3, 1, 1), (2, 20, 20, 256, 512, 1, 2, 0),
                                               (1, 13, 17, 64, 128, 3, 2, 1), (2, 10, 14, 512, 128, 1, 1, 0), (3, 21, 19, 128, 64, 3, 1, 1)])
def test_conv_wgrad_dma_variants(hip, variant, B, H, W, Ci, Co, k, s, p):
    g = torch.Generator().manual_seed(variant * 100 + H)
    x = bf(torch.randn(B, Ci, H, W, generator=g)).float()
    w = torch.zeros(Co, Ci, k, k, requires_grad=True)
    y = F.conv2d(x, w, None, stride=s, padding=p)
    dy = bf(torch.randn(y.shape, generator=g))
    y.backward(dy.float())
    Ho, Wo = y.shape[-2:]
    scale = torch.rand(Co, generator=g) + 0.5
    dw = torch.zeros(Co, k, k, Ci, device="cuda"); db = torch.zeros(Co, device="cuda")
    geom = (B, H, W, Ci, Ho, Wo, Co, k, k, s, p)
    ref_dw = w.grad.permute(0, 2, 3, 1) * scale.view(-1, 1, 1, 1)
    # split partials through the workspace (plain stores + reduction pass), sole-writer read-modify-write (msplit 1),
    # and fp32 atomics (no workspace) must all accumulate the same gradient
    for msplit, ws in ((0, True), (1, True), (3, True), (3, False), (0, False)):
        dw.fill_(1.0); db.zero_()
        hip.conv_wgrad(nhwc(dy).cuda(), nhwc(x).bfloat16().cuda(), dw, geom=geom, scale=scale.cuda(), dbias=db, msplit=msplit,
                       variant=variant, workspace=ws)
        assert rel(dw - 1.0, ref_dw) < 5 * TOL_F32, (msplit, ws)
        assert rel(db, dy.float().sum((0, 2, 3))) < 1e-5


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 5, 9])
@pytest.mark.parametrize("M,K,N", [(200, 256, 256), (3520, 256, 2048), (3520, 2048, 256), (320, 768, 768), (77, 64, 72), (1000, 128, 64)])
def test_linear_wgrad_dma_variants(hip, variant, M, K, N):
    g = torch.Generator().manual_seed(M + N + variant)
    x = bf(torch.randn(M, K, generator=g)); dy = bf(torch.randn(M, N, generator=g))
    dw = torch.zeros(N, K, device="cuda"); db = torch.zeros(N, device="cuda")
    hip.linear_wgrad(dy.cuda(), x.cuda(), dw, dbias=db, variant=variant)
    assert rel(dw, dy.float().T @ x.float()) < TOL_F32
    assert rel(db, dy.float().sum(0)) < 1e-5


def test_grouped_wgrad_equals_individual_launches(hip):
    """rt_conv_wgrad_grouped: a mix of groupable Linear weight gradients (incl. > 24 of them, ragged M, split and unsplit)
    and non-groupable ones gives the same dw / dbias as one rt_conv_wgrad call each."""
    g = torch.Generator().manual_seed(11)
    shapes = [(3520, 256, 256), (3520, 256, 2048), (3520, 2048, 256), (320, 768, 768), (320, 768, 3072), (48, 256, 256),
              (3520, 256, 512), (1000, 128, 136), (200, 256, 64)] * 3 + [(77, 64, 72)]
    batch = hip.WgradBatch(workspace_mb=64)
    refs = []
    for i, (M, K, N) in enumerate(shapes):
        x = bf(torch.randn(M, K, generator=g)).cuda(); dy = bf(torch.randn(M, N, generator=g)).cuda()
        dw = torch.full((N, K), 0.5, device="cuda"); db = torch.zeros(N, device="cuda")
        dw2 = torch.full((N, K), 0.5, device="cuda"); db2 = torch.zeros(N, device="cuda")
        hip.linear_wgrad(dy, x, dw2, dbias=db2)
        batch.add(dy, x, dw, db)
        refs.append((dw, db, dw2, db2, dy, x))
    batch.run()
    for dw, db, dw2, db2, dy, x in refs:
        assert rel(dw, dw2) < 2e-6 and rel(db, db2) < 1e-6
        assert rel(dw - 0.5, dy.float().T @ x.float()) < TOL_F32


def test_errors_are_loud(hip):
    x = torch.zeros(4, 48, dtype=torch.bfloat16, device="cuda")      # K not a multiple of 64
    w = torch.zeros(8, 48, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(RuntimeError):
        hip.linear(x, w)
    with pytest.raises(RuntimeError):
        hip.linear(x.cpu(), w.cpu())


@pytest.mark.parametrize("M,K,N", [(200, 256, 256), (3520, 256, 2048), (8, 256, 256), (320, 768, 64)])
def test_wgrad_fused_bias_grad(hip, M, K, N):
    g = torch.Generator().manual_seed(M + 3 * N)
    x = bf(torch.randn(M, K, generator=g)); dy = bf(torch.randn(M, N, generator=g))
    dw = torch.zeros(N, K, device="cuda"); db = torch.ones(N, device="cuda")
    hip.linear_wgrad(dy.cuda(), x.cuda(), dw, dbias=db)
    assert rel(dw, dy.float().T @ x.float()) < TOL_F32
    assert rel(db, 1 + dy.float().sum(0)) < 1e-5


def test_weight_prep_batched(hip):
    g = torch.Generator().manual_seed(12)
    # 16-B path (C and N multiples of 8), mixed (one of them), element-wise path, several tiles per job, ragged tiles
    jobs = [(24, 9, 16, True), (100, 1, 70, False), (4, 1, 256, False), (33, 4, 33, True), (768, 1, 3072, False),
            (136, 9, 72, True), (72, 1, 100, True), (100, 1, 72, False)]
    batch = hip.WeightPrepBatch("cuda")
    keep = []
    for N, T, C, scaled in jobs:
        src = torch.randn(N, T, C, generator=g)
        sc = (torch.rand(N, generator=g) + 0.5) if scaled else None
        dst = torch.zeros(N, T, C, dtype=torch.bfloat16, device="cuda"); dst_t = torch.zeros(C, T, N, dtype=torch.bfloat16, device="cuda")
        sc_c = sc.cuda() if scaled else None
        batch.add(src.cuda(), N, T, C, scale=sc_c, dst=dst, dst_t=dst_t)
        keep.append((src, sc, dst, dst_t))
    batch.run()
    for src, sc, dst, dst_t in keep:
        ref = (src * sc.view(-1, 1, 1) if sc is not None else src).bfloat16()
        assert torch.equal(dst.cpu(), ref) and torch.equal(dst_t.cpu(), ref.permute(2, 1, 0).contiguous())


def test_grouped_gemm_equals_single_launches(hip):
    """rt_conv_gemm_grouped: the same kernel body behind one launch -- bit-identical to n single launches, for every epilogue
    the grouped call sites use (bias, fp32 in-place accumulation, ragged M / N); mixes it cannot group fall back to singles."""
    torch.manual_seed(3)
    jobs = []
    for M, K, N, acc in ((3520, 256, 512, False), (3520, 256, 256, False), (3333, 512, 256, True), (440, 256, 264, True), (100, 768, 64, False)):
        x = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
        b = torch.randn(N, device="cuda"); r = torch.randn(M, N, device="cuda") if acc else None
        jobs.append((x, w, b, r))

    def run(group):
        outs = []
        for x, w, b, r in jobs:
            if r is None:
                ob, of = hip.linear(x, w, bias=b, out_bf16=True, out_f32=True, group=group)
            else:
                buf = r.clone()
                ob, of = hip.linear(x, w, bias=b, res_f32=buf, out_bf16=False, out_f32=buf, group=group)
            outs.append((ob, of))
        if group is not None:
            group.run()
        torch.cuda.synchronize()
        return outs
    single, grouped = run(None), run(hip.GemmGroup())
    for (a16, a32), (b16, b32) in zip(single, grouped):
        assert torch.equal(a32, b32) and (a16 is None or torch.equal(a16, b16))
    # not groupable (a skinny product and a K >= 1024 product in the mix): still the right answers
    x = torch.randn(8, 256, device="cuda").bfloat16(); w = (torch.randn(256, 256, device="cuda") * 0.05).bfloat16()
    x2 = torch.randn(512, 2048, device="cuda").bfloat16(); w2 = (torch.randn(256, 2048, device="cuda") * 0.02).bfloat16()
    g = hip.GemmGroup()
    _, a = hip.linear(x, w, out_bf16=False, out_f32=True, group=g); _, b = hip.linear(x2, w2, out_bf16=False, out_f32=True, group=g)
    g.run()
    _, a0 = hip.linear(x, w, out_bf16=False, out_f32=True); _, b0 = hip.linear(x2, w2, out_bf16=False, out_f32=True)
    assert torch.equal(a, a0) and torch.equal(b, b0)


def test_wgrad_v2_grouped_policy_matches_torch(hip):
    """Second-generation weight gradients (csrc/rt_wgrad2.hip) as the training step uses them: ONE rt_conv_wgrad_grouped call
    over a stage-like mix -- 1x1 / 3x3 / stride-2 convolutions with a FrozenBN scale, Linears with a fused bias gradient, every
    tile configuration (256x256, 128x256, 256x128, 128x128), ragged channel counts and row counts, a group small enough that
    the m axis is split (workspace + reduction) -- against torch fp32; dw is accumulated into a non-zero buffer."""
    g = torch.Generator().manual_seed(21)
    batch = hip.WgradBatch(workspace_mb=256)
    checks = []
    convs = [(2, 40, 40, 256, 1024, 1, 1, 0), (2, 40, 40, 1024, 256, 1, 1, 0), (2, 40, 40, 256, 256, 3, 1, 1), (2, 41, 39, 128, 128, 3, 2, 1),
             (2, 40, 40, 512, 1024, 1, 2, 0), (1, 80, 80, 128, 512, 1, 1, 0), (1, 80, 80, 512, 128, 1, 1, 0), (2, 23, 19, 192, 320, 3, 1, 1)]
    for B, H, W, Ci, Co, k, s, p in convs:
        x = bf(torch.randn(B, Ci, H, W, generator=g)).float()
        w = torch.zeros(Co, Ci, k, k, requires_grad=True)
        y = F.conv2d(x, w, None, stride=s, padding=p)
        dy = bf(torch.randn(y.shape, generator=g))
        y.backward(dy.float())
        Ho, Wo = y.shape[-2:]
        scale = torch.rand(Co, generator=g) + 0.5
        dw = torch.full((Co, k, k, Ci), 0.25, device="cuda")
        batch.add_conv(nhwc(dy).cuda(), nhwc(x).bfloat16().cuda(), dw, (B, H, W, Ci, Ho, Wo, Co, k, k, s, p), scale=scale.cuda())
        checks.append((dw, w.grad.permute(0, 2, 3, 1) * scale.view(-1, 1, 1, 1), None, None))
    for M, K, N in [(320, 768, 2304), (3520, 2048, 256), (3520, 256, 512), (333, 64, 72), (20000, 256, 256)]:
        x = bf(torch.randn(M, K, generator=g)); dy = bf(torch.randn(M, N, generator=g))
        dw = torch.full((N, K), 0.25, device="cuda"); db = torch.full((N,), 2.0, device="cuda")
        batch.add(dy.cuda(), x.cuda(), dw, db)
        checks.append((dw, dy.float().T @ x.float(), db, dy.float().sum(0)))
    batch.run()
    for dw, ref, db, ref_b in checks:
        assert rel(dw - 0.25, ref) < TOL_F32, (tuple(dw.shape), rel(dw - 0.25, ref))
        if db is not None:
            assert rel(db - 2.0, ref_b) < 1e-5


def test_wgrad_v2_single_problem_splits_the_row_axis(hip):
    """A group of one: few tiles, many rows -> the row axis is cut into ~256 workgroups whose partial tiles meet in the reduction."""
    g = torch.Generator().manual_seed(22)
    for M, K, N in [(51200, 128, 128), (12800, 256, 1024), (100000, 64, 64)]:
        x = bf(torch.randn(M, K, generator=g)); dy = bf(torch.randn(M, N, generator=g))
        dw = torch.full((N, K), -1.0, device="cuda"); db = torch.zeros(N, device="cuda")
        hip.linear_wgrad(dy.cuda(), x.cuda(), dw, dbias=db)
        assert rel(dw + 1.0, dy.float().T @ x.float()) < TOL_F32
        assert rel(db, dy.float().sum(0)) < 1e-5


@pytest.mark.parametrize("B,H,W,Ci,Co", [
    (2, 40, 40, 256, 256),       # layer3: two channel tiles each side
    (1, 80, 80, 128, 128),       # layer2: rows of 80 (a 32-row chunk crosses an image row 2 times out of 5)
    (2, 20, 20, 512, 512),       # layer4: every chunk holds one or two row ends
    (3, 13, 9, 64, 72),          # W < 32: several image rows per chunk; odd sizes; chunks cross IMAGE boundaries; ragged channels
    (2, 23, 19, 192, 320),       # ragged tiles on both sides
    (1, 8, 100, 128, 64),        # W > 32 with few rows: the first / last image row masks cover whole chunks
])
@pytest.mark.parametrize("overwrite", [False, True])
def test_wgrad_v2_fused_taps_3x3(hip, B, H, W, Ci, Co, overwrite):
    """3x3 / stride 1 / pad 1 weight gradients on the tap-fused tile (three kw taps of a kernel row per workgroup, x staged as one
    34-row window, border terms removed by row masks) against torch fp32, alone (row axis split) and inside a group (unsplit)."""
    g = torch.Generator().manual_seed(B * 1000 + W)
    x = bf(torch.randn(B, Ci, H, W, generator=g)).float()
    w = torch.zeros(Co, Ci, 3, 3, requires_grad=True)
    y = F.conv2d(x, w, None, stride=1, padding=1)
    dy = bf(torch.randn(y.shape, generator=g))
    y.backward(dy.float())
    ref = w.grad.permute(0, 2, 3, 1)
    geom = (B, H, W, Ci, H, W, Co, 3, 3, 1, 1)
    dyc, xc = nhwc(dy).cuda(), nhwc(x).bfloat16().cuda()
    base = 0.0 if overwrite else 0.5
    dw = torch.full((Co, 3, 3, Ci), 7.0 if overwrite else base, device="cuda")
    hip.conv_wgrad(dyc, xc, dw, geom=geom, overwrite=overwrite)
    e1 = rel(dw - base, ref)
    # inside a group with enough other tile tasks that nothing is split
    batch = hip.WgradBatch(workspace_mb=256)
    dw2 = torch.full((Co, 3, 3, Ci), 7.0 if overwrite else base, device="cuda")
    batch.add_conv(dyc, xc, dw2, geom, overwrite=overwrite)
    keep = []
    for _ in range(3):
        xa = torch.randn(2048, 1024, device="cuda").bfloat16(); da = torch.randn(2048, 1024, device="cuda").bfloat16()
        dwa = torch.zeros(1024, 1024, device="cuda"); keep.append((xa, da, dwa))
        batch.add(da, xa, dwa, None)
    batch.run()
    e2 = rel(dw2 - base, ref)
    print("fused taps", (B, H, W, Ci, Co), "single", e1, "grouped", e2)
    assert e1 < TOL_F32 and e2 < TOL_F32
    for xa, da, dwa in keep:
        assert rel(dwa, da.float().T @ xa.float()) < TOL_F32


@pytest.mark.parametrize("B,H,W,Ci,Co,dil", [(2, 12, 16, 512, 512, 2), (1, 13, 9, 64, 128, 2), (2, 20, 20, 128, 64, 3)])
def test_dilated_conv_fwd_dgrad_wgrad(hip, B, H, W, Ci, Co, dil):
    """`dil` of rt_conv_gemm / rt_conv_wgrad (--dilation: layer4's 3x3 convolutions, stride 1, padding = dilation) against
    torch fp32: forward gather, transposed (backward-data) gather and the weight-gradient gather (both generations' paths)."""
    g = torch.Generator().manual_seed(B * 100 + H + Ci + dil)
    x = bf(torch.randn(B, Ci, H, W, generator=g)).float().requires_grad_(True)
    w = bf(torch.randn(Co, Ci, 3, 3, generator=g) / (Ci * 9) ** 0.5).float().requires_grad_(True)
    bias = torch.randn(Co, generator=g)
    y = F.conv2d(x, w, bias, stride=1, padding=dil, dilation=dil)
    assert y.shape[-2:] == (H, W)
    dy = bf(torch.randn(y.shape, generator=g))
    y.backward(dy.float())
    x_nhwc = nhwc(x.detach()).bfloat16().cuda()
    w_k = w.detach().permute(0, 2, 3, 1).contiguous().bfloat16().cuda()
    geom = (B, H, W, Ci, H, W, Co, 3, 3, 1, dil)
    ob, of = hip.conv_gemm(x_nhwc, w_k, geom=geom, bias=bias.cuda(), out_bf16=True, out_f32=True, dil=dil)
    ref = nhwc(y.detach()).reshape(-1, Co)
    assert rel(of, ref) < TOL_F32 and rel(ob, ref) < TOL_BF16
    w_t = w.detach().permute(1, 2, 3, 0).contiguous().bfloat16().cuda()
    dy_nhwc = nhwc(dy).cuda()
    _, dxf = hip.conv_gemm(dy_nhwc, w_t, geom=(B, H, W, Co, H, W, Ci, 3, 3, 1, dil), transposed=True, out_bf16=False, out_f32=True, dil=dil)
    assert rel(dxf, nhwc(x.grad).reshape(-1, Ci)) < TOL_F32
    dw = torch.zeros(Co, 3, 3, Ci, device="cuda")
    hip.conv_wgrad(dy_nhwc.view(-1, Co), x_nhwc, dw, geom=geom, dil=dil)
    assert rel(dw, w.grad.permute(0, 2, 3, 1)) < 5e-5
    # the grouped second-generation launch (what the training step uses)
    dw2 = torch.zeros(Co, 3, 3, Ci, device="cuda")
    batch = hip.WgradBatch(workspace_mb=64)
    batch.add_conv(dy_nhwc.view(-1, Co), x_nhwc, dw2, geom, dil=dil)
    batch.run()
    assert rel(dw2, w.grad.permute(0, 2, 3, 1)) < 5e-5
    # a hint that names a register-staged tile has no dilation: refused, not silently wrong
    with pytest.raises(RuntimeError):
        hip.conv_gemm(x_nhwc, w_k, geom=geom, bias=bias.cuda(), tile_hint=3, dil=dil)
