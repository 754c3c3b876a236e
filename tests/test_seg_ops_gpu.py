"""GPU parity of the RES-head kernels (csrc/rt_seg.hip) against plain torch fp32 references of the same ops
(reference call sites: models/reftr_segmentation.py:196-208, 240-280, 314-337)."""
import pytest
import torch
import torch.nn.functional as F

from test_gemm_gpu import bf, rel

pytestmark = pytest.mark.gpu


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("B,H,W,C,ld,act", [(2, 5, 7, 520, 576, 1), (3, 12, 9, 16, 64, 1), (2, 6, 6, 128, 128, 0), (1, 20, 20, 32, 64, 1)])
def test_gn_nhwc_fwd_bwd(hip, B, H, W, C, ld, act):
    g = torch.Generator().manual_seed(C + H)
    x = torch.randn(B, C, H, W, generator=g, requires_grad=True)
    gam = (torch.rand(C, generator=g) + 0.5).requires_grad_(True); bet = (torch.randn(C, generator=g) * 0.2).requires_grad_(True)
    y = F.group_norm(x, 8, gam, bet, 1e-5)
    if act:
        y = F.relu(y)
    dy = torch.randn(B, C, H, W, generator=g)
    y.backward(dy)
    xp = torch.zeros(B * H * W, ld); xp[:, :C] = nhwc(x.detach()).reshape(-1, C)
    xp[:, C:] = 7.0                                                   # padding content of x must not matter
    yb, stats = hip.gn_nhwc_fwd(xp.cuda(), gam.detach().cuda(), bet.detach().cuda(), B, H * W, C, 8, ldy=ld, act=act)
    assert rel(yb[:, :C], nhwc(y.detach()).reshape(-1, C)) < 3e-3
    assert float(yb[:, C:].float().abs().max()) == 0.0 if ld > C else True
    dyp = torch.zeros(B * H * W, ld); dyp[:, :C] = nhwc(dy).reshape(-1, C)
    dg = torch.zeros(C, device="cuda"); db = torch.zeros(C, device="cuda")
    dx = hip.gn_nhwc_bwd(dyp.cuda(), xp.cuda(), gam.detach().cuda(), bet.detach().cuda(), stats, dg, db, B, H * W, C, 8, lddx=ld, act=act)
    assert rel(dx[:, :C], nhwc(x.grad).reshape(-1, C)) < 4e-3
    assert rel(dg, gam.grad) < 1e-4 and rel(db, bet.grad) < 1e-4
    if ld > C:
        assert float(dx[:, C:].float().abs().max()) == 0.0


@pytest.mark.parametrize("B,h,w,H,W,C,ld", [(2, 5, 7, 10, 14, 128, 128), (1, 13, 9, 25, 18, 32, 64), (2, 4, 4, 8, 8, 64, 64)])
def test_upsample_add_fwd_bwd(hip, B, h, w, H, W, C, ld):
    g = torch.Generator().manual_seed(H * W)
    a = bf(torch.randn(B, C, h, w, generator=g)).float().requires_grad_(True)
    fpn = torch.randn(B, C, H, W, generator=g)
    y = fpn + F.interpolate(a, size=(H, W), mode="nearest")
    dy = torch.randn(B, C, H, W, generator=g)
    y.backward(dy)
    ap = torch.zeros(B * h * w, ld); ap[:, :C] = nhwc(a.detach()).reshape(-1, C)
    out = hip.upsample_add(nhwc(fpn).reshape(-1, C).cuda(), ap.bfloat16().cuda(), B, H, W, h, w, C, ldo=ld)
    assert rel(out[:, :C], nhwc(y.detach()).reshape(-1, C)) < 3e-3
    dyp = torch.zeros(B * H * W, ld); dyp[:, :C] = nhwc(dy).reshape(-1, C)
    da, dyb = hip.upsample_add_bwd(dyp.cuda(), B, H, W, h, w, C, ldda=ld, lddyb=ld)
    assert rel(da[:, :C], nhwc(a.grad).reshape(-1, C)) < 1e-5
    assert rel(dyb[:, :C], dyp[:, :C]) < 3e-3


@pytest.mark.parametrize("B,h,w,L", [(2, 5, 7, 3), (3, 20, 20, 40)])
def test_attn_map_fwd_bwd(hip, B, h, w, L):
    E, nh = 256, 8
    g = torch.Generator().manual_seed(h * w)
    HW, S = h * w, L + h * w
    q = torch.randn(B, E, generator=g, requires_grad=True)
    k_all = torch.randn(B * S, E, generator=g).mul(0.3).requires_grad_(True)
    mask = torch.zeros(B, h, w, dtype=torch.bool); mask[1, :, w - 2:] = True; mask[1, h - 1:, :] = True
    k = k_all.view(B, S, E)[:, L:].reshape(B, h, w, nh, E // nh)
    logits = torch.einsum("bnc,bhwnc->bnhw", q.view(B, nh, E // nh) * float(E / nh) ** -0.5, k)
    logits = logits.masked_fill(mask[:, None], float("-inf"))
    P = F.softmax(logits.flatten(1), dim=-1).view(B, nh, HW)
    dP = torch.randn(B, nh, HW, generator=g)
    (P * dP).sum().backward()
    concat = torch.full((B * HW, 576), 3.0, dtype=torch.bfloat16, device="cuda")
    Pg = hip.attn_map_fwd(q.detach().cuda(), k_all.detach().cuda(), mask.reshape(B, HW).to(torch.uint8).cuda(), B, HW, E, nh, S, L,
                          concat=concat, concat_col=512)
    assert rel(Pg, P.detach()) < 1e-5
    assert rel(concat[:, 512:520].float().view(B, HW, nh), P.detach().permute(0, 2, 1)) < 3e-3
    assert float((concat[:, :512].float() - 3.0).abs().max()) == 0.0
    dcon = torch.zeros(B * HW, 576); dcon[:, 512:520] = dP.permute(0, 2, 1).reshape(B * HW, nh)
    dq, dk = hip.attn_map_bwd(q.detach().cuda(), k_all.detach().cuda(), Pg, dcon.cuda(), B, HW, E, nh, S, L, 512)
    assert rel(dq, q.grad) < 1e-4 and rel(dk, k_all.grad) < 1e-4


def test_seg_concat(hip):
    B, HW, E, nh, L = 2, 35, 256, 8, 5
    g = torch.Generator().manual_seed(3)
    src = torch.randn(B * HW, E, generator=g); mem = torch.randn(B * (L + HW), E, generator=g)
    out = torch.full((B * HW, 576), 9.0, dtype=torch.bfloat16, device="cuda")
    hip.seg_concat(src.cuda(), mem.cuda(), out, B, HW, E, nh, L + HW, L, src_dense=True)
    o = out.float().cpu()
    assert torch.equal(o[:, :E], src.bfloat16().float())
    assert torch.equal(o[:, E:2 * E], mem.view(B, L + HW, E)[:, L:].reshape(B * HW, E).bfloat16().float())
    assert float((o[:, 2 * E:2 * E + nh] - 9.0).abs().max()) == 0.0 and float(o[:, 2 * E + nh:].abs().max()) == 0.0


@pytest.mark.parametrize("B,h,w,Ht,Wt", [(2, 24, 32, 96, 128), (3, 10, 13, 37, 50)])
def test_mask_loss_fwd_bwd(hip, B, h, w, Ht, Wt):
    from oracle import reftr_oracle as O          # checker only
    g = torch.Generator().manual_seed(Ht)
    pred = torch.randn(B, 1, h, w, generator=g).mul(2).requires_grad_(True)
    tgt = torch.rand(B, 1, Ht, Wt, generator=g) > 0.6
    src = F.interpolate(pred, size=(Ht, Wt), mode="bilinear", align_corners=False).view(B, -1)
    lf = O.sigmoid_focal_loss(src, tgt.float().view(B, -1), B); ld = O.dice_loss(src, tgt.float().view(B, -1), B)
    (0.7 * lf + 1.3 * ld).backward()
    pp = torch.zeros(B * h * w, 4); pp[:, 0] = pred.detach().reshape(-1)
    losses, sums = hip.mask_loss(pp.cuda(), tgt.to(torch.uint8).cuda().contiguous(), B, h, w, Ht, Wt, 4, float(B))
    assert abs(float(losses[0]) - float(lf)) < 2e-5 * max(1.0, abs(float(lf))) and abs(float(losses[1]) - float(ld)) < 2e-5
    dpred = torch.zeros(B * h * w, 64, device="cuda")
    hip.mask_loss(pp.cuda(), tgt.to(torch.uint8).cuda().contiguous(), B, h, w, Ht, Wt, 4, float(B), sums=sums, dpred=dpred,
                  g_focal=torch.tensor([0.7], device="cuda"), g_dice=torch.tensor([1.3], device="cuda"))
    assert rel(dpred[:, 0], pred.grad.reshape(-1)) < 2e-4
    assert float(dpred[:, 1:].abs().max()) == 0.0
