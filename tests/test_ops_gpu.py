"""GPU parity of the non-GEMM kernels against plain torch fp32 references of the same op (floating point)
or exact integer/bool expectations (masks, selection)."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from test_gemm_gpu import bf, hash_keep, rel

pytestmark = pytest.mark.gpu


def cu(t):
    return t.cuda() if t is not None else None


# ------------------------------------------------------------------ LayerNorm
@pytest.mark.parametrize("M,D,act", [(37, 256, 0), (320, 768, 0), (5, 256, 1), (64, 100, 1)])
def test_layernorm_fwd_bwd(hip, M, D, act):
    g = torch.Generator().manual_seed(M + D)
    x = torch.randn(M, D, generator=g, requires_grad=True)
    gam = torch.rand(D, generator=g) + 0.5; bet = torch.randn(D, generator=g) * 0.1
    gam.requires_grad_(True); bet.requires_grad_(True)
    pos = torch.randn(M, D, generator=g)
    y = F.layer_norm(x, (D,), gam, bet, 1e-5)
    if act:
        y = F.relu(y)
    dy = torch.randn(M, D, generator=g); dy2 = torch.randn(M, D, generator=g)
    y.backward(dy + dy2)
    yf, yb, ypb, mean, rstd = hip.layernorm_fwd(cu(x.detach()), cu(gam.detach()), cu(bet.detach()), 1e-5, act=act, pos=cu(pos))
    assert rel(yf, y) < 1e-5 and rel(yb, y) < 3e-3 and rel(ypb, y + pos) < 3e-3
    dgam = torch.zeros(D, device="cuda"); dbet = torch.zeros(D, device="cuda")
    dxf, dxb = hip.layernorm_bwd(cu(dy), cu(x.detach()), cu(gam.detach()), cu(bet.detach()), mean, rstd, dgam, dbet,
                                 dy2=cu(dy2), act=act)
    assert rel(dxf, x.grad) < 2e-5 and rel(dxb, x.grad) < 3e-3
    assert rel(dgam, gam.grad) < 2e-5 and rel(dbet, bet.grad) < 2e-5


def test_layernorm_rowmap_and_dropout(hip):
    g = torch.Generator().manual_seed(3)
    B, L, S, D = 3, 5, 12, 256
    x = torch.randn(B * L, D, generator=g)
    gam = torch.rand(D, generator=g) + 0.5; bet = torch.randn(D, generator=g) * 0.1
    seq = torch.zeros(B * S, D, device="cuda")
    p, seed = 0.1, 77
    hip.layernorm_fwd(cu(x), cu(gam), cu(bet), 1e-5, act=1, drop_p=p, drop_seed=seed, y_f32=seq, want_bf16=False,
                      rowmap=(L, S, 2))
    keep = torch.from_numpy(hash_keep(seed, np.arange(B * L * D, dtype=np.uint64), p).reshape(B * L, D))
    ref = F.relu(F.layer_norm(x, (D,), gam, bet, 1e-5)) * keep / (1 - np.float32(p))
    got = seq.cpu().view(B, S, D)[:, 2:2 + L].reshape(B * L, D)
    assert rel(got, ref) < 1e-5
    assert float(seq.cpu().view(B, S, D)[:, :2].abs().max()) == 0.0
    # backward: drop2 mask on the bf16 output
    dy = torch.randn(B * S, D, generator=g)
    xg = x.clone().requires_grad_(True)
    (F.relu(F.layer_norm(xg, (D,), gam, bet, 1e-5)) * keep / (1 - np.float32(p))).backward(dy.view(B, S, D)[:, 2:2 + L].reshape(B * L, D))
    yf, _, _, mean, rstd = hip.layernorm_fwd(cu(x), cu(gam), cu(bet), 1e-5, act=1, drop_p=p, drop_seed=seed, want_bf16=False)
    dxf, dxb = hip.layernorm_bwd(cu(dy), cu(x), cu(gam), cu(bet), mean, rstd, None, None, act=1, drop_p=p, drop_seed=seed,
                                 drop2_p=0.2, drop2_seed=5, rowmap=(L, S, 2))
    assert rel(dxf, xg.grad) < 2e-5
    keep2 = torch.from_numpy(hash_keep(5, np.arange(B * L * D, dtype=np.uint64), 0.2).reshape(B * L, D))
    assert rel(dxb, xg.grad * keep2 / (1 - np.float32(0.2))) < 3e-3


def test_groupnorm_fwd_bwd(hip):
    g = torch.Generator().manual_seed(9)
    B, HW, C, G, L = 3, 12, 256, 32, 4
    S = L + HW
    x = torch.randn(B, HW, C, generator=g, requires_grad=True)
    gam = (torch.rand(C, generator=g) + 0.5).requires_grad_(True); bet = (torch.randn(C, generator=g) * 0.1).requires_grad_(True)
    pos = torch.randn(B * S, C, generator=g)
    y = F.group_norm(x.permute(0, 2, 1), G, gam, bet, 1e-5).permute(0, 2, 1)     # [B, HW, C]
    dy = torch.randn(B, S, C, generator=g)
    y.backward(dy[:, L:])
    yf = torch.zeros(B * S, C, device="cuda"); yb = torch.zeros(B * S, C, device="cuda", dtype=torch.bfloat16)
    ypb = torch.zeros_like(yb)
    stats = hip.groupnorm_fwd(cu(x.detach()), cu(gam.detach()), cu(bet.detach()), G, 1e-5, y_f32=yf, y_bf16=yb, pos=cu(pos),
                              ypos_bf16=ypb, rows_per_img=S, row_off=L)
    assert rel(yf.view(B, S, C)[:, L:], y) < 1e-5
    assert rel(ypb.view(B, S, C)[:, L:], y + pos.view(B, S, C)[:, L:]) < 3e-3
    dg = torch.zeros(C, device="cuda"); db = torch.zeros(C, device="cuda")
    dxf, dxb = hip.groupnorm_bwd(cu(dy.reshape(B * S, C)), cu(x.detach()), cu(gam.detach()), stats, dg, db, G, 1e-5,
                                 rows_per_img=S, row_off=L, want_f32=True)
    assert rel(dxf, x.grad) < 3e-5 and rel(dxb, x.grad) < 3e-3
    assert rel(dg, gam.grad) < 3e-5 and rel(db, bet.grad) < 3e-5


# ------------------------------------------------------------------ attention
def ref_attn(q, k, v, kpm, B, H, Sq, Sk, dh, scale):
    qh = q.view(B, Sq, H, dh).transpose(1, 2); kh = k.view(B, Sk, H, dh).transpose(1, 2); vh = v.view(B, Sk, H, dh).transpose(1, 2)
    s = qh @ kh.transpose(-1, -2) * scale
    if kpm is not None:
        s = s.masked_fill(kpm[:, None, None, :].bool(), float("-inf"))
    return (s.softmax(-1) @ vh).transpose(1, 2).reshape(B * Sq, H * dh)


@pytest.mark.parametrize("B,H,Sq,Sk,dh", [(2, 8, 440, 440, 32), (3, 12, 40, 40, 64), (2, 8, 1, 440, 32),
                                           (2, 8, 5, 5, 32), (1, 8, 130, 715, 32),
                                           # inner axes that do not fit the CU's LDS at once: walked in chunks (--dilation at 640 x 640: S = 1640)
                                           (2, 8, 1640, 1640, 32), (1, 8, 1, 1640, 32), (1, 8, 70, 1000, 32), (1, 12, 600, 600, 64),
                                           (1, 8, 900, 33, 32)])
def test_attention_fwd_bwd(hip, B, H, Sq, Sk, dh):
    g = torch.Generator().manual_seed(Sq + Sk)
    E = H * dh
    q = bf(torch.randn(B * Sq, E, generator=g)).float().requires_grad_(True)
    kv = bf(torch.randn(B * Sk, 2 * E, generator=g)).float().requires_grad_(True)   # packed [k | v]
    kpm = torch.zeros(B, Sk, dtype=torch.uint8)
    for b in range(B):
        kpm[b, Sk - (b + 1) * max(1, Sk // 7):] = 1
    if Sk > 3:
        kpm[0, 1] = 1
    scale = dh ** -0.5
    out = ref_attn(q, kv[:, :E], kv[:, E:], kpm, B, H, Sq, Sk, dh, scale)
    do = bf(torch.randn(B * Sq, E, generator=g))
    out.backward(do.float())
    qc = q.detach().bfloat16().cuda(); kvc = kv.detach().bfloat16().cuda()
    o, lse = hip.attn_fwd(qc, kvc[:, :E], kvc[:, E:], kpm.cuda(), B=B, H=H, Sq=Sq, Sk=Sk, dh=dh, scale=scale)
    assert rel(o, out) < 4e-3
    dkv = torch.zeros(B * Sk, 2 * E, dtype=torch.bfloat16, device="cuda")
    dq, _, _ = hip.attn_bwd(qc, kvc[:, :E], kvc[:, E:], o, do.cuda(), lse, kpm.cuda(), B=B, H=H, Sq=Sq, Sk=Sk, dh=dh,
                            scale=scale, dk=dkv[:, :E], dv=dkv[:, E:])
    assert rel(dq, q.grad) < 1e-2
    assert rel(dkv, kv.grad) < 1e-2


_CHUNK_SCRIPT = r"""
import sys, torch
sys.path.insert(0, sys.argv[1])
from reftr_amd import hip
B, H, Sq, Sk, dh = 2, 8, 200, 600, 32
E = H * dh
g = torch.Generator().manual_seed(3)
q = torch.randn(B * Sq, E, generator=g).bfloat16().cuda(); k = torch.randn(B * Sk, E, generator=g).bfloat16().cuda()
v = torch.randn(B * Sk, E, generator=g).bfloat16().cuda(); do = torch.randn(B * Sq, E, generator=g).bfloat16().cuda()
kpm = torch.zeros(B, Sk, dtype=torch.uint8); kpm[1, 500:] = 1; kpm[0, ::5] = 1
hip.set_seed_dev(None)
o, lse = hip.attn_fwd(q, k, v, kpm.cuda(), B=B, H=H, Sq=Sq, Sk=Sk, dh=dh, scale=dh ** -0.5, drop_p=0.1, drop_seed=77)
dq, dk, dv = hip.attn_bwd(q, k, v, o, do, lse, kpm.cuda(), B=B, H=H, Sq=Sq, Sk=Sk, dh=dh, scale=dh ** -0.5, drop_p=0.1, drop_seed=77)
torch.cuda.synchronize()
torch.save({"o": o.cpu(), "lse": lse.cpu(), "dq": dq.cpu(), "dk": dk.cpu(), "dv": dv.cpu()}, sys.argv[2])
"""


def test_attention_long_axis_kernels_are_bit_identical_to_the_whole_axis_kernels(hip, tmp_path):
    """REFTR_ATTN_CHUNK forces the chunked kernels (inner axis staged 64 rows at a time) on a shape the whole-axis kernels handle
    (Sk = 600: the two-pass forward, the fused backward): same lane <-> element assignment, same key order per lane, dropout and
    key-padding mask included -> identical bits.  One process per setting (the library reads the variable once)."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for chunk in ("0", "64", "224"):
        f = tmp_path / f"attn_{chunk}.pt"
        env = dict(os.environ); env["REFTR_ATTN_CHUNK"] = chunk
        subprocess.run([sys.executable, "-c", _CHUNK_SCRIPT, root, str(f)], check=True, env=env, timeout=600)
        outs.append(torch.load(f))
    for other in outs[1:]:
        for key in ("o", "lse", "dq", "dk", "dv"):
            a, b = outs[0][key], other[key]
            assert torch.equal(a.view(torch.int16) if a.dtype == torch.bfloat16 else a, b.view(torch.int16) if b.dtype == torch.bfloat16 else b), key


def test_attention_dropout_mask(hip):
    B, H, Sq, Sk, dh = 1, 8, 7, 9, 32
    g = torch.Generator().manual_seed(1)
    E = H * dh
    q = bf(torch.randn(B * Sq, E, generator=g)); k = bf(torch.randn(B * Sk, E, generator=g)); v = bf(torch.randn(B * Sk, E, generator=g))
    p, seed = 0.25, 99
    o, _ = hip.attn_fwd(q.cuda(), k.cuda(), v.cuda(), None, B=B, H=H, Sq=Sq, Sk=Sk, dh=dh, scale=dh ** -0.5, drop_p=p, drop_seed=seed)
    qh = q.float().view(B, Sq, H, dh).transpose(1, 2); kh = k.float().view(B, Sk, H, dh).transpose(1, 2)
    vh = v.float().view(B, Sk, H, dh).transpose(1, 2)
    P = (qh @ kh.transpose(-1, -2) * dh ** -0.5).softmax(-1)
    keep = torch.from_numpy(hash_keep(seed, np.arange(B * H * Sq * Sk, dtype=np.uint64), p).reshape(B, H, Sq, Sk))
    ref = ((P * keep / (1 - np.float32(p))) @ vh).transpose(1, 2).reshape(B * Sq, E)
    assert rel(o, ref) < 4e-3


def test_attention_single_query_kernels_with_dropout_and_mask(hip):
    """Sq = 1 (the decoder's cross-attention with one query per image) runs on its own fwd / bwd kernels: same definition,
    same dropout hash index ((b*H + h)*Sq + i)*Sk + j, key-padding mask, dq / dk / dv in one launch."""
    B, H, Sq, Sk, dh = 3, 8, 1, 440, 32
    g = torch.Generator().manual_seed(5)
    E = H * dh
    q = bf(torch.randn(B * Sq, E, generator=g)).float().requires_grad_(True)
    k = bf(torch.randn(B * Sk, E, generator=g)).float().requires_grad_(True)
    v = bf(torch.randn(B * Sk, E, generator=g)).float().requires_grad_(True)
    kpm = torch.zeros(B, Sk, dtype=torch.uint8); kpm[1, 300:] = 1; kpm[2, ::3] = 1
    p, seed, scale = 0.2, 4242, dh ** -0.5
    qh = q.view(B, Sq, H, dh).transpose(1, 2); kh = k.view(B, Sk, H, dh).transpose(1, 2); vh = v.view(B, Sk, H, dh).transpose(1, 2)
    S = qh @ kh.transpose(-1, -2) * scale
    S = S.masked_fill(kpm.bool()[:, None, None, :], float("-inf"))
    keep = torch.from_numpy(hash_keep(seed, np.arange(B * H * Sq * Sk, dtype=np.uint64), p).reshape(B, H, Sq, Sk))
    out = ((S.softmax(-1) * keep / (1 - np.float32(p))) @ vh).transpose(1, 2).reshape(B * Sq, E)
    do = bf(torch.randn(B * Sq, E, generator=g))
    out.backward(do.float())
    hip.set_seed_dev(None)
    o, lse = hip.attn_fwd(q.detach().bfloat16().cuda(), k.detach().bfloat16().cuda(), v.detach().bfloat16().cuda(), kpm.cuda(),
                          B=B, H=H, Sq=Sq, Sk=Sk, dh=dh, scale=scale, drop_p=p, drop_seed=seed)
    assert rel(o, out) < 4e-3
    assert rel(lse.view(B, H), torch.logsumexp(S.detach(), -1).view(B, H)) < 1e-5
    dq, dk, dv = hip.attn_bwd(q.detach().bfloat16().cuda(), k.detach().bfloat16().cuda(), v.detach().bfloat16().cuda(), o, do.cuda(),
                              lse, kpm.cuda(), B=B, H=H, Sq=Sq, Sk=Sk, dh=dh, scale=scale, drop_p=p, drop_seed=seed)
    assert rel(dq, q.grad) < 1e-2 and rel(dk, k.grad) < 1e-2 and rel(dv, v.grad) < 1e-2
    assert float(dk.float().view(B, Sk, E)[1, 300:].abs().max()) == 0.0          # masked keys: exact zeros


# ------------------------------------------------------------------ backbone-side
def test_stem_and_maxpool(hip):
    g = torch.Generator().manual_seed(2)
    B, H, W = 2, 70, 52
    img = bf(torch.randn(B, 3, H, W, generator=g)).float()
    w = bf(torch.randn(64, 3, 7, 7, generator=g) / 12).float()
    scale = torch.rand(64, generator=g) + 0.5; shift = torch.randn(64, generator=g) * 0.1
    wf = bf(w * scale.view(-1, 1, 1, 1)).float()
    ref = F.relu(F.conv2d(img, wf, None, 2, 3) + shift.view(1, -1, 1, 1))
    Ho, Wo, Hp, Wp = hip.stem_geometry(H, W)
    assert (Ho, Wo) == tuple(ref.shape[-2:])
    xp = hip.img_pack(img.cuda())
    wk = torch.empty(64, 7, 8, 4, dtype=torch.bfloat16, device="cuda")
    hip.stem_weight_prep(w.permute(0, 2, 3, 1).contiguous().cuda(), scale.cuda(), wk)
    y = hip.stem_conv(xp, wk, shift.cuda(), Ho, Wo)
    assert rel(y.permute(0, 3, 1, 2), ref) < 3e-3
    mp = hip.maxpool3x3s2(y)
    ref_mp = F.max_pool2d(y.float().cpu().permute(0, 3, 1, 2), 3, 2, 1)
    assert torch.equal(mp.float().cpu().permute(0, 3, 1, 2), ref_mp)          # max is exact


@pytest.mark.parametrize("B,H,W", [(2, 70, 52), (1, 64, 128), (3, 33, 47), (2, 130, 61), (1, 17, 16), (8, 640, 640)])
def test_stem_pool_in_one_launch_is_bit_identical(hip, B, H, W):
    """rt_stem_pool = rt_maxpool3x3s2(rt_stem_conv(x)) bit for bit: same products in the same K order, one bf16 rounding, and
    max commutes with it (ragged tiles, odd sizes, pool padding on every side, one full-size case)."""
    g = torch.Generator().manual_seed(B * 100 + H)
    img = torch.randn(B, 3, H, W, generator=g).cuda()
    w = (torch.randn(64, 7, 7, 3, generator=g) / 12).cuda()
    scale = (torch.rand(64, generator=g) + 0.5).cuda(); shift = (torch.randn(64, generator=g) * 0.3).cuda()
    Ho, Wo, Hp, Wp = hip.stem_geometry(H, W)
    xp = hip.img_pack(img)
    wk = torch.empty(64, 7, 8, 4, dtype=torch.bfloat16, device="cuda")
    hip.stem_weight_prep(w, scale, wk)
    two = hip.maxpool3x3s2(hip.stem_conv(xp, wk, shift, Ho, Wo))
    one = hip.stem_pool(xp, wk, shift, Ho, Wo)
    torch.cuda.synchronize()
    assert one.shape == two.shape
    assert torch.equal(one.view(torch.int16), two.view(torch.int16))


def test_weight_prep_and_bn_fold(hip):
    g = torch.Generator().manual_seed(4)
    N, T, C = 24, 9, 16
    src = torch.randn(N, T, C, generator=g); scale = torch.rand(N, generator=g) + 0.5
    dst = torch.empty(N, T, C, dtype=torch.bfloat16, device="cuda"); dst_t = torch.empty(C, T, N, dtype=torch.bfloat16, device="cuda")
    hip.weight_prep(src.cuda(), N, T, C, scale=scale.cuda(), dst=dst, dst_t=dst_t)
    ref = (src * scale.view(-1, 1, 1)).bfloat16()
    assert torch.equal(dst.cpu(), ref) and torch.equal(dst_t.cpu(), ref.permute(2, 1, 0).contiguous())
    w, b, rm, rv = (torch.rand(8) + 0.5, torch.randn(8), torch.randn(8), torch.rand(8) + 0.5)
    sc = torch.empty(8, device="cuda"); sh = torch.empty(8, device="cuda")
    hip.bn_fold(w.cuda(), b.cuda(), rm.cuda(), rv.cuda(), 1e-5, sc, sh)
    rs = w * (rv + 1e-5).rsqrt()
    assert rel(sc, rs) < 1e-6 and rel(sh, b - rm * rs) < 1e-6


def test_mask_posenc_against_golden(hip):
    import os
    gd = np.load(os.path.join(os.path.dirname(__file__), "golden", "sine_pos.npz"))
    mask = torch.from_numpy(gd["mask"])                      # [2, 5, 7] already at feature resolution
    B, h, w = mask.shape
    up = mask.repeat_interleave(32, 1).repeat_interleave(32, 2)   # full-res mask whose nearest downsample is `mask`
    L, C = 3, 256
    S = L + h * w
    kpm = torch.zeros(B, S, dtype=torch.uint8, device="cuda")
    pos = torch.zeros(B * S, C, device="cuda")
    add = torch.randn(C)
    hip.mask_posenc(up.to(torch.uint8).cuda(), h, w, C, add.cuda(), kpm, L, pos, S, L)
    assert torch.equal(kpm.cpu()[:, L:].bool(), mask.flatten(1))                 # bool: exact
    ref = torch.from_numpy(gd["pos"]).flatten(2).permute(0, 2, 1) + add          # [B, hw, C]
    got = pos.view(B, S, C)[:, L:].cpu()
    # Columns / rows that are entirely padding divide by (0 + 1e-6): arguments of ~3e6 rad whose sine is
    # ulp-sensitive.  Those positions are masked keys (never attended); compare every other position tightly.
    nm = ~mask
    col_ok = nm.any(1, keepdim=True).expand_as(mask); row_ok = nm.any(2, keepdim=True).expand_as(mask)
    ok = (col_ok & row_ok).flatten(1)
    assert rel(got[ok], ref[ok]) < 2e-5
    assert torch.isfinite(got).all()


# ------------------------------------------------------------------ small fused ops
def test_colsum_rows_add_embed(hip):
    g = torch.Generator().manual_seed(6)
    dy = torch.randn(333, 200, generator=g)
    db = torch.zeros(200, device="cuda")
    hip.colsum(dy.cuda(), db); hip.colsum(bf(dy).cuda(), db)
    assert rel(db, dy.sum(0) + bf(dy).float().sum(0)) < 1e-5
    B, L, S, D = 3, 4, 10, 64
    a = torch.randn(B * L, D, generator=g); b = torch.randn(B * S, D, generator=g)
    out = torch.ones(B * S, D, device="cuda")
    hip.rows_add(B * L, D, a_f32=a.cuda(), b_f32=b.cuda(), out_f32=out, alpha=0.5, accumulate=True,
                 b_map=(L, S, 1), o_map=(L, S, 1))
    ref = torch.ones(B, S, D); ref[:, 1:1 + L] += 0.5 * (a.view(B, L, D) + b.view(B, S, D)[:, 1:1 + L])
    assert rel(out, ref.view(B * S, D)) < 1e-6
    ob = torch.empty(B * L, D, dtype=torch.bfloat16, device="cuda")
    hip.rows_add(B * L, D, a_bf16=bf(a).cuda(), b_f32=a.cuda(), out_bf16=ob)
    assert rel(ob, bf(a).float() + a) < 3e-3
    ids = torch.randint(0, 50, (B, L), generator=g)
    word = torch.randn(50, D, generator=g); pos = torch.randn(16, D, generator=g); typ = torch.randn(2, D, generator=g)
    e = hip.bert_embed_fwd(ids.cuda(), word.cuda(), pos.cuda(), typ.cuda(), L)
    assert rel(e, (word[ids] + pos[:L][None] + typ[0]).view(B * L, D)) < 1e-6
    de = torch.randn(B * L, D, generator=g)
    dw = torch.zeros(50, D, device="cuda"); dp = torch.zeros(16, D, device="cuda"); dt = torch.zeros(2, D, device="cuda")
    hip.bert_embed_bwd(ids.cuda(), de.cuda(), dw, dp, dt, L)
    rw = torch.zeros(50, D).index_add_(0, ids.view(-1), de)
    assert rel(dw, rw) < 1e-5 and rel(dp.cpu()[:L], de.view(B, L, D).sum(0)) < 1e-5 and rel(dt.cpu()[0], de.sum(0)) < 1e-5


def test_context_mask_exact(hip):
    from oracle import reftr_oracle as O
    sm = torch.tensor([[1, 1, 1, 1, 0, 0], [1, 1, 1, 1, 1, 1], [1, 1, 0, 0, 0, 0]], dtype=torch.uint8)
    ctx, qm = hip.context_mask(sm.cuda())
    rc, rq = O.context_masks({"sentence_mask": sm})
    assert torch.equal(ctx.cpu().bool(), rc) and torch.equal(qm.cpu().bool(), rq)
    pm = torch.zeros(3, 2, 4, dtype=torch.uint8); pm[:, 0, :3] = 1; pm[0, 1, :4] = 1; pm[1, 1, :2] = 1
    pl = torch.tensor([[1, 2], [1, 0], [0, 0]]); pr = torch.tensor([[2, 4], [3, 1], [1, 1]])
    ctx, qm = hip.context_mask(sm.cuda(), pm.cuda(), pl.cuda(), pr.cuda())
    rc, rq = O.context_masks({"sentence_mask": sm, "phrase": pm, "phrase_mask": pm, "phrase_pos_l": pl, "phrase_pos_r": pr})
    assert torch.equal(ctx.cpu().bool(), rc) and torch.equal(qm.cpu().bool(), rq)


def test_qenc_attention(hip):
    g = torch.Generator().manual_seed(8)
    B, P, L, E = 3, 4, 11, 256
    k = torch.randn(B, E, generator=g, requires_grad=True)
    qs = (torch.randn(B, L, E, generator=g) * 0.2).requires_grad_(True); vs = torch.randn(B, L, E, generator=g, requires_grad=True)
    ctx = torch.rand(B, P, L, generator=g) < 0.4
    ctx[:, :, 0] = False
    w = (k[:, None, :] @ qs.transpose(1, 2)).expand(-1, P, -1).masked_fill(ctx, float("-inf")).softmax(-1)
    c = (vs.unsqueeze(1) * w.unsqueeze(-1)).sum(-2)
    dc = torch.randn(B, P, E, generator=g)
    c.backward(dc)
    wg, cg = hip.qenc_attn_fwd(cu(k.detach()), cu(qs.detach()), cu(vs.detach()), ctx.to(torch.uint8).cuda())
    assert rel(wg, w) < 1e-5 and rel(cg, c) < 1e-5
    dk, dqs, dvs = hip.qenc_attn_bwd(cu(k.detach()), cu(qs.detach()), cu(vs.detach()), wg, dc.cuda())
    assert rel(dk, k.grad) < 1e-4 and rel(dqs, qs.grad) < 1e-4 and rel(dvs, vs.grad) < 1e-4


# ------------------------------------------------------------------ loss + optimizer
def test_box_loss_against_golden_and_autograd(hip):
    import os
    from oracle import reftr_oracle as O
    gd = np.load(os.path.join(os.path.dirname(__file__), "golden", "criterion.npz"))
    pm = torch.from_numpy(gd["mask"])                                     # [3, 4]
    tg = [torch.from_numpy(gd[f"t{i}"]) for i in range(3)]
    boxes = torch.stack([torch.from_numpy(gd["aux"]), torch.from_numpy(gd["pred"])])    # [2,3,4,1,4]
    logits = torch.log(boxes / (1 - boxes)).requires_grad_(True)
    off = torch.tensor([0] + list(np.cumsum([len(t) for t in tg])), dtype=torch.int32)
    nb = torch.tensor([float(sum(len(t) for t in tg))])
    losses, total, dl = hip.box_loss(logits.detach().cuda(), pm.to(torch.uint8).cuda(), torch.cat(tg).cuda(), off.cuda(),
                                     nb.cuda(), 1.0, 2.0)
    for li, sfx in ((0, "_0"), (1, "")):
        assert abs(float(losses[li, 0]) - float(gd["loss_bbox" + sfx])) < 2e-6
        assert abs(float(losses[li, 1]) - float(gd["loss_giou" + sfx])) < 2e-6
    targets = [{"boxes": t, "labels": torch.zeros(len(t))} for t in tg]
    b = logits.sigmoid()
    ol = O.criterion({"pred_boxes": b[1], "phrase_mask": pm, "aux_outputs": [{"pred_boxes": b[0], "phrase_mask": pm}]}, targets)
    tot = ol["loss_bbox"] + ol["loss_bbox_0"] + 2.0 * (ol["loss_giou"] + ol["loss_giou_0"])
    tot.backward()
    assert abs(float(total) - float(tot)) < 1e-5
    assert rel(dl, logits.grad) < 1e-4


def test_sqnorm_adamw(hip):
    from oracle import reftr_oracle as O
    g = torch.Generator().manual_seed(11)
    n = 4096 * 3 + 8
    p = torch.randn(n, generator=g); gr = torch.randn(n, generator=g) * 0.01
    ranges = [(0, 4096, 1e-4, 1e-4), (4096, n, 1e-5, 1e-4)]
    P = {"a": p[:4096].clone(), "b": p[4096:].clone()}
    state = {}
    pc, gc = p.cuda(), gr.cuda()
    m = torch.zeros(n, device="cuda"); v = torch.zeros(n, device="cuda")
    sq = torch.zeros(1, device="cuda"); gn = torch.zeros(1, device="cuda")
    for step in (1, 2, 3):
        grads = {"a": gr[:4096] * step, "b": gr[4096:] * step}
        total, clipped = O.clip_grad_norm(grads, 0.1)
        O.adamw_step(P, clipped, state, step, {"a": 1e-4, "b": 1e-5})
        gs = gc * step
        hip.sqnorm(gs, sq)
        hip.adamw_flat(pc, gs, m, v, step=step, ranges=ranges, gnorm_sq=sq, gnorm_out=gn, max_norm=0.1)
        assert abs(float(gn) - float(total)) < 1e-5 * float(total)
    assert rel(pc, torch.cat([P["a"], P["b"]])) < 1e-6


def test_sgd_flat_matches_torch_sgd_with_clipping_and_param_groups(hip):
    """rt_sgd_flat = the reference's --sgd optimizer (torch.optim.SGD(momentum=0.9, weight_decay), main_vg.py:263-265) behind
    clip_grad_norm_(0.1): three steps on two parameter groups with different learning rates, against torch itself."""
    g = torch.Generator().manual_seed(12)
    n = 4096 * 2 + 8
    p = torch.randn(n, generator=g); gr = torch.randn(n, generator=g) * 0.01
    ranges = [(0, 4096, 1e-2, 1e-4), (4096, n, 1e-3, 1e-4)]
    a = torch.nn.Parameter(p[:4096].clone()); b = torch.nn.Parameter(p[4096:].clone())
    ref = torch.optim.SGD([{"params": [a], "lr": 1e-2}, {"params": [b], "lr": 1e-3}], lr=1e-2, momentum=0.9, weight_decay=1e-4)
    pc = p.cuda(); m = torch.zeros(n, device="cuda"); v = m[:4]
    sq = torch.zeros(1, device="cuda"); gn = torch.zeros(1, device="cuda")
    for step in (1, 2, 3):
        a.grad = (gr[:4096] * step).clone(); b.grad = (gr[4096:] * step).clone()
        total = torch.nn.utils.clip_grad_norm_([a, b], 0.1)
        ref.step()
        gs = gr.cuda() * step
        hip.sqnorm(gs, sq)
        hip.adamw_flat(pc, gs, m, v, step=step, ranges=ranges, gnorm_sq=sq, gnorm_out=gn, max_norm=0.1, beta1=0.9, sgd=True)
        assert abs(float(gn) - float(total)) < 1e-5 * float(total)
    assert rel(pc, torch.cat([a.detach(), b.detach()])) < 1e-6
    assert rel(m, torch.cat([ref.state[a]["momentum_buffer"], ref.state[b]["momentum_buffer"]])) < 1e-5


def test_sqnorm_adamw_bf16_gradients_spans_active_and_device_lr(hip):
    """The optimizer variants the engine uses: gradients read from a bf16 buffer (the data-parallel exchange format) give
    exactly what the fp32 path gives on the same (bf16-representable) values; the update issued as two spans equals one
    launch; a launch with active == 0 changes nothing; lr_dev overrides the descriptor's rates."""
    g = torch.Generator().manual_seed(5)
    n = 4096 * 5 + 16
    p0 = torch.randn(n, generator=g).cuda(); gr16 = (torch.randn(n, generator=g) * 0.01).cuda().bfloat16()
    gr = gr16.float()
    ranges = [(0, 8192, 1e-4, 1e-4), (8192, n, 1e-5, 1e-4)]

    def run(ranges=ranges, **kw):
        p = p0.clone(); m = torch.zeros(n, device="cuda"); v = torch.zeros(n, device="cuda")
        sq = torch.zeros(1, device="cuda"); gn = torch.zeros(1, device="cuda")
        src = kw.pop("src")
        hip.sqnorm(src, sq)
        spans = kw.pop("spans", [None])
        for step in (1, 2):
            for sp in spans:
                hip.adamw_flat(p, gr, m, v, step=step, ranges=ranges, gnorm_sq=sq, gnorm_out=gn, max_norm=0.05, span=sp, **kw)
        return p, m, v, float(gn)

    ref = run(src=gr)
    got = run(src=gr16, g16=gr16)
    assert abs(got[3] - ref[3]) < 1e-6 * ref[3]
    for a, b in zip(got[:3], ref[:3]):                 # the norm is summed in another order: the clip factor moves by an ulp
        assert rel(a, b) < 1e-6
    two = run(src=gr, spans=[(0, 12288), (12288, n)])
    for a, b in zip(two[:3], ref[:3]):                 # (the norm's atomics make two runs differ by an ulp of the clip factor)
        assert rel(a, b) < 1e-6
    off = torch.zeros(1, dtype=torch.int32, device="cuda")
    idle = run(src=gr, active=off)
    assert torch.equal(idle[0], p0) and float(idle[1].abs().max()) == 0.0
    lr_dev = torch.tensor([1e-4, 1e-5, 0, 0, 0, 0, 0, 0], dtype=torch.float32, device="cuda")
    wrong = [(0, 8192, 7.0, 1e-4), (8192, n, 9.0, 1e-4)]          # rates by value that the device words must override
    dev = run(ranges=wrong, src=gr, lr_dev=lr_dev)
    for a, b in zip(dev[:3], ref[:3]):
        assert rel(a, b) < 1e-6


def test_head_granular_epilogue_dropout_equals_one_key_attention_dropout(hip):
    """rt_conv_gemm's drop_shift = log2(head_dim): one keep/drop decision per head, at the attention kernel's hash index
    b * H + h -- the zero pattern and the kept values of `attn(v, v, v)` over a single key are reproduced exactly."""
    torch.manual_seed(0)
    B, Hh, dh = 8, 8, 32
    E = Hh * dh
    x = torch.randn(B, E, device="cuda").bfloat16(); w = (torch.randn(E, E, device="cuda") * 0.06).bfloat16()
    bias = torch.randn(E, device="cuda") * 0.1
    qmask = torch.zeros(B, 1, dtype=torch.uint8, device="cuda")
    p, seed = 0.3, 12345
    hip.set_seed_dev(None)
    v, _ = hip.linear(x, w, bias=bias)
    o_attn, _ = hip.attn_fwd(v, v, v, qmask, B=B, H=Hh, Sq=1, Sk=1, dh=dh, scale=dh ** -0.5, drop_p=p, drop_seed=seed)
    o_fold, _ = hip.linear(x, w, bias=bias, drop_p=p, drop_seed=seed, drop_shift=5)
    za, zf = (o_attn.float().view(B, Hh, dh) == 0).all(-1), (o_fold.float().view(B, Hh, dh) == 0).all(-1)
    assert torch.equal(za, zf) and 0 < int(za.sum()) < B * Hh                 # whole heads are dropped, the same ones
    kept = ~za[..., None].expand(B, Hh, dh).reshape(B, E)
    assert rel(o_fold.float()[kept], o_attn.float()[kept]) < 6e-3               # kept values: v / (1 - p), one bf16 rounding apart
    assert rel(o_attn.float()[kept], (v.float() / (1 - p))[kept]) < 6e-3
