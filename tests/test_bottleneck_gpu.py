"""GPU parity of rt_bottleneck_fwd (one frozen layer1 bottleneck in one launch) against
 (1) a plain torch fp32 reference of the same block with the kernel's bf16 rounding points (h1, h2, out), and
 (2) the three / four rt_conv_gemm launches it replaces (same operands, same rounding points).
Reference block: torchvision Bottleneck v1.5 with FrozenBatchNorm2d folded (models/modeling/backbone.py:43-80, 87-89)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def bf(x):
    return x.bfloat16()


def make_block(cin, down, seed):
    g = torch.Generator().manual_seed(seed)
    w1 = bf(torch.randn(64, cin, generator=g) / cin ** 0.5)
    w2 = bf(torch.randn(64, 3, 3, 64, generator=g) / 576 ** 0.5)          # [N][KH][KW][C]
    w3 = bf(torch.randn(256, 64, generator=g) / 8.0)
    wd = bf(torch.randn(256, cin, generator=g) / cin ** 0.5) if down else None
    b1, b2, b3 = (torch.randn(n, generator=g) * 0.3 for n in (64, 64, 256))
    bd = torch.randn(256, generator=g) * 0.3 if down else None
    return w1, b1, w2, b2, w3, b3, wd, bd


def torch_block(x, w1, b1, w2, b2, w3, b3, wd, bd, round_idt=False):
    """x bf16 [B,H,W,cin] -> fp32 [B,H,W,256] BEFORE the final bf16 rounding; h1 / h2 rounded to bf16 like the kernel's."""
    xc = x.float().permute(0, 3, 1, 2)
    h1 = bf(torch.relu(F.conv2d(xc, w1.float()[:, :, None, None]) + b1[None, :, None, None])).float()
    h2 = bf(torch.relu(F.conv2d(h1, w2.float().permute(0, 3, 1, 2), padding=1) + b2[None, :, None, None])).float()
    y = F.conv2d(h2, w3.float()[:, :, None, None]) + b3[None, :, None, None]
    if wd is not None:
        idt = F.conv2d(xc, wd.float()[:, :, None, None]) + bd[None, :, None, None]
        if round_idt:
            idt = bf(idt).float()
    else:
        idt = xc
    return torch.relu(y + idt).permute(0, 2, 3, 1).contiguous()


CASES = [
    # B, H, W, cin, down
    (2, 16, 32, 256, False),      # whole tiles
    (2, 16, 32, 64, True),
    (1, 21, 37, 256, False),      # ragged in both directions (partial tiles, halo clipped by the image on every side)
    (1, 21, 37, 64, True),
    (3, 8, 16, 256, False),       # a single tile per image: every halo pixel is padding
    (2, 40, 40, 256, False),
    (2, 40, 40, 64, True),
]


@pytest.mark.parametrize("form", [0, 1])
@pytest.mark.parametrize("B,H,W,cin,down", CASES)
def test_fused_bottleneck_vs_torch(hip, B, H, W, cin, down, form):
    g = torch.Generator().manual_seed(B * 1000 + H * 10 + cin)
    x = bf(torch.relu(torch.randn(B, H, W, cin, generator=g)))           # a block input is a ReLU output
    blk = make_block(cin, down, seed=cin + H)
    ref = torch_block(x, *blk, round_idt=True)
    dev = [t.cuda() if t is not None else None for t in blk]
    w1, b1, w2, b2, w3, b3, wd, bd = dev
    out = hip.bottleneck_fwd(x.cuda(), w1, b1, w2.view(64, 9, 64), b2, w3, b3, wd=wd, bd=bd, form=form)
    torch.cuda.synchronize()
    o = out.float().cpu()
    assert o.shape == ref.shape
    # h1 / h2 are re-rounded to bf16 from fp32 sums whose order differs from torch's: a flipped rounding of an intermediate moves an
    # output by ~1e-3 relative; the bulk must sit at the final bf16 rounding (2^-9 relative)
    err = (o - ref).abs()
    scale = ref.abs().mean()
    assert float((o - ref).norm() / ref.norm()) < 4e-3
    assert float(err.max()) < 0.06 * max(1.0, float(ref.abs().max()))
    assert float(err.mean() / scale) < 3e-3
    # padding semantics: conv2 pads h1 with zeros -- the border ring is where a wrong halo shows up first
    ring = torch.ones(H, W, dtype=torch.bool); ring[1:-1, 1:-1] = False
    assert float((o[:, ring] - ref[:, ring]).norm() / ref[:, ring].norm()) < 4e-3


@pytest.mark.parametrize("form", [0, 1])
@pytest.mark.parametrize("B,H,W,cin,down", [(2, 24, 40, 256, False), (2, 24, 40, 64, True), (8, 160, 160, 256, False), (8, 160, 160, 64, True),
                                            (5, 100, 100, 256, False), (5, 100, 100, 64, True)])
def test_fused_bottleneck_vs_the_launches_it_replaces(hip, B, H, W, cin, down, form):
    g = torch.Generator(device="cuda").manual_seed(7 + cin)
    x = torch.relu(torch.randn(B, H, W, cin, generator=g, device="cuda")).bfloat16()
    w1, b1, w2, b2, w3, b3, wd, bd = [t.cuda() if t is not None else None for t in make_block(cin, down, seed=3)]
    fused = hip.bottleneck_fwd(x, w1, b1, w2.view(64, 9, 64), b2, w3, b3, wd=wd, bd=bd, form=form)
    g1 = (B, H, W, cin, H, W, 64, 1, 1, 1, 0)
    g2 = (B, H, W, 64, H, W, 64, 3, 3, 1, 1)
    g3 = (B, H, W, 64, H, W, 256, 1, 1, 1, 0)
    gd = (B, H, W, cin, H, W, 256, 1, 1, 1, 0)
    xf = x.view(-1, cin)
    idt = xf
    if down:
        idt, _ = hip.conv_gemm(xf, wd, geom=gd, bias=bd, act=hip.ACT_NONE)
    h1, _ = hip.conv_gemm(xf, w1, geom=g1, bias=b1, act=hip.ACT_RELU)
    h2, _ = hip.conv_gemm(h1, w2.view(64, 9, 64), geom=g2, bias=b2, act=hip.ACT_RELU)
    y, _ = hip.conv_gemm(h2, w3, geom=g3, bias=b3, res_bf16=idt, res_first=True, act=hip.ACT_RELU)
    torch.cuda.synchronize()
    a, r = fused.float().view(-1, 256), y.float().view(-1, 256)
    # same operands, same rounding points (identity variant): the two differ by flipped bf16 roundings only
    assert float((a - r).norm() / r.norm()) < 2e-3
    frac_equal = float((a == r).float().mean())
    assert frac_equal > 0.97, frac_equal


def test_fused_bottleneck_rejects_what_it_does_not_implement(hip):
    x = torch.zeros(1, 8, 16, 128, device="cuda", dtype=torch.bfloat16)
    w1 = torch.zeros(64, 128, device="cuda", dtype=torch.bfloat16)
    w2 = torch.zeros(64, 9, 64, device="cuda", dtype=torch.bfloat16)
    w3 = torch.zeros(256, 64, device="cuda", dtype=torch.bfloat16)
    b = torch.zeros(256, device="cuda")
    with pytest.raises(RuntimeError):
        hip.bottleneck_fwd(x, w1, b, w2, b, w3, b)                       # cin = 128: neither variant
    x64 = torch.zeros(1, 8, 16, 64, device="cuda", dtype=torch.bfloat16)
    with pytest.raises(RuntimeError):
        hip.bottleneck_fwd(x64, torch.zeros(64, 64, device="cuda", dtype=torch.bfloat16), b, w2, b, w3, b)     # cin = 64 needs the downsample


@pytest.mark.parametrize("B,H,W,cin,down", [(3, 37, 53, 256, False), (3, 37, 53, 64, True)])
def test_two_launches_are_bit_identical_and_removed_forms_are_rejected(hip, B, H, W, cin, down):
    """A tile's result does not depend on which workgroup computed it: two launches agree bit for bit.  The persistent / 8-wave
    forms of round 4 (desc.form = 2, 3: correct, bit-identical, slower) were removed in round 5: asking for them is an error."""
    g = torch.Generator(device="cuda").manual_seed(11)
    x = torch.relu(torch.randn(B, H, W, cin, generator=g, device="cuda")).bfloat16()
    w1, b1, w2, b2, w3, b3, wd, bd = [t.cuda() if t is not None else None for t in make_block(cin, down, seed=5)]
    a = hip.bottleneck_fwd(x, w1, b1, w2.view(64, 9, 64), b2, w3, b3, wd=wd, bd=bd, form=1)
    b = hip.bottleneck_fwd(x, w1, b1, w2.view(64, 9, 64), b2, w3, b3, wd=wd, bd=bd, form=0)
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    for form in (2, 3):
        with pytest.raises(RuntimeError):
            hip.bottleneck_fwd(x, w1, b1, w2.view(64, 9, 64), b2, w3, b3, wd=wd, bd=bd, form=form)
