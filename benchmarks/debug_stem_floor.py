"""Drill-down: HIP vs q-oracle at the stem / max-pool / first bottleneck of layer1 (where does layer1 leave the order floor?)."""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from oracle import reftr_oracle as O
from oracle.synth import make_inputs
from test_parity_fullsize_gpu import build_full
from test_model_gpu import rel, to_cuda
from reftr_amd import hip as H

samples, targets = make_inputs("seg_full", B=2, H=320, W=320, L=40)
model, crit, P, ocfg = build_full()
s, tg = to_cuda(samples, targets)
with torch.no_grad():
    model(s)                                   # builds the operands
    body = model.body
    img = s["img"].tensors.float().contiguous()
    B, _, Hh, Ww = img.shape
    Ho, Wo, _, _ = H.stem_geometry(Hh, Ww)
    xp = H.img_pack(img)
    y = H.stem_conv(xp, body.W["stem"], body.bn[body.PFX + "bn1."][1], Ho, Wo)
    yp = H.maxpool3x3s2(y)
    # oracle, q-mode
    pfx = "img_backbone.0.body."
    x = samples["img"]
    scale, shift = O.frozen_bn_affine(P, pfx + "bn1.")
    w = P[pfx + "conv1.weight"]
    oy = O.rq(F.relu(O.conv2d_acc(O.rq(x, True), O.rq_fwd(w * scale.view(-1, 1, 1, 1), True), None, 2, 3) + shift.view(1, -1, 1, 1)), True)
    with O.accumulate_permuted(3):
        oy2 = O.rq(F.relu(O.conv2d_acc(O.rq(x, True), O.rq_fwd(w * scale.view(-1, 1, 1, 1), True), None, 2, 3) + shift.view(1, -1, 1, 1)), True)
    op = F.max_pool2d(oy, 3, 2, 1)

    def nchw(t):
        return t.float().cpu().permute(0, 3, 1, 2)
    print("input  bf16(img) exact:", bool(torch.equal(xp[..., :3].float().cpu().permute(0, 3, 1, 2)[:, :, 3:3 + Hh, 3:3 + Ww] if xp.dim() == 4 and xp.shape[1] != Hh else xp[..., :3].float().cpu().permute(0, 3, 1, 2), x.to(torch.bfloat16).float())) if False else "skipped")
    print("stem   HIP vs q %.3e   floor %.3e   shapes %s %s" % (rel(nchw(y), oy), rel(oy2, oy), tuple(y.shape), tuple(oy.shape)))
    d = (nchw(y) - oy).abs()
    print("       max abs diff %.3e, fraction of elements that differ %.3e, mean |oracle| %.3e" % (float(d.max()), float((d > 0).float().mean()), float(oy.abs().mean())))
    print("pool   HIP vs q %.3e" % rel(nchw(yp), op))
    # first bottleneck of layer1, conv by conv, each fed with the HIP tensor of the previous one on both sides
    blk = body.blocks[0][0]
    shp = (B, yp.shape[1], yp.shape[2])
    xin = yp.view(-1, 64)
    xo = nchw(yp)                              # the oracle consumes the HIP activations: isolates each convolution
    p0 = pfx + "layer1.0."
    idt, _, _ = body._conv(xin, shp, blk.down, relu=False)
    o_idt = O.conv_bn(xo, P, p0 + "downsample.0.", p0 + "downsample.1.", 1, 0, True, relu=False)
    h1, s1, _ = body._conv(xin, shp, blk.conv1, relu=True)
    o_h1 = O.conv_bn(xo, P, p0 + "conv1.", p0 + "bn1.", 1, 0, True)
    h2, s2, _ = body._conv(h1, s1, blk.conv2, relu=True)
    o_h2 = O.conv_bn(nchw(h1.view(B, s1[1], s1[2], -1)), P, p0 + "conv2.", p0 + "bn2.", 1, 1, True)
    out, s3, _ = body._conv(h2, s2, blk.conv3, relu=True, res=idt)
    o_out = O.conv_bn(nchw(h2.view(B, s2[1], s2[2], -1)), P, p0 + "conv3.", p0 + "bn3.", 1, 0, True, relu=True, residual=nchw(idt.view(B, s3[1], s3[2], -1)))
    for nm, a, b in (("downsample", idt, o_idt), ("conv1", h1, o_h1), ("conv2", h2, o_h2), ("conv3 + identity", out, o_out)):
        a = nchw(a.view(B, shp[1], shp[2], -1))
        d = (a - b).abs()
        print("%-18s HIP vs q (same inputs) %.3e   differing elements %.3e   max abs %.3e" % (nm, rel(a, b), float((d > 0).float().mean()), float(d.max())))
