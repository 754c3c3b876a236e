import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import reftr_oracle as O
from oracle.shapes import param_shapes
from oracle.synth import make_inputs
from oracle.weights import formula_state
from reftr_amd.models import layout as L
from reftr_amd.models.reftr_transformer import RefTR
from reftr_amd import hip as H

def rel(a, b):
    a = a.detach().float().cpu(); b = b.detach().float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))

ocfg = O.Cfg(enc_layers=2, dec_layers=2, bert=O.BertCfg(layers=2))
cfg = L.ModelConfig(enc_layers=2, dec_layers=2, bert=L.BertConfig(layers=2))
samples, targets = make_inputs("e2e_single", B=2, H=96, W=128, L=12)
P = formula_state(param_shapes(ocfg))
model = RefTR(cfg, device="cuda"); model.load_state_dict(P); model.eval(); model.refresh_operands()
x = samples["img"]
pfx = "img_backbone.0.body."
with torch.no_grad():
    sc, sh = O.frozen_bn_affine(P, pfx + "bn1.")
    y0 = F.relu(F.conv2d(x, P[pfx + "conv1.weight"], None, 2, 3) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    y1 = F.max_pool2d(y0, 3, 2, 1)
    feats = O.resnet_body(x, P, q=True)
    wq = (P[pfx + "conv1.weight"] * sc.view(-1, 1, 1, 1)).bfloat16().float()
    y0q = F.relu(F.conv2d(x, wq, None, 2, 3) + sh.view(1, -1, 1, 1)).bfloat16().float()
    y1q = F.max_pool2d(y0q, 3, 2, 1)
body = model.body
xc = x.cuda()
Ho, Wo, _, _ = H.stem_geometry(x.shape[2], x.shape[3])
xp = H.img_pack(xc)
print("pack", rel(xp[:, 3:3 + 96, 3:3 + 128, :3].permute(0, 3, 1, 2), x), float(xp[..., 3].abs().max()))
y = H.stem_conv(xp, body.W["stem"], body.bn[pfx + "bn1."][1], Ho, Wo)
print("stem", rel(y.permute(0, 3, 1, 2), y0), y.shape, y0.shape)
print("stem vs q", rel(y.permute(0, 3, 1, 2), y0q), float((y.permute(0, 3, 1, 2).float().cpu() != y0q).float().mean()))
wst = body.W["stem"].float().cpu()      # [64][7][8][4]
print("stem w", rel(wst[:, :, :7, :3], wq.permute(0, 2, 3, 1)), float(wst[:, :, 7].abs().max()), float(wst[..., 3].abs().max()))
mp = H.maxpool3x3s2(y)
print("pool", rel(mp.permute(0, 3, 1, 2), y1), " vs q", rel(mp.permute(0, 3, 1, 2), y1q))
fs, saved = body.forward(xc)
for (f, shp), ref in zip(fs, feats):
    Bn, h, w = shp
    print("stage", shp, rel(f.view(Bn, h, w, -1).permute(0, 3, 1, 2), ref))
# first block pieces
b = body.blocks[0][0]
xin = mp.view(-1, 64)
h1, s1, _ = body._conv(xin, (2, mp.shape[1], mp.shape[2]), b.conv1, relu=True)
ref_h1 = O.conv_bn(y1q, P, pfx + "layer1.0.conv1.", pfx + "layer1.0.bn1.", 1, 0, True)
print("l1.0.conv1", rel(h1.view(2, s1[1], s1[2], -1).permute(0, 3, 1, 2), ref_h1))
h2, s2, _ = body._conv(h1, s1, b.conv2, relu=True)
ref_h2 = O.conv_bn(ref_h1, P, pfx + "layer1.0.conv2.", pfx + "layer1.0.bn2.", 1, 1, True)
print("l1.0.conv2", rel(h2.view(2, s2[1], s2[2], -1).permute(0, 3, 1, 2), ref_h2))
w = body.W[b.conv1.name]
wr = (P[b.conv1.name] * O.frozen_bn_affine(P, b.conv1.bn)[0].view(-1, 1, 1, 1)).permute(0, 2, 3, 1).reshape(w.shape)
print("w conv1", rel(w, wr), " bias", rel(body.bn[b.conv1.bn][1], O.frozen_bn_affine(P, b.conv1.bn)[1]))
with torch.no_grad():
    idt_ref = O.conv_bn(y1q, P, pfx + "layer1.0.downsample.0.", pfx + "layer1.0.downsample.1.", 1, 0, True, relu=False)
    out_ref = O.conv_bn(ref_h2, P, pfx + "layer1.0.conv3.", pfx + "layer1.0.bn3.", 1, 0, True, relu=True, residual=idt_ref)
    blk_ref = O.bottleneck(y1q, P, pfx + "layer1.0.", 1, True)
idt, _, _ = body._conv(xin, (2, mp.shape[1], mp.shape[2]), b.down, relu=False)
print("l1.0.down", rel(idt.view(2, s2[1], s2[2], -1).permute(0, 3, 1, 2), idt_ref))
out, s3, _ = body._conv(h2, s2, b.conv3, relu=True, res=idt)
print("l1.0.conv3+res", rel(out.view(2, s3[1], s3[2], -1).permute(0, 3, 1, 2), out_ref), rel(out_ref, blk_ref))
x2 = out
for bi in (1, 2):
    bb_ = body.blocks[0][bi]
    with torch.no_grad():
        blk_ref = O.bottleneck(blk_ref, P, pfx + f"layer1.{bi}.", 1, True)
    h1, s1, _ = body._conv(x2, s3, bb_.conv1, relu=True)
    h2, s2, _ = body._conv(h1, s1, bb_.conv2, relu=True)
    x2, s3, _ = body._conv(h2, s2, bb_.conv3, relu=True, res=x2)
    print(f"l1.{bi}", rel(x2.view(2, s3[1], s3[2], -1).permute(0, 3, 1, 2), blk_ref))
print("feats0 vs chain", rel(feats[0], blk_ref))
