"""Debug aid (GPU box): rt_qenc_bwd / rt_head_loss against the launched chains on one shape, printing every tensor's distance."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from test_model_gpu import build, rel
from test_qregion_gpu import _qenc_inputs
from reftr_amd import hip
from reftr_amd.models.criterion import _box_weights
from reftr_amd.models.net import RELU

model, crit, P, ocfg = build(small=True)
model.eval()
net, st = model.net, model.store
model.refresh_now()
E = 256
junk = torch.full((64 << 20,), float("nan"), device="cuda"); del junk          # poison the allocator's free blocks
B, Pn = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (5, 3)
nq, NL = 1, model.cfg.dec_layers
N = B * Pn
g = torch.Generator(device="cuda").manual_seed(11)
t3 = torch.randn(NL * N, E, device="cuda", generator=g)
valid = torch.rand(B, Pn, device="cuda", generator=g) < 0.7
valid[:, 0] = True
nt = [int(v) for v in valid.sum(1).tolist()]
tg = [{"boxes": torch.rand(n, 4, device="cuda", generator=g) * 0.4 + 0.3, "labels": torch.zeros(n, dtype=torch.long, device="cuda")} for n in nt]
prepared = crit.prepare(tg, torch.device("cuda"))
vt = "vl_transformer."
hs16 = torch.empty(NL * N, E, dtype=torch.bfloat16, device="cuda")
_, _, _, hm, hr = net.ln_fwd(t3, vt + "decoder.norm.", y_bf16=hs16, want_f32=False)
y1, _ = net.lin_fwd("bbox_embed.layers.0.", hs16, act=RELU)
y2, _ = net.lin_fwd("bbox_embed.layers.1.", y1, act=RELU)
_, logits = net.lin_fwd("bbox_embed.layers.2.", y2, out_bf16=False, out_f32=True)
torch.cuda.synchronize()
print("chain finite:", {k: bool(torch.isfinite(v.float()).all()) for k, v in dict(hs16=hs16, y1=y1, y2=y2, logits=logits).items()})
hs16f = torch.empty_like(hs16)
h = model._head_loss_fused((crit, prepared), t3, hs16f, valid, NL, B, Pn, nq, N)
torch.cuda.synchronize()
print("fused finite:", {k: bool(torch.isfinite(v.float()).all()) for k, v in h.items()})
print("logits rel", rel(h["logits"], logits), "rows with nan (fused)", torch.isnan(h["logits"]).any(1).nonzero().flatten().tolist(),
      "(chain)", torch.isnan(logits).any(1).nonzero().flatten().tolist())
