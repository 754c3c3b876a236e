"""Encoder-shaped attention launches (B = 8, H = 8, S = 440, dh = 32) back to back inside one hipGraph: us per launch."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reftr_amd import hip
from tile_sweep import graph_time
B, H, S, dh = 8, 8, int(os.environ.get("S", "440")), 32
E = H * dh
q = torch.randn(B * S, E, device="cuda").bfloat16(); k = torch.randn(B * S, E, device="cuda").bfloat16(); v = torch.randn(B * S, E, device="cuda").bfloat16()
kpm = torch.zeros(B, S, dtype=torch.uint8, device="cuda"); kpm[:, S - 30:] = 1
o, lse = hip.attn_fwd(q, k, v, kpm, B=B, H=H, Sq=S, Sk=S, dh=dh, scale=dh ** -0.5)
do = torch.randn_like(o)
print("fwd  %.1f us" % graph_time(lambda: hip.attn_fwd(q, k, v, kpm, B=B, H=H, Sq=S, Sk=S, dh=dh, scale=dh ** -0.5)))
print("bwd  %.1f us (dq + dkv)" % graph_time(lambda: hip.attn_bwd(q, k, v, o, do, lse, kpm, B=B, H=H, Sq=S, Sk=S, dh=dh, scale=dh ** -0.5)))
