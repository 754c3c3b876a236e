"""A few launches of rt_bottleneck_fwd (every form, both variants) at 8 x 160 x 160 for rocprofv3 --pmc (stall breakdown)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reftr_amd import hip
B, Hh = 8, 160
for cin, down in ((64, True), (256, False)):
    x = torch.relu(torch.randn(B, Hh, Hh, cin, device="cuda")).bfloat16()
    w1 = (torch.randn(64, cin, device="cuda") / cin ** 0.5).bfloat16(); w2 = (torch.randn(64, 9, 64, device="cuda") / 24).bfloat16()
    w3 = (torch.randn(256, 64, device="cuda") / 8).bfloat16(); wd = (torch.randn(256, cin, device="cuda") / cin ** 0.5).bfloat16() if down else None
    b1 = torch.randn(64, device="cuda"); b2 = torch.randn(64, device="cuda"); b3 = torch.randn(256, device="cuda"); bd = torch.randn(256, device="cuda") if down else None
    out = torch.empty(B, Hh, Hh, 256, device="cuda", dtype=torch.bfloat16)
    for form in (1, 2, 3):
        for _ in range(4):
            hip.bottleneck_fwd(x, w1, b1, w2, b2, w3, b3, wd=wd, bd=bd, out=out, form=form)
torch.cuda.synchronize()
