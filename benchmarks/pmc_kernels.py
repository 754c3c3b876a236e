"""A few launches of the step's representative conv GEMM / wgrad shapes, for rocprofv3 --pmc (stall breakdown)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reftr_amd import hip
B = 8
def conv(Hh, ci, co, k, st=1):
    pad = k // 2; ho = (Hh + 2 * pad - k) // st + 1
    x = torch.randn(B, Hh, Hh, ci, device="cuda").bfloat16(); w = (torch.randn(co, k, k, ci, device="cuda") / (k * k * ci) ** 0.5).bfloat16()
    wt = (torch.randn(ci, k, k, co, device="cuda") / (k * k * ci) ** 0.5).bfloat16(); dy = torch.randn(B, ho, ho, co, device="cuda").bfloat16()
    dw = torch.zeros(co, k, k, ci, device="cuda")
    geom = (B, Hh, Hh, ci, ho, ho, co, k, k, st, pad); geom_t = (B, ho, ho, co, Hh, Hh, ci, k, k, st, pad)
    for _ in range(5):
        hip.conv_gemm(x, w, geom=geom, act=hip.ACT_RELU)
        hip.conv_gemm(dy, wt, geom=geom_t, transposed=True)
        hip.conv_wgrad(dy, x, dw, geom=geom)
conv(40, 256, 256, 3); conv(40, 256, 1024, 1); conv(40, 1024, 256, 1); conv(160, 64, 256, 1); conv(80, 128, 128, 3); conv(20, 512, 512, 3)
x = torch.randn(3520, 256, device="cuda").bfloat16(); w = torch.randn(256, 256, device="cuda").bfloat16(); dy = torch.randn(3520, 256, device="cuda").bfloat16()
dw = torch.zeros(256, 256, device="cuda")
for _ in range(5):
    hip.linear(x, w); hip.linear_wgrad(dy, x, dw)
torch.cuda.synchronize()
