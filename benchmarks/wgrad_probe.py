import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reftr_amd import hip
from tile_sweep import graph_time
from wgrad_sweep import CONVS, LINS
CONVS = CONVS + [(n, M, 1, K, N, 1, 1) for n, M, K, N in LINS]
MS = (0, 1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128, 256)
print("msplit: " + " ".join("%6d" % m for m in MS))
for name, B, Hh, ci, co, k, st in CONVS:
    pad = k // 2
    ho = (Hh + 2 * pad - k) // st + 1
    if Hh == 1: x = None
    x = torch.randn(B, Hh, Hh, ci, device="cuda").bfloat16(); dy = torch.randn(B, ho, ho, co, device="cuda").bfloat16()
    dw = torch.zeros(co, k, k, ci, device="cuda"); sc = torch.rand(co, device="cuda")
    geom = (B, Hh, Hh, ci, ho, ho, co, k, k, st, pad)
    row = [graph_time(lambda: hip.conv_wgrad(dy, x, dw, geom=geom, scale=sc, msplit=ms)) for ms in MS]
    print("%-22s " % name + " ".join("%6.1f" % t for t in row) + "   best %d" % MS[row.index(min(row))], flush=True)
