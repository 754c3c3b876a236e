"""rt_conv_wgrad variants (0 = register-staged kernel, 1-5 = LDS-DMA variants) on the step's weight-gradient shapes,
20 back-to-back launches inside one hipGraph each."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reftr_amd import hip
from tile_sweep import graph_time
VARS = [int(v) for v in os.environ.get('VARS', '0,8,2').split(',')]      # 0 = library default (v2 kernels), 8 = first-generation default (128x128, 32-row chunks, 3 stages), 9 = register-staged
WS = os.environ.get('WS', '1') == '1'
# (name, B, H, Cin, Cout, k, stride)
CONVS = [("l1 1x1 64->256 @160", 8, 160, 64, 256, 1, 1), ("l1 1x1 256->64 @160", 8, 160, 256, 64, 1, 1), ("l1 3x3 64 @160", 8, 160, 64, 64, 3, 1),
         ("l2 1x1 128->512 @80", 8, 80, 128, 512, 1, 1), ("l2 1x1 512->128 @80", 8, 80, 512, 128, 1, 1), ("l2 3x3 128 @80", 8, 80, 128, 128, 3, 1),
         ("l2 3x3s2 128 @160", 8, 160, 128, 128, 3, 2),
         ("l3 1x1 256->1024 @40", 8, 40, 256, 1024, 1, 1), ("l3 1x1 1024->256 @40", 8, 40, 1024, 256, 1, 1), ("l3 3x3 256 @40", 8, 40, 256, 256, 3, 1),
         ("l3 3x3s2 256 @80", 8, 80, 256, 256, 3, 2),
         ("l4 1x1 512->2048 @20", 8, 20, 512, 2048, 1, 1), ("l4 1x1 2048->512 @20", 8, 20, 2048, 512, 1, 1), ("l4 3x3 512 @20", 8, 20, 512, 512, 3, 1),
         ("l4 3x3s2 512 @40", 8, 40, 512, 512, 3, 2)]
LINS = [("enc 256->256", 3520, 256, 256), ("enc 256->512", 3520, 256, 512), ("enc 256->2048", 3520, 256, 2048), ("enc 2048->256", 3520, 2048, 256),
        ("bert 768->768", 320, 768, 768), ("bert 768->2304", 320, 768, 2304), ("bert 768->3072", 320, 768, 3072), ("bert 3072->768", 320, 3072, 768)]
if __name__ == "__main__": print("variants: " + " ".join("%6d" % v for v in VARS))
for name, B, Hh, ci, co, k, st in (CONVS if __name__ == '__main__' else []):
    pad = k // 2
    ho = (Hh + 2 * pad - k) // st + 1
    x = torch.randn(B, Hh, Hh, ci, device="cuda").bfloat16(); dy = torch.randn(B, ho, ho, co, device="cuda").bfloat16()
    dw = torch.zeros(co, k, k, ci, device="cuda"); sc = torch.rand(co, device="cuda")
    geom = (B, Hh, Hh, ci, ho, ho, co, k, k, st, pad)
    row = [graph_time(lambda: hip.conv_wgrad(dy, x, dw, geom=geom, scale=sc, variant=v, workspace=WS)) for v in VARS]
    fl = 2.0 * B * ho * ho * co * ci * k * k
    print("%-22s " % name + " ".join("%6.1f" % v for v in row) + "  best %d (%4.0f TF)" % (VARS[row.index(min(row))], fl / min(row) / 1e6), flush=True)
for name, M, K, N in (LINS if __name__ == '__main__' else []):
    x = torch.randn(M, K, device="cuda").bfloat16(); dy = torch.randn(M, N, device="cuda").bfloat16()
    dw = torch.zeros(N, K, device="cuda"); db = torch.zeros(N, device="cuda")
    row = [graph_time(lambda: hip.linear_wgrad(dy, x, dw, dbias=db, variant=v, workspace=WS)) for v in VARS]
    fl = 2.0 * M * N * K
    print("%-22s " % name + " ".join("%6.1f" % v for v in row) + "  best %d (%4.0f TF)" % (VARS[row.index(min(row))], fl / min(row) / 1e6), flush=True)
