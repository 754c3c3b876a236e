"""rt_bottleneck_fwd against the rt_conv_gemm launches it replaces, layer1 shapes of configs[1] (8 x 160 x 160), cold caches
(FLUSH=1: a 640 MB streaming pass between launches, its own time subtracted) and warm; 20 repetitions inside one hipGraph."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from reftr_amd import hip
import tile_sweep as TS

B, Hh = int(os.environ.get("B", 8)), int(os.environ.get("HW", 160))
for flush in (True, False):
    TS.FLUSH = flush
    for cin, down in ((64, True), (256, False)):
        x = torch.relu(torch.randn(B, Hh, Hh, cin, device="cuda")).bfloat16()
        w1 = (torch.randn(64, cin, device="cuda") / cin ** 0.5).bfloat16(); w2 = (torch.randn(64, 9, 64, device="cuda") / 24).bfloat16()
        w3 = (torch.randn(256, 64, device="cuda") / 8).bfloat16(); wd = (torch.randn(256, cin, device="cuda") / cin ** 0.5).bfloat16() if down else None
        b1 = torch.randn(64, device="cuda"); b2 = torch.randn(64, device="cuda"); b3 = torch.randn(256, device="cuda"); bd = torch.randn(256, device="cuda") if down else None
        out = torch.empty(B, Hh, Hh, 256, device="cuda", dtype=torch.bfloat16)
        g1 = (B, Hh, Hh, cin, Hh, Hh, 64, 1, 1, 1, 0); g2 = (B, Hh, Hh, 64, Hh, Hh, 64, 3, 3, 1, 1)
        g3 = (B, Hh, Hh, 64, Hh, Hh, 256, 1, 1, 1, 0); gd = (B, Hh, Hh, cin, Hh, Hh, 256, 1, 1, 1, 0)
        xf = x.view(-1, cin)
        def chain():
            idt = xf
            if down:
                idt, _ = hip.conv_gemm(xf, wd, geom=gd, bias=bd, act=hip.ACT_NONE)
            h1, _ = hip.conv_gemm(xf, w1, geom=g1, bias=b1, act=hip.ACT_RELU)
            h2, _ = hip.conv_gemm(h1, w2, geom=g2, bias=b2, act=hip.ACT_RELU)
            hip.conv_gemm(h2, w3, geom=g3, bias=b3, res_bf16=idt, res_first=True, act=hip.ACT_RELU)
        def fused1():
            hip.bottleneck_fwd(x, w1, b1, w2, b2, w3, b3, wd=wd, bd=bd, out=out, form=1)
        def fused():
            hip.bottleneck_fwd(x, w1, b1, w2, b2, w3, b3, wd=wd, bd=bd, out=out, form=2)
        def fused3():
            hip.bottleneck_fwd(x, w1, b1, w2, b2, w3, b3, wd=wd, bd=bd, out=out, form=3)
        tc, t1, tf, t3 = TS.graph_time(chain), TS.graph_time(fused1), TS.graph_time(fused), TS.graph_time(fused3)
        M = B * Hh * Hh
        fl = 2.0 * M * (cin * 64 + 576 * 64 + 64 * 256 + (cin * 256 if down else 0))
        io = M * (cin + 256) * 2
        print("%s cin %3d %s  launches %6.1f us   form 1 (8x16, 2 WG/CU) %6.1f us   form 3 (8x32, 8 waves) %6.1f us   form 2 (persistent) %6.1f us (%4.0f TF useful, %4.2f TB/s of block input + output: form 2)" % (
            "cold" if flush else "warm", cin, "downsample" if down else "identity  ", tc, t1, t3, tf, fl / tf / 1e6, io / tf / 1e6), flush=True)
