"""rt_stem_pool against rt_stem_conv + rt_maxpool3x3s2, the stem of configs[1] (8 x 640 x 640) and of configs[4] (8 x 800 x 800),
cold caches (FLUSH=1: a 640 MB streaming pass between launches, its own time subtracted) and warm; inside one hipGraph."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from reftr_amd import hip
import tile_sweep as TS

for flush in (True, False):
    TS.FLUSH = flush
    for B, S in ((8, 640), (8, 800)):
        img = torch.randn(B, 3, S, S, device="cuda")
        Ho, Wo, Hp, Wp = hip.stem_geometry(S, S)
        xp = hip.img_pack(img)
        wk = torch.empty(64, 7, 8, 4, dtype=torch.bfloat16, device="cuda")
        hip.stem_weight_prep(torch.randn(64, 7, 7, 3, device="cuda") / 12, torch.rand(64, device="cuda") + 0.5, wk)
        shift = torch.randn(64, device="cuda") * 0.3
        def two():
            hip.maxpool3x3s2(hip.stem_conv(xp, wk, shift, Ho, Wo))
        def conv_only():
            hip.stem_conv(xp, wk, shift, Ho, Wo)
        def one():
            hip.stem_pool(xp, wk, shift, Ho, Wo)
        t2, tc, t1 = TS.graph_time(two), TS.graph_time(conv_only), TS.graph_time(one)
        out_mb = B * Ho * Wo * 64 * 2 / 1e6
        print("%s %d x %d x %d  stem_conv + maxpool %6.1f us (conv alone %6.1f; %5.0f MB written + read back)   stem_pool %6.1f us (%5.0f MB written)" % (
            "cold" if flush else "warm", B, S, S, t2, tc, out_mb, t1, out_mb / 4), flush=True)
