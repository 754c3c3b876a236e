"""Weight gradients the way the step launches them: one rt_conv_wgrad_grouped call per ResNet stage (all of its 1x1 / 3x3
convolutions) and per transformer section, timed inside a hipGraph.  REFTR_WG2=0 gives the first-generation kernels."""
import os, sys, torch
OW = os.environ.get("OVERWRITE", "1") == "1"      # the training loop's mode: first writer assigns
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reftr_amd import hip
from tile_sweep import graph_time

def conv(batch, keep, B, H, ci, co, k, st):
    pad = k // 2
    ho = (H + 2 * pad - k) // st + 1
    x = torch.randn(B, H, H, ci, device="cuda").bfloat16(); dy = torch.randn(B, ho, ho, co, device="cuda").bfloat16()
    dw = torch.zeros(co, k, k, ci, device="cuda"); sc = torch.rand(co, device="cuda")
    keep.append((x, dy, dw, sc, (B, H, H, ci, ho, ho, co, k, k, st, pad)))
    return 2.0 * B * ho * ho * co * ci * k * k

def stage(planes, H, blocks, inpl):
    """(B, H_in, cin, cout, k, stride) of every trainable conv of a ResNet stage whose OUTPUT is H/2 (stride on block 0's 3x3)."""
    out = []
    for b in range(blocks):
        s = 2 if b == 0 else 1
        hin = H if b == 0 else H // 2
        out += [(8, hin, inpl if b == 0 else planes * 4, planes, 1, 1), (8, hin, planes, planes, 3, s), (8, H // 2, planes, planes * 4, 1, 1)]
        if b == 0:
            out.append((8, hin, inpl, planes * 4, 1, s))
    return out

GROUPS = {"layer2": stage(128, 160, 4, 256), "layer3": stage(256, 80, 6, 512), "layer4": stage(512, 40, 3, 1024)}
if os.environ.get("SYN"):          # balanced synthetic groups: the effect of XCD-local placement without the imbalance of a real stage
    GROUPS = {"16 x layer3 3x3": [(8, 40, 256, 256, 3, 1)] * 16, "16 x layer4 1x1 2048->512": [(8, 20, 2048, 512, 1, 1)] * 16,
              "16 x layer2 3x3": [(8, 80, 128, 128, 3, 1)] * 16, "8 x layer4 3x3": [(8, 20, 512, 512, 3, 1)] * 8}
LIN = {"bert x12": [(320, 768, 2304), (320, 768, 768), (320, 768, 3072), (320, 3072, 768)] * 12,
       "encoder x6": [(3520, 256, 512), (3520, 256, 256), (3520, 256, 256), (3520, 256, 2048), (3520, 2048, 256)] * 6,
       "decoder kv x6 + input_proj": [(3520, 256, 256)] * 12 + [(3200, 2048, 256)]}
tot_t = tot_f = 0.0
for name, convs in (GROUPS.items() if os.environ.get("ONLY") != "lin" else []):
    keep = []; fl = 0.0
    for c in convs:
        fl += conv(None, keep, *c)
    def run():
        b = hip.WgradBatch(workspace_mb=1024)
        for x, dy, dw, sc, geom in keep:
            b.add_conv(dy, x, dw, geom, scale=sc, overwrite=OW)
        b.run()
    t = graph_time(run, iters=5)
    tot_t += t; tot_f += fl
    print(f"{name:28s} {len(convs):3d} problems  {t:8.1f} us  {fl / t / 1e6:6.0f} TF/s", flush=True)
for name, lins in (LIN.items() if os.environ.get("ONLY") != "conv" else []):
    keep = []; fl = 0.0
    for M, K, N in lins:
        keep.append((torch.randn(M, K, device="cuda").bfloat16(), torch.randn(M, N, device="cuda").bfloat16(), torch.zeros(N, K, device="cuda"), torch.zeros(N, device="cuda")))
        fl += 2.0 * M * N * K
    def run():
        b = hip.WgradBatch(workspace_mb=512)
        for x, dy, dw, db in keep:
            b.add(dy, x, dw, db, overwrite=OW)
        b.run()
    t = graph_time(run, iters=5)
    tot_t += t; tot_f += fl
    print(f"{name:28s} {len(lins):3d} problems  {t:8.1f} us  {fl / t / 1e6:6.0f} TF/s", flush=True)
print(f"{'all weight gradients':28s}               {tot_t:8.1f} us  {tot_f / tot_t / 1e6:6.0f} TF/s")
