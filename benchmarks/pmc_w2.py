"""Eager launches of the layer3 / layer4 weight-gradient groups (second-generation kernels) for rocprofv3 --pmc."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reftr_amd import hip
from wgrad_group_bench import GROUPS, conv
for name in ("layer3", "layer4", "layer2"):
    keep = []
    for c in GROUPS[name]:
        conv(None, keep, *c)
    for _ in range(3):
        b = hip.WgradBatch(workspace_mb=1024)
        for x, dy, dw, sc, geom in keep:
            b.add_conv(dy, x, dw, geom, scale=sc)
        b.run()
    torch.cuda.synchronize()
