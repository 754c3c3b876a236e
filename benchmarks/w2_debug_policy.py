import os, sys, torch
sys.path.insert(0, "benchmarks"); sys.path.insert(0, ".")
os.environ["ONLY"]="none"
from reftr_amd import hip
from wgrad_group_bench import GROUPS, conv
for name in ("layer3","layer4"):
    keep=[]
    for c in GROUPS[name]: conv(None, keep, *c)
    b = hip.WgradBatch(workspace_mb=1024)
    for x, dy, dw, sc, geom in keep: b.add_conv(dy, x, dw, geom, scale=sc, overwrite=True)
    print("==", name, file=sys.stderr); b.run(); torch.cuda.synchronize()
