"""Stage-by-stage wall-clock of rt_qenc_fwd / rt_head_loss / rt_qenc_bwd (workgroup 0) inside the replayed configs[1] step."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from reftr_amd import hip

sys.argv = ["bench.py", "--steps", "10", "--warmup", "5", "--no-cpu-baseline", "--no-kernel-roofline"]
bench.main()
torch.cuda.synchronize()
tr = hip.qregion_trace()
for name, row in zip(("rt_qenc_fwd", "rt_head_loss", "rt_qenc_bwd"), tr):
    st = [v for v in row if v]
    if len(st) < 2:
        continue
    d = [(b - a) / 100.0 for a, b in zip(st, st[1:])]
    print(f"{name}: total {(st[-1] - st[0]) / 100.0:.1f} us, stages [us] " + " ".join(f"{x:.1f}" for x in d))
