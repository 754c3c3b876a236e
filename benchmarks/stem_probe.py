"""us per launch of the stem convolution and the max-pool at the bench shape (8 x 640 x 640), 20 launches per hipGraph."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reftr_amd import hip
from tile_sweep import graph_time
x = torch.randn(8, 320, 320, 64, device="cuda").bfloat16(); y = torch.empty(8, 160, 160, 64, device="cuda", dtype=torch.bfloat16)
print("maxpool %.1f us" % graph_time(lambda: hip._check(hip.lib().rt_maxpool3x3s2(hip._p(x), hip._p(y), 8, 320, 320, 64, 160, 160, hip._stream()), "mp")))
