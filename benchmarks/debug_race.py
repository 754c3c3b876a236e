"""Widen stream-race windows on the small fixture: delay the side stream (or the main stream) with a spin kernel at the
fork of every SideStream.run / before every join and see whether the eager trajectory moves."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_model_gpu import build, to_cuda, rel, make_inputs
from reftr_amd import hip as H
from reftr_amd.engine_vg import train_step
from reftr_amd.optim import FusedAdamW
samples, targets = make_inputs("e2e_single", B=2, H=96, W=128, L=12)
s, tg = to_cuda(samples, targets)
MODE = ["none"]
_run, _join = H.SideStream.run, H.SideStream.join
def run(self, fn, *keep):
    if MODE[0] == "delay_side" and self.enabled:
        def g():
            torch.cuda._sleep(20_000_000)
            return fn()
        return _run(self, g, *keep)
    if MODE[0] == "delay_main_after_fork" and self.enabled:
        r = _run(self, fn, *keep)
        torch.cuda._sleep(20_000_000)
        return r
    return _run(self, fn, *keep)
H.SideStream.run = run
for mode in ("none", "delay_side", "delay_main_after_fork", "none", "delay_side"):
    MODE[0] = mode
    model, crit, P, ocfg = build(small=True)
    model.eval()
    opt = FusedAdamW(model, lr=1e-4, lr_backbone=1e-5, weight_decay=1e-4)
    out = []
    for it in range(3):
        lv, _, _, gn = train_step(model, crit, s, tg, opt, None, max_norm=0.1)
        torch.cuda.synchronize()
        out.append((lv, float(gn)))
    print("%-24s" % mode, ["%.7f" % o[0] for o in out], ["%.5f" % o[1] for o in out], flush=True)
