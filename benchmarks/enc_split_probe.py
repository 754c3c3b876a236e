"""Premise check (round 6): the transformer region runs on ONE stream (the language stream is idle from the forward join to the BERT
backward), its launches have 28-450 tiles for 256 CUs.  Does the encoder's 6-layer forward chain go faster as TWO half-batch chains on two
streams (row slices of the same tensors; nothing in the encoder couples images)?  Same functions the model calls (Net.enc_layer_fwd),
captured into hipGraphs, replayed."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from reftr_amd import hip
from reftr_amd.models import layout as Lm
from reftr_amd.models.reftr_transformer import RefTR

dev = torch.device("cuda")
model = RefTR(Lm.ModelConfig(), device=dev)
model.train()
model.refresh_operands() if hasattr(model, "refresh_operands") else None
net, cfg = model.net, model.cfg
B, S, E = 8, 440, cfg.hidden
vt = "vl_transformer."
g = torch.Generator(device="cuda").manual_seed(1)
x32 = torch.randn(B * S, E, device=dev, generator=g)
pos = torch.randn(B * S, E, device=dev, generator=g)
x16 = x32.bfloat16(); xp16 = (x32 + pos).bfloat16()
kpm = torch.zeros(B, S, dtype=torch.uint8, device=dev)

def chain(b0, b1):
    sl = slice(b0 * S, b1 * S)
    a32, a16, ap16 = x32[sl], x16[sl], xp16[sl]
    for i in range(cfg.enc_layers):
        a32, a16, ap16, r, _ = net.enc_layer_fwd(f"{vt}encoder.layers.{i}.", a32, a16, ap16, pos[sl], kpm[b0:b1], b1 - b0, S)
    return a32

def whole():
    return chain(0, B)

side = torch.cuda.Stream()
def split():
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        o2 = chain(B // 2, B)
    o1 = chain(0, B // 2)
    main.wait_stream(side)
    return o1, o2

def timed(fn, reps=50):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn(); fn()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    gph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gph):
        keep = fn()
    for _ in range(5): gph.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): gph.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6

for r in range(3):
    print("encoder forward x6, B = 8, S = 440:  one chain %.1f us   two half-batch chains on two streams %.1f us" % (timed(whole), timed(split)))
