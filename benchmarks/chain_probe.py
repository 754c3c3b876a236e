"""Cost of one link of a dependent chain of tiny kernels under hipGraph replay: trivial kernel, skinny GEMM with hot
weights, skinny GEMM with cold (never-reused) weights, LayerNorm on 8 rows."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reftr_amd import hip
from tile_sweep import graph_time
dev = "cuda"
x = torch.randn(8, 256, device=dev).bfloat16(); xf = torch.randn(8, 256, device=dev)
W = [(torch.randn(256, 256, device=dev) / 16).bfloat16() for _ in range(64)]
Wbig = [(torch.randn(2048, 256, device=dev) / 16).bfloat16() for _ in range(64)]
flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
gam = torch.ones(256, device=dev); bet = torch.zeros(256, device=dev)
def chain_trivial():
    t = xf
    for _ in range(64):
        o = torch.empty_like(t); hip.rows_add(8, 256, a_f32=t, out_f32=o); t = o
def chain_hot():
    t = x
    for _ in range(64):
        t, _ = hip.linear(t, W[0])
bias = torch.randn(256, device=dev); res = torch.randn(8, 256, device=dev)
def chain_hot_epi():          # decoder-style link: + bias, + fp32 residual, fp32 and bf16 outputs
    t = x
    for _ in range(64):
        t, _ = hip.linear(t, W[0], bias=bias, res_f32=res, out_bf16=True, out_f32=True)
def chain_cold():
    flush.fill_(1)            # evict L2 / infinity cache
    t = x
    for i in range(64):
        t, _ = hip.linear(t, W[i])
def flush_only():
    flush.fill_(1)
def chain_ln():
    t = xf
    for _ in range(64):
        t = hip.layernorm_fwd(t, gam, bet, 1e-5, want_bf16=False)[0]
def chain_ffn_cold():
    flush.fill_(1)
    t = x
    for i in range(32):
        h, _ = hip.linear(t, Wbig[i]); t, _ = hip.linear(h, Wbig[i + 32].t().contiguous() if False else Wbig[i].view(256, 2048))
for name, fn, n in (("trivial rows_add", chain_trivial, 64), ("skinny 256x256 hot weights", chain_hot, 64),
                    ("skinny 256x256 hot + bias + res", chain_hot_epi, 64),
                    ("layernorm 8 rows", chain_ln, 64)):
    print("%-32s %.2f us per link" % (name, graph_time(fn, iters=5) / n))
tf = graph_time(flush_only, iters=5)
print("%-32s %.2f us per link" % ("skinny 256x256 cold weights", (graph_time(chain_cold, iters=5) - tf) / 64))
print("%-32s %.2f us per link" % ("skinny 256<->2048 cold weights", (graph_time(chain_ffn_cold, iters=5) - tf) / 64))
