"""Tile / staging sweep of rt_conv_gemm on the step's shapes: register-staged tiles (hints 1-3) against the LDS-DMA
variants (hints 11-33), timed as 20 back-to-back launches inside one hipGraph."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reftr_amd import hip
HINTS = [int(h) for h in os.environ["HINTS"].split(",")] if os.environ.get("HINTS") else [1, 2, 3, 11, 12, 13, 21, 22, 31, 32, 33]
SHAPES = [("l1 64->256 @160", 204800, 64, 256), ("l1 256->64 @160", 204800, 256, 64), ("l1 64->64", 204800, 64, 64),
          ("l2 256->128 @160", 204800, 256, 128), ("l2 128->512 @80", 51200, 128, 512), ("l2 512->128 @80", 51200, 512, 128),
          ("l3 256->1024 @40", 12800, 256, 1024), ("l3 1024->256 @40", 12800, 1024, 256), ("l3 512->256 @80", 51200, 512, 256),
          ("l4 512->2048 @20", 3200, 512, 2048), ("l4 2048->512 @20", 3200, 2048, 512), ("l4 1024->512 @40", 12800, 1024, 512),
          ("enc 256->256", 3520, 256, 256), ("enc 256->2048", 3520, 256, 2048), ("enc 2048->256", 3520, 2048, 256),
          ("bert 768->768", 320, 768, 768), ("bert 768->3072", 320, 768, 3072), ("bert 3072->768", 320, 3072, 768),
          ("bert 768->2304", 320, 768, 2304), ("bert 2304->768", 320, 2304, 768),
          ("big 4096^3", 4096, 4096, 4096)]
# FLUSH=1: every timed launch runs on COLD caches, as in the training step (weights untouched since the last step, activations
# at best in the MALL): a 640 MB streaming pass between launches evicts L2 + MALL; its own time (a graph of flushes only) is
# subtracted.  Warm back-to-back launches (the default) flatter deep software pipelines: round 3 measured the pipelined 128 x 128
# tile 17 % faster warm and 12 % slower in the step.
FLUSH = os.environ.get("FLUSH", "0") == "1"
_fl = None
def _flush():
    global _fl
    if _fl is None:
        _fl = (torch.empty(160 << 20, dtype=torch.float32, device="cuda"), torch.zeros(1, device="cuda"))
    _fl[0].add_(1.0)
def graph_time(fn, iters=20):
    def body():
        if FLUSH: _flush()
        fn()
    body(); torch.cuda.synchronize()
    def timed(b):
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            b()
        torch.cuda.current_stream().wait_stream(s)
        with torch.cuda.graph(g):
            for _ in range(iters): b()
        g.replay(); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e3
    t = timed(body)
    if FLUSH:
        global _flush_us
        if "_flush_us" not in globals():
            _flush_us = timed(_flush)
        t -= _flush_us
    return t
if __name__ != "__main__":
    CONVS = SHAPES = []
elif os.environ.get("ONLY") == "lin":
    CONVS = []
else:
  CONVS = [("l1 3x3 64 @160", 8, 160, 64, 64, 1), ("l2 3x3 128 @80", 8, 80, 128, 128, 1), ("l2 3x3s2 128 @160", 8, 160, 128, 128, 2),
           ("l3 3x3 256 @40", 8, 40, 256, 256, 1), ("l4 3x3 512 @20", 8, 20, 512, 512, 1), ("l4 3x3s2 512 @40", 8, 40, 512, 512, 2)]
if __name__ == "__main__": print("hints: " + " ".join("%6d" % h for h in HINTS))
for name, B, Hh, ci, co, st in CONVS:
    ho = (Hh + 2 - 3) // st + 1
    x = torch.randn(B, Hh, Hh, ci, device="cuda").bfloat16(); w = (torch.randn(co, 3, 3, ci, device="cuda") / (9 * ci) ** 0.5).bfloat16()
    wt = (torch.randn(ci, 3, 3, co, device="cuda") / (9 * ci) ** 0.5).bfloat16(); dy = torch.randn(B, ho, ho, co, device="cuda").bfloat16()
    geom = (B, Hh, Hh, ci, ho, ho, co, 3, 3, st, 1); geom_t = (B, ho, ho, co, Hh, Hh, ci, 3, 3, st, 1)
    rf, rd = [], []
    for hint in HINTS:
        rf.append(graph_time(lambda: hip.conv_gemm(x, w, geom=geom, act=hip.ACT_RELU, tile_hint=hint)))
        rd.append(graph_time(lambda: hip.conv_gemm(dy, wt, geom=geom_t, transposed=True, tile_hint=hint)))
    fl = 2.0 * B * ho * ho * co * ci * 9
    print("%-20s fwd   " % name + " ".join("%6.1f" % v for v in rf) + "  best %d (%4.0f TF)" % (HINTS[rf.index(min(rf))], fl / min(rf) / 1e6), flush=True)
    print("%-20s dgrad " % name + " ".join("%6.1f" % v for v in rd) + "  best %d (%4.0f TF)" % (HINTS[rd.index(min(rd))], fl / min(rd) / 1e6), flush=True)
for name, M, K, N in (SHAPES if os.environ.get("ONLY") != "conv" else []):
    x = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
    res = torch.randn(M, N, device="cuda").bfloat16()
    ob = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    row = []
    for hint in HINTS:
        try:
            row.append(graph_time(lambda: hip.linear(x, w, act=hip.ACT_RELU, res_bf16=res, res_first=True, out_bf16=ob, tile_hint=hint)))
        except RuntimeError:                     # a form that does not take this shape (e.g. 501: K = 128 / 256 only)
            row.append(float("nan"))
    gb = (M * K + N * K + 2 * M * N) * 2 / 1e9
    fl = 2.0 * M * N * K
    best = min(v for v in row if v == v)
    print("%-20s       " % name + " ".join("%6.1f" % v for v in row) + "  best %d (%4.0f TF, %4.2f TB/s)" % (
        HINTS[row.index(best)], fl / best / 1e6, gb / best * 1e3), flush=True)
