"""Per-kernel microbenchmarks on the GPU box: ResNet-50 @640x640 B=8 conv GEMM shapes
and the transformer GEMM shapes, forward / backward-data / weight-gradient, in TFLOP/s.

    python benchmarks/kernel_bench.py [--out gpurun_out/kernel_bench.json]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reftr_amd import hip  # noqa: E402

# (name, Cin, Cout, k, stride, Hin) at 640x640 input, per image
R50 = [
    ("l1.conv1x1_64_64", 64, 64, 1, 1, 160), ("l1.conv3x3_64", 64, 64, 3, 1, 160),
    ("l1.conv1x1_64_256", 64, 256, 1, 1, 160), ("l1.conv1x1_256_64", 256, 64, 1, 1, 160),
    ("l2.0.conv1", 256, 128, 1, 1, 160), ("l2.0.conv2_s2", 128, 128, 3, 2, 160),
    ("l2.conv3", 128, 512, 1, 1, 80), ("l2.0.ds_s2", 256, 512, 1, 2, 160),
    ("l2.conv1", 512, 128, 1, 1, 80), ("l2.conv2", 128, 128, 3, 1, 80),
    ("l3.0.conv1", 512, 256, 1, 1, 80), ("l3.0.conv2_s2", 256, 256, 3, 2, 80),
    ("l3.conv3", 256, 1024, 1, 1, 40), ("l3.0.ds_s2", 512, 1024, 1, 2, 80),
    ("l3.conv1", 1024, 256, 1, 1, 40), ("l3.conv2", 256, 256, 3, 1, 40),
    ("l4.0.conv1", 1024, 512, 1, 1, 40), ("l4.0.conv2_s2", 512, 512, 3, 2, 40),
    ("l4.conv3", 512, 2048, 1, 1, 20), ("l4.0.ds_s2", 1024, 2048, 1, 2, 40),
    ("l4.conv1", 2048, 512, 1, 1, 20), ("l4.conv2", 512, 512, 3, 1, 20),
]
LIN = [("enc.proj", 3520, 256, 256), ("enc.ffn1", 3520, 256, 2048), ("enc.ffn2", 3520, 2048, 256),
       ("bert.qkv", 320, 768, 2304), ("bert.ffn1", 320, 768, 3072), ("bert.ffn2", 320, 3072, 768)]


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="")
    ap.add_argument("--batch", type=int, default=8)
    args = ap.parse_args()
    B = args.batch
    dev = "cuda"
    rows = []
    for name, ci, co, k, s, hin in R50:
        p = k // 2
        ho = (hin + 2 * p - k) // s + 1
        x = torch.randn(B, hin, hin, ci, device=dev).bfloat16()
        w = (torch.randn(co, k, k, ci, device=dev) / (ci * k * k) ** 0.5).bfloat16()
        wt = (torch.randn(ci, k, k, co, device=dev) / (ci * k * k) ** 0.5).bfloat16()
        dy = torch.randn(B, ho, ho, co, device=dev).bfloat16()
        dw = torch.zeros(co, k, k, ci, device=dev)
        geom = (B, hin, hin, ci, ho, ho, co, k, k, s, p)
        geom_t = (B, ho, ho, co, hin, hin, ci, k, k, s, p)
        flops = 2.0 * B * ho * ho * co * ci * k * k
        tf = timeit(lambda: hip.conv_gemm(x, w, geom=geom, act=hip.ACT_RELU))
        td = timeit(lambda: hip.conv_gemm(dy, wt, geom=geom_t, transposed=True))
        tw = timeit(lambda: hip.conv_wgrad(dy, x, dw, geom=geom))
        rows.append(dict(name=name, gflop=flops / 1e9, fwd_us=tf * 1e6, fwd_tf=flops / tf / 1e12,
                         dgrad_us=td * 1e6, dgrad_tf=flops / td / 1e12, wgrad_us=tw * 1e6, wgrad_tf=flops / tw / 1e12))
        print("%-18s %7.2f GF  fwd %8.1f us %7.1f TF | dgrad %8.1f us %7.1f TF | wgrad %8.1f us %7.1f TF" % (
            name, flops / 1e9, tf * 1e6, flops / tf / 1e12, td * 1e6, flops / td / 1e12, tw * 1e6, flops / tw / 1e12), flush=True)
    for name, M, K, N in LIN:
        x = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
        dy = torch.randn(M, N, device=dev).bfloat16(); dw = torch.zeros(N, K, device=dev)
        flops = 2.0 * M * N * K
        tf = timeit(lambda: hip.linear(x, w))
        tw = timeit(lambda: hip.linear_wgrad(dy, x, dw))
        rows.append(dict(name=name, gflop=flops / 1e9, fwd_us=tf * 1e6, fwd_tf=flops / tf / 1e12,
                         wgrad_us=tw * 1e6, wgrad_tf=flops / tw / 1e12))
        print("%-18s %7.2f GF  fwd %8.1f us %7.1f TF | wgrad %8.1f us %7.1f TF" % (
            name, flops / 1e9, tf * 1e6, flops / tf / 1e12, tw * 1e6, flops / tw / 1e12), flush=True)
    tot = {k: sum(r.get(k, 0) for r in rows[:len(R50)]) for k in ("fwd_us", "dgrad_us", "wgrad_us")}
    print("sum over unique R50 shapes (us):", tot)
    if args.out:
        os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
