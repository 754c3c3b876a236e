"""The training step AS IT RUNS: wall-clock stamps taken inside the replayed hipGraph, on both streams, un-profiled.

rocprofv3 serialises the concurrent streams of a replayed graph (one kernel resident 93 % of the step, 8.5 ms instead of 7.2), so
the phase files it gives describe a step that does not exist.  Here the step is captured with stamping on (reftr_amd.hip.mark ->
rt_stamp: a one-thread kernel node that writes the device's 100 MHz s_memrealtime clock), replayed REPS times, and the stamps are
read back after every replay.  Printed: per mark the median offset from the step's first stamp, per stream; the phase durations
between consecutive marks of one stream; the critical path; the same graph's step time without stamps for the perturbation.

usage: python tools/concurrent_timeline.py [--reps 50] [--out profiles/r04_concurrent_timeline.txt]
"""
import argparse
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from reftr_amd import hip  # noqa: E402
from reftr_amd.engine_vg import CapturedTrainStep  # noqa: E402
from reftr_amd.models import layout as Lm  # noqa: E402
from reftr_amd.models.criterion import CriterionVGMultiPhrase  # noqa: E402
from reftr_amd.models.reftr_transformer import RefTR  # noqa: E402
from reftr_amd.optim import FusedAdamW  # noqa: E402
from reftr_amd.util.misc import NestedTensor  # noqa: E402


def build(dev):
    model = RefTR(Lm.ModelConfig(), device=dev)
    wd = {"loss_giou": 2.0, "loss_bbox": 5.0}
    wd.update({f"{k}_{i}": v for i in range(5) for k, v in list(wd.items())})
    crit = CriterionVGMultiPhrase(wd, ["boxes"])
    model.store.P["bbox_embed.layers.2.weight"].normal_(0, 0.02); model.mark_dirty()
    opt = FusedAdamW(model)
    model.train()
    samples, targets = bench.synth_batch(8, 640, 640, 40, dev, 1234)
    s = {k: v.to(dev) for k, v in samples.items() if k not in ("img", "img_mask")}
    s["img"] = NestedTensor(samples["img"].to(dev), samples["img_mask"].to(dev))
    tg = [{k: v.to(dev) for k, v in t.items()} for t in targets]
    return model, crit, opt, s, tg


def time_steps(cap, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        cap(*cap.batch)
        float(cap.out[0])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=50)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    dev = torch.device("cuda")
    lines = []

    def say(s=""):
        print(s); lines.append(s)

    # plain graph first (same process, same box): the reference step time
    model, crit, opt, s, tg = build(dev)
    cap = CapturedTrainStep(model, crit, opt, 0.1, s, tg)
    time_steps(cap, 10)
    plain = time_steps(cap, args.reps)
    del cap, model, opt
    torch.cuda.empty_cache()

    marks = hip.enable_marks(dev)
    model, crit, opt, s, tg = build(dev)
    cap = CapturedTrainStep(model, crit, opt, 0.1, s, tg)
    names = dict(marks["names"])
    main_stream = names["step start"][1]
    time_steps(cap, 10)
    stamped = time_steps(cap, args.reps)
    samples = {n: [] for n in names}
    spans = []
    for _ in range(args.reps):
        cap(*cap.batch)
        torch.cuda.synchronize()
        host = marks["buf"].tolist()
        t0 = host[names["step start"][0]]
        for n, (slot, _st) in names.items():
            samples[n].append((host[slot] - t0) / 100.0)            # 100 MHz ticks -> microseconds
        spans.append((host[names["gradient norm done (step end)"][0]] - t0) / 100.0)
    hip.disable_marks()

    med = {n: statistics.median(v) for n, v in samples.items()}
    say(f"concurrent timeline of one replayed training step (configs[1]: R50, 640x640, B = 8, L = 40), median of {args.reps} replays")
    say(f"build {__import__('reftr_amd._build', fromlist=['x']).build_id()}; stamps: rt_stamp nodes inside the graph, s_memrealtime (10 ns)")
    say(f"step time, host clock, same process: {plain:.3f} ms without stamps, {stamped:.3f} ms with the {len(names)} stamp nodes "
        f"(first stamp -> last stamp: {statistics.median(spans) / 1e3:.3f} ms)")
    say()
    say("%10s  %-6s %s" % ("t [us]", "stream", "mark"))
    order = sorted(names, key=lambda n: med[n])
    for n in order:
        st = "main" if names[n][1] == main_stream else ("opt" if n.startswith("opt:") else "lang")
        say("%10.1f  %-6s %s" % (med[n], st, n))
    for label, pick in (("main", lambda n: names[n][1] == main_stream), ("lang", lambda n: names[n][1] != main_stream and not n.startswith("opt:")),
                        ("opt", lambda n: n.startswith("opt:"))):
        seq = [n for n in order if pick(n)]
        if len(seq) < 2:
            continue
        say()
        say(f"phases on the {label} stream (between consecutive marks of that stream):")
        for a, b in zip(seq, seq[1:]):
            say("%10.1f us  %s  ->  %s" % (med[b] - med[a], a, b))
    # critical path: at each join the later arrival is the one that gates
    say()
    fj, rf, lf = med.get("forward join (language branch in)"), med.get("ResNet forward done"), med.get("lang: map_sentence / map_phrase done", med.get("lang: positional / mask work done"))
    if fj is not None and rf is not None and lf is not None:
        say(f"forward join at {fj:.1f} us: main stream arrives at {rf:.1f}, language stream at {lf:.1f} -> gated by the "
            f"{'language' if lf > rf else 'main'} stream (slack of the other: {abs(lf - rf):.1f} us)")
    bj, rb = med.get("backward join (language branch in)"), med.get("ResNet backward done")
    lb = max((med[n] for n in names if n.startswith("lang:") and "backward" in n or "squared norm" in n), default=None)
    if bj is not None and rb is not None and lb is not None:
        say(f"backward join at {bj:.1f} us: main stream launched its last kernel at {rb:.1f}, language stream its last at {lb:.1f} -> gated by "
            f"the {'language' if lb > rb else 'main'} stream (slack of the other: {abs(lb - rb):.1f} us)")
    if args.out:
        with open(args.out, "w") as f:
            f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
