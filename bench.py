"""RefTR training-step benchmark on MI355X (BASELINE.json metric: images/sec of one full training step,
RefCOCO-shaped batch, ResNet-50, 640x640, batch 8 per GPU).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = the loop body of engine_vg.train_one_epoch (engine_vg.py:40-72): forward, criterion, backward,
(N > 1: gradient all-reduce over RCCL), clip_grad_norm(0.1), AdamW, on one seeded synthetic batch that is
already resident in HBM (images N(0,1), half the batch padded on the right quarter, sentences
of 4..20 of L = 40 tokens, one box per image).  Random-init weights, dropout ON (train mode).  One process
per GPU, weak scaling (8 images per GPU); value = whole-job images/s from the max-over-ranks wall time.

Extra objects on the JSON line:
  roofline      MFMA roofline of the dominant kernel family (the implicit-GEMM conv/linear kernels
                rt_conv_gemm + rt_conv_wgrad): algorithmic 2*MAC FLOPs of their launches in one step divided by
                the summed launch durations, measured with HIP events on the launch stream in an instrumented
                pass of the same step right after the timed region (every rank runs the pass so that the
                collectives of N > 1 match; rank 0 reports its own events, collectives excluded);
                peak = 2.5 PFLOP/s dense bf16.  `traffic` = HBM bytes per launch from the rocprofv3 PMC passes
                (FETCH_SIZE / WRITE_SIZE, separate runs: benchmarks/round_artifacts.sh -> profiles/*_pmc_traffic.json);
                it is only reported when that file was taken on THIS build (build_id) with the same launch count.
  step_roofline the same ratio for the whole step (219.56 GFLOP/img from SURVEY.md §8d x img/s).
  cpu_baseline  the CPU oracle (oracle/reftr_oracle.py, the restatement pinned against the reference) running
                the same loop body on the host cores for a bounded sample (rank 0, N = 1 only).
`value` / `ms_per_step` come from the wall time of the K timed steps (the driver's contract); `ms_per_step_median` is the
median of the K per-step host timings (each step ends in the reference loop's `.item()` sync, engine_vg.py:53).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GF_PER_IMG = 219.56          # fwd+bwd matmul+conv GFLOP per image, cfg2 (SURVEY.md §8d / BASELINE.md §4)
PEAK_BF16_TFLOPS = 2500.0    # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md)
PEAK_HBM_BYTES_PER_S = 8.0e12  # HBM3E peak (MI355X_MICROARCH.md; ~6.3e12 achievable)


def synth_batch(B, H, W, L, device, seed):
    g = torch.Generator().manual_seed(seed)
    img = torch.randn(B, 3, H, W, generator=g)
    mask = torch.zeros(B, H, W, dtype=torch.bool)
    for b in range(0, B, 2):                      # landscape images: right quarter is padding
        wv = (W * 3) // 4
        mask[b, :, wv:] = True
        img[b, :, :, wv:] = 0
    ids = torch.zeros(B, L, dtype=torch.long)
    smask = torch.zeros(B, L, dtype=torch.long)
    for b in range(B):
        n = int(torch.randint(4, 21, (1,), generator=g))
        ids[b, :n] = torch.randint(1000, 30000, (n,), generator=g)
        ids[b, 0] = 101; ids[b, n - 1] = 102
        smask[b, :n] = 1
    targets = []
    for b in range(B):
        u = torch.rand(4, generator=g)
        box = torch.tensor([[0.3 + 0.4 * u[0], 0.3 + 0.4 * u[1], 0.1 + 0.4 * u[2], 0.1 + 0.4 * u[3]]])
        targets.append({"boxes": box, "labels": torch.zeros(1, dtype=torch.long)})
    samples = {"img": img, "img_mask": mask, "sentence": ids, "sentence_mask": smask}
    return samples, targets


def cpu_baseline(B, H, W, L, max_seconds=28.0, sample_batch=2, threads=32):
    """Times the oracle's train_step (fp32, dropout on, clip 0.1, AdamW) on the host cores, on a bounded sample of the same workload:
    the FULL batch (B images, like for like with the GPU step -- BASELINE.md section 3) when one warm-up step says that a warm-up + two
    timed steps fit `max_seconds`, else `sample_batch` images (the per-image cost of this path is close to batch-independent on CPU).
    The batch that was timed is reported in the result ("batch") and in the bench line's config."""
    from oracle import reftr_oracle as O
    from oracle.shapes import param_shapes
    from oracle.weights import formula_state
    cores = min(os.cpu_count() or 1, threads)
    torch.set_num_threads(cores)
    cfg = O.Cfg()
    P = formula_state(param_shapes(cfg))
    t_start = time.time()

    def run(batch, warm):
        samples, targets = synth_batch(batch, H, W, L, "cpu", 1234)
        state, step, warm_t = {}, 1, []
        for _ in range(warm):
            t0 = time.time()
            O.train_step(P, samples, targets, cfg, state, step, max_norm=0.1, train=True)
            warm_t.append(time.time() - t0)
            step += 1
        return samples, targets, state, step, warm_t

    batch, warm = B, 1
    samples, targets, state, step, warm_t = run(batch, warm)
    if B > sample_batch and warm_t[0] * 3.0 > max_seconds:          # the full batch does not fit the budget: the small sample
        batch, warm = sample_batch, 2
        t_start = time.time()
        samples, targets, state, step, warm_t = run(batch, warm)
    times = []
    while len(times) < 10 and (not times or time.time() - t_start + times[-1] < max_seconds):
        t0 = time.time()
        O.train_step(P, samples, targets, cfg, state, step, max_norm=0.1, train=True)
        times.append(time.time() - t0)
        step += 1
    import statistics
    med = statistics.median(times)                  # true median (mean of the middle two for an even count: ADVICE r05)
    return {"value": batch / med, "unit": "images/s", "cores": cores, "kind": "port", "batch": batch,
            "timed_steps_s": [round(t, 3) for t in times], "warmup_steps_s": [round(t, 3) for t in warm_t],
            "sample": f"median of {len(times)} timed step(s) after {warm} warm-up(s) on {batch} images of the same workload "
                      f"({H}x{W}, L={L}, fp32, dropout on, clip 0.1, AdamW), torch CPU threads = {cores}"}


def launch_decision(gpus, environ):
    """What `bench.py --gpus N` does with the environment it finds (host logic, covered on CPU by tests/test_host_logic_cpu.py):
      ("run", world)        this process is one rank of a `world`-rank job (or the single-GPU run);
      ("spawn", gpus)       --gpus N > 1 without a launcher: re-execute through torch.distributed.run, one rank per GPU
                            (the reference's launch: one process per GPU, main_vg.py:290-296, util/misc.py:392-431);
      ("refuse", message)   launcher and flag disagree -- never time a different number of GPUs than the one asked for."""
    if gpus < 1:
        return ("refuse", f"--gpus must be >= 1 (got {gpus})")
    ws = environ.get("WORLD_SIZE")
    if ws is None:
        return ("run", 1) if gpus == 1 else ("spawn", gpus)
    world = int(ws)
    if world != gpus:
        return ("refuse", f"--gpus {gpus} but the launcher started WORLD_SIZE={world} ranks: pass --gpus {world} "
                          f"(or start {gpus} ranks)")
    return ("run", world)


def spawn_ranks(gpus, argv):
    """Re-execute this script as `gpus` ranks on 127.0.0.1 (one per GPU, RCCL) and return the launcher's exit code; rank 0 of
    the child job prints the JSON line on the inherited stdout."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC only on these hosts (RCCL needs it)
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--size", type=int, default=640)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-roofline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying hipGraphs")
    args = ap.parse_args()

    what, arg = launch_decision(args.gpus, os.environ)
    if what == "refuse":
        print(f"[bench] {arg}", file=sys.stderr)
        sys.exit(2)
    if what == "spawn":
        sys.exit(spawn_ranks(arg, sys.argv[1:]))
    rank = int(os.environ.get("RANK", 0))
    world = arg
    local = int(os.environ.get("LOCAL_RANK", 0))
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU path exists for the product)"
    # REFTR_BENCH_ONE_DEVICE=1 + REFTR_DIST_BACKEND=gloo: every rank on GPU 0 with gloo carrying the (device-tensor) collectives --
    # NOT a measurement: a functional run of the whole N > 1 control flow (segment graphs, collectives between replays, deferred
    # optimizer, collective capture decisions) on a one-GPU box, where RCCL refuses two ranks on one device (benchmarks/dp_smoke_gloo.sh)
    one_dev = os.environ.get("REFTR_BENCH_ONE_DEVICE") == "1"
    if one_dev:
        local = 0
    assert torch.cuda.device_count() > local, f"rank {rank}: local rank {local} but only {torch.cuda.device_count()} GPU(s) visible"
    torch.cuda.set_device(local)
    force_dist = os.environ.get("REFTR_DDP_FORCE") == "1"        # one-GPU exercise of the data-parallel schedule
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend=os.environ.get("REFTR_DIST_BACKEND", "nccl"), init_method="env://", world_size=world, rank=rank)
    dev = torch.device("cuda", local)

    from reftr_amd import hip
    from reftr_amd.engine_vg import CapturedTrainStep, train_step
    from reftr_amd.models import layout as Lm
    from reftr_amd.models.criterion import CriterionVGMultiPhrase
    from reftr_amd.models.reftr_transformer import RefTR
    from reftr_amd.optim import FusedAdamW
    from reftr_amd.parallel import DistributedDataParallel
    from reftr_amd.util.misc import NestedTensor

    B, S_, Lq = args.batch, args.size, 40
    cfg = Lm.ModelConfig()
    model = RefTR(cfg, device=dev, aux_loss=True)
    wd = {"loss_giou": 1.0, "loss_bbox": 1.0}
    wd.update({f"{k}_{i}": v for i in range(cfg.dec_layers - 1) for k, v in list(wd.items())})
    crit = CriterionVGMultiPhrase(wd, ["boxes"])
    # a non-degenerate head so the loss has gradients everywhere (the reference zero-inits the last bbox layer)
    torch.manual_seed(1234)
    model.store.P["bbox_embed.layers.2.weight"].normal_(0, 0.02)
    model.mark_dirty()
    opt = FusedAdamW(model, lr=1e-4, lr_backbone=1e-5, weight_decay=1e-4)
    runner = DistributedDataParallel(model) if (world > 1 or force_dist) else model
    model.train()

    samples, targets = synth_batch(B, S_, S_, Lq, dev, 1234 + rank)
    s = {k: v.to(dev) for k, v in samples.items() if k not in ("img", "img_mask")}
    s["img"] = NestedTensor(samples["img"].to(dev), samples["img_mask"].to(dev))
    tg = [{k: v.to(dev) for k, v in t.items()} for t in targets]

    def eager_step():
        return train_step(runner, crit, s, tg, opt, None, max_norm=0.1)

    mode = "eager"
    step = eager_step
    if not args.no_graph:
        try:
            cap = CapturedTrainStep(runner, crit, opt, 0.1, s, tg)

            # the batch is resident: it lives in the graphs' static input buffers (an input pipeline writes there)
            sb, tb = cap.batch

            def step():
                losses, _, gn = cap(sb, tb)
                return (losses.item(), None, None, gn)   # same host sync as the reference loop (engine_vg.py:53)
            mode = "hipgraph"
        except Exception as e:                               # capture unsupported in this environment: stay eager
            if rank == 0:
                print(f"[bench] graph capture failed ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)

    for _ in range(args.warmup):
        step()
    dp = world > 1 or force_dist
    if dp:
        runner.timing = []                 # (event, event) around the end-of-backward waits: the exposed part of the exchange
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    trace, per_step = [], []
    tp = t0
    for _ in range(args.steps):
        loss_value = step()[0]
        trace.append(loss_value)
        tn = time.perf_counter(); per_step.append(tn - tp); tp = tn
    if os.environ.get("BENCH_TRACE") and rank == 0:
        print("[bench] losses:", " ".join("%.4g" % v for v in trace), file=sys.stderr)
        print("[bench] step_dev", int(opt.step_dev), "seed_dev", int(model.seed_dev), "gnorm", float(opt.grad_norm),
              "|p|", float(model.store.flat_p.norm()), "|g|", float(model.store.flat_g.norm()), "dirty", model._operands_dirty, file=sys.stderr)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    el = time.perf_counter() - t0
    t = torch.tensor([el], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    el = float(t)
    rccl = None
    if dp:
        exposed = [a.elapsed_time(b) for a, b in runner.timing]
        runner.timing = None
        rccl = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "exchange_dtype": "bf16" if runner.bf16 else "fp32",
                "bytes_per_rank_per_step": runner.exchange_bytes(), "schedule": model.dp_schedule, "boundaries": list(runner.phases),
                "exposed_exchange_ms": (sum(exposed) / len(exposed)) if exposed else None,
                "note": "exposed = time the compute stream waits for outstanding all-reduces after backward (HIP events, rank 0, mean over the timed steps)"}

    dc = getattr(model.net, "dec_counters", None)           # cooperative decoder launch (rt_decoder_fwd): word -1 = "a wait gave up"
    if dc is not None and int(dc[-1]) != 0:
        raise RuntimeError("rt_decoder_fwd: a stage wait gave up (a workgroup never arrived) -- the timed steps are not valid")
    import math
    assert math.isfinite(loss_value) and abs(loss_value) < 1e3, f"training step produced a non-finite / absurd loss ({loss_value})"
    ms_per_step = el / args.steps * 1e3
    value = B * world * args.steps / el

    roof = None
    if not args.no_kernel_roofline:
        recs = []
        # HIP events bracket every rt_conv_gemm / rt_conv_wgrad launch on the launch stream.  The stream is first blocked
        # by a spin kernel while the host enqueues the whole step (no host sync inside), so the kernels then run back to
        # back as they do under graph replay and the event pairs measure kernel time, not host launch gaps.  With N > 1
        # (or the forced single-rank exchange) EVERY rank runs this pass -- the eager loop body, whose collectives match
        # across ranks -- and rank 0 reports its own events; the all-reduces are not among the timed launches.
        side_was = model.net.side.enabled
        model.net.side.enabled = False          # one stream for this pass: per-launch durations must be additive
        if mode == "hipgraph" and not dp:
            torch.cuda.synchronize()
            torch.cuda._sleep(int(2.4e9 * 0.4))       # longer than the host needs to enqueue the whole eager step (~0.1-0.15 s of Python)
            hip.set_launch_timer(recs)
            cap._fwd_bwd(); cap._opt()
        else:
            if mode == "hipgraph":
                cap.flush()
            torch.cuda.synchronize()
            if dp:
                dist.barrier()
            torch.cuda._sleep(int(2.4e9 * 0.4))       # longer than the host needs to enqueue the whole eager step (~0.1-0.15 s of Python)
            hip.set_launch_timer(recs)
            eager_step()
        torch.cuda.synchronize()
        hip.set_launch_timer(None)
        model.net.side.enabled = side_was
        fl = sum(r["flops"] for r in recs)
        tm = sum(r["start"].elapsed_time(r["end"]) for r in recs) * 1e-3
        ach = fl / tm / 1e12
        if rank == 0:
            roof = {"bound": "mfma", "kernel": "rt_conv_gemm / rt_conv_wgrad(+_grouped) / rt_bottleneck_fwd kernels (conv_gemm_dma_kernel, w2_grouped_kernel + w2_reduce_kernel, conv_wgrad*_kernel, bottleneck_fwd_kernel, skinny / small-M)",
                    "achieved": ach, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_BF16_TFLOPS,
                    "traffic": None, "launches_per_step": len(recs), "avg_launch_us": tm / max(len(recs), 1) * 1e6,
                    "algorithmic_gflop_per_step": fl / 1e9, "kernel_ms_per_step": tm * 1e3,
                    "algorithmic_bytes_per_launch": sum(r["bytes"] for r in recs) / max(len(recs), 1),
                    # the elementwise operations fused into the GEMM epilogues move bytes of their own (residual / ReLU-gate /
                    # GELU' reads, second outputs): the reference runs them as separate kernels; `traffic` should be read
                    # against algorithmic + epilogue bytes
                    "epilogue_bytes_per_launch": sum(r.get("epi_bytes", 0.0) for r in recs) / max(len(recs), 1),
                    "measured_on": "rank 0, HIP events around every launch of the family on the launch stream"}
    from reftr_amd._build import build_id
    bid = build_id()
    if roof is not None:
        # HBM bytes per launch from the PMC passes of this same command (benchmarks/round_artifacts.sh -> tools/pmc_traffic.py;
        # FETCH_SIZE and WRITE_SIZE need separate rocprofv3 runs, so they cannot be collected inside the timed process).
        # Only a file taken on THIS build (same sources -> build_id) with the same number of launches per step is quoted.
        import glob
        note = "no profiles/*_pmc_traffic.json"
        for pmc in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
            t = json.load(open(pmc))
            if t.get("build_id") != bid:
                note = f"{os.path.basename(pmc)} was taken on build {t.get('build_id')}, this is {bid}: not quoted"
                continue
            if t.get("bench_launches_per_step") != len(recs) or t.get("workload") != [B, S_, world]:
                note = (f"{os.path.basename(pmc)}: {t.get('bench_launches_per_step')} launches/step for workload {t.get('workload')} "
                        f"vs {len(recs)} for {[B, S_, world]} here: not quoted")
                continue
            fam = t["gemm_family"]
            roof["traffic"] = fam["hbm_bytes_per_step"] / max(len(recs), 1)
            roof["traffic_bytes_per_step"] = fam["hbm_bytes_per_step"]
            roof["traffic_kernel_dispatches_per_step"] = fam["kernel_dispatches_per_step"]
            # the other roofline, side by side: the family's HBM bytes over the family's time against 8 TB/s
            roof["hbm_frac"] = fam["hbm_bytes_per_step"] / (roof["kernel_ms_per_step"] * 1e-3) / PEAK_HBM_BYTES_PER_S
            if fam.get("kernel_only_ms_per_step"):
                # kernel durations only (rocprofv3 --kernel-trace of the same command under graph replay): no inter-launch gaps
                roof["kernel_only_ms_per_step"] = fam["kernel_only_ms_per_step"]
                roof["kernel_only_frac"] = fl / (fam["kernel_only_ms_per_step"] * 1e-3) / 1e12 / PEAK_BF16_TFLOPS
            note = (f"HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) of the family's kernel dispatches per step / launches_per_step "
                    f"({os.path.basename(pmc)}, same build; one launch = one rt_conv_gemm / rt_conv_wgrad(_grouped) call, which may "
                    f"dispatch several kernels: tile kernel + split reduction)")
            break
        roof["traffic_note"] = note
    out = {
        "metric": "images/sec training step, RefCOCO R50 640x640 bs=8/GPU", "value": value, "unit": "images/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"RefCOCO-shaped REC train step, ResNet-50 + BERT-base + VL transformer 6+6, "
                               f"{S_}x{S_}, batch {B}/GPU, L=40, aux loss, dropout on, clip 0.1, AdamW (configs[1])",
                   "global_batch": B * world, "batch_per_gpu": B, "image_size": S_, "parallelism": f"dp{world}", "launch": mode + ("+dp-overlap" if (world > 1 or force_dist) else ""),
                   **({"grad_exchange": "bf16" if getattr(runner, "bf16", False) else "fp32"} if (world > 1 or force_dist) else {})},
        "loss": loss_value, "build_id": bid,
        # every switch of this package found in the environment: a run with any of them set is not the default configuration
        "env_overrides": {k: v for k, v in sorted(os.environ.items()) if k.startswith(("REFTR_", "BENCH_"))},
        "ms_per_step_median": sorted(per_step)[len(per_step) // 2] * 1e3,
    }
    if rank == 0:
        step_tf = value * GF_PER_IMG / 1e3
        out["step_roofline"] = {"bound": "mfma", "achieved": step_tf, "peak": PEAK_BF16_TFLOPS * world, "unit": "TFLOP/s",
                                "frac": step_tf / (PEAK_BF16_TFLOPS * world), "gflop_per_img": GF_PER_IMG}
        if roof is not None:
            out["roofline"] = roof
        if rccl is not None:
            out["rccl"] = rccl
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(B, S_, S_, Lq)
            out["config"]["cpu_baseline_batch"] = out["cpu_baseline"]["batch"]     # B when the full batch fits the time budget
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    # RCCL writes a banner ("Librccl path : ...") through C stdio, which is block-buffered on a pipe and would otherwise
    # surface AFTER the JSON line at exit: drain it first so that the JSON line is the last line of stdout
    import ctypes
    ctypes.CDLL(None).fflush(None)
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
