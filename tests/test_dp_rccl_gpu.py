"""GPU: the data-parallel wrapper + CapturedTrainStep + RCCL on the one-GPU box -- a single-rank `nccl` process group
(REFTR_DDP_FORCE=1) drives the complete schedule: one hipGraph per backward segment, the slice all-reduces issued eagerly
between replays (bf16 and fp32 exchange), clip + AdamW on the exchanged buffer.  With one rank the sums are the local
gradients, so the trajectory must follow the plain eager loop (to bf16 rounding of the gradients in the bf16 mode)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist

from test_model_gpu import build, make_inputs, rel, to_cuda

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


@pytest.fixture()
def single_rank_group(monkeypatch):
    monkeypatch.setenv("REFTR_DDP_FORCE", "1")
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", world_size=1, rank=0)
    try:
        yield
    finally:
        torch.cuda.synchronize()
        dist.destroy_process_group()


@pytest.mark.parametrize("dtype,schedule", [("bf16", "interleave"), ("fp32", "interleave"), ("bf16", "serial"), ("fp32", "serial")])
def test_dp_schedule_single_rank_rccl_follows_the_eager_loop(hip, single_rank_group, monkeypatch, dtype, schedule):
    from reftr_amd.engine_vg import CapturedTrainStep, train_step
    from reftr_amd.optim import FusedAdamW
    from reftr_amd.parallel import DistributedDataParallel
    monkeypatch.setenv("REFTR_DDP_DTYPE", dtype)
    monkeypatch.setenv("REFTR_DDP_SCHEDULE", schedule)
    names = ["main", "bert", "layer4"] if schedule == "serial" else ["main", "pair4", "pair3"]     # 2 BERT layers: no thirds
    samples, targets = make_inputs("e2e_single", B=2, H=96, W=128, L=12)
    s, tg = to_cuda(samples, targets)
    out = {}
    for mode in ("eager", "dp-eager", "dp-graph"):
        model, crit, P, ocfg = build(small=True)
        model.eval()
        opt = FusedAdamW(model, lr=1e-4, lr_backbone=1e-5, weight_decay=1e-4)
        runner = model
        if mode != "eager":
            runner = DistributedDataParallel(model)
            assert runner.active and runner.bf16 == (dtype == "bf16") and model.dp_mode
            assert list(runner.phase_bounds()) == names + ["end"]
        if mode == "dp-graph":
            p0, m0, v0 = model.store.flat_p.clone(), opt.m.clone(), opt.v.clone()
            cap = CapturedTrainStep(runner, crit, opt, 0.1, s, tg, warmup=1)
            assert not cap.deferred and cap.deferred_dp and cap.phases == names and len(cap.g_seg) == 3
            cap.reset_pending()                 # the warm-up iteration's (deferred) update is taken back with the weights
            model.store.flat_p.copy_(p0); opt.m.copy_(m0); opt.v.copy_(v0); opt.step_dev.zero_(); opt.step_count = 0
            model.mark_dirty(full=True)
        losses, first = [], None
        for it in range(3):
            if mode == "dp-graph":
                l, _, gn = cap(s, tg); lv = float(l)
            else:
                lv, _, _, gn = train_step(runner, crit, s, tg, opt, None, max_norm=0.1)
            losses.append(lv)
            if it == 0:
                if mode == "dp-graph":
                    cap.flush()                 # deferred optimizer: the update of this iteration lands now (as state_dict() would force it)
                torch.cuda.synchronize()
                g = model.store.flat_g16.float() if getattr(model.store, "flat_g16", None) is not None else model.store.flat_g.clone()
                first = (g, model.store.flat_p.clone(), float(gn))
        out[mode] = (losses, first)
    (le, fe), (l1, f1), (l2, f2) = out["eager"], out["dp-eager"], out["dp-graph"]
    tol_g = 2 ** -8 if dtype == "bf16" else 1e-6                        # the exchanged gradients: rounded once to bf16, or exact
    for l, f in ((l1, f1), (l2, f2)):
        assert abs(l[0] - le[0]) < 1e-6 * abs(le[0])
        assert rel(f[0], fe[0]) < tol_g and abs(f[2] - fe[2]) < max(tol_g, 1e-5) * fe[2]
        assert rel(f[1], fe[1]) < (2e-6 if dtype == "bf16" else 1e-7)   # weights after the first update
        # later steps: the fixture amplifies 1-ulp weight differences (tests/test_model_gpu.py::test_captured_step_matches_eager);
        # rounding the gradients to bf16 is a larger perturbation of the same kind
        t3 = 8e-2 if dtype == "bf16" else 3e-2
        assert abs(l[1] - le[1]) < 2e-3 * abs(le[1]) and abs(l[2] - le[2]) < t3 * abs(le[2]), (l, le)


def test_train_one_epoch_under_the_dp_wrapper(hip, single_rank_group):
    """The reference's entry point with the data-parallel wrapper: batches replayed from the per-segment graphs, collectives
    between replays, meters reduced across ranks (util/misc.py:136-160) -- runs and learns on a fixed-shape loader."""
    from reftr_amd.engine_vg import train_one_epoch
    from reftr_amd.optim import FusedAdamW
    from reftr_amd.parallel import DistributedDataParallel
    from reftr_amd.util.misc import NestedTensor
    b1 = make_inputs("e2e_single", B=2, H=96, W=128, L=12)
    loader = []
    for samples, targets in [b1] * 4:
        s = {k: v for k, v in samples.items() if k not in ("img", "img_mask")}
        s["img"] = NestedTensor(samples["img"], samples["img_mask"])
        loader.append((s, targets))
    model, crit, P, ocfg = build(small=True)
    opt = FusedAdamW(model, lr=1e-4, lr_backbone=1e-5, weight_decay=1e-4)
    ddp = DistributedDataParallel(model)
    stats = train_one_epoch(ddp, crit, loader, opt, None, torch.device("cuda"), 0, max_norm=0.1)
    caps = model._captured_steps
    assert len(caps) == 1 and not next(iter(caps.values())).deferred and next(iter(caps.values())).phases == ["main", "pair4", "pair3"]
    assert opt.step_count == 4 and stats["loss"] > 0 and stats["grad_norm"] > 0
    moved = (model.state_dict()["bbox_embed.layers.1.weight"].float().cpu() - P["bbox_embed.layers.1.weight"]).abs().max()
    assert 1e-5 < float(moved) < 1e-3


def test_bf16_vs_fp32_gradient_exchange_of_two_emulated_ranks_along_a_trajectory(hip):
    """Bounds what the default bf16 gradient exchange (reftr_amd/parallel.py) does to training, relative to the reference's
    fp32 DDP exchange (main_vg.py:290-296), with TWO ranks emulated on the one-GPU box.  Along a 6-step trajectory (driven by
    the fp32 exchange) every step computes the local gradients g0, g1 of two different shards on the same replica and forms
    the all-reduced buffer both ways, exactly as RCCL would -- fp32: g0 + g1; bf16: bf16(bf16(g0) + bf16(g1)), read by clip +
    AdamW straight from the bf16 buffer -- then applies clip(0.1) + AdamW (1/world folded in) from the SAME weights and
    optimizer state and compares the two updates: total gradient norm and the direction / size of the weight update.
    (Whole trajectories are not compared: on this fixture two fp32 runs already drift apart by 1e-2 in the loss after three
    steps through atomics-order noise, tests/test_model_gpu.py, and Adam's first steps are sign-like.)"""
    from reftr_amd.optim import FusedAdamW
    shards = [to_cuda(*make_inputs(f"dp_shard{r}", B=2, H=96, W=128, L=12)) for r in range(2)]
    model, crit, P, ocfg = build(small=True)
    model.eval()
    opt = FusedAdamW(model, lr=1e-4, lr_backbone=1e-5, weight_decay=1e-4)
    st = model.store
    model._grad_scale = 0.5
    g16 = torch.zeros_like(st.flat_g, dtype=torch.bfloat16)
    rows, losses = [], []
    for it in range(6):
        gs, lv = [], 0.0
        for s, tg in shards:
            out = model(s)
            ld = crit(out, tg)
            total = sum(ld[k] * crit.weight_dict[k] for k in ld if k in crit.weight_dict)
            st.flat_g.zero_()
            total.backward()
            gs.append(st.flat_g.clone()); lv += 0.5 * float(total)
        losses.append(lv)
        state = (st.flat_p.clone(), opt.m.clone(), opt.v.clone(), opt.step_dev.clone(), opt.step_count)

        def update(bf16):
            st.flat_p.copy_(state[0]); opt.m.copy_(state[1]); opt.v.copy_(state[2]); opt.step_dev.copy_(state[3]); opt.step_count = state[4]
            if bf16:
                g16.copy_(gs[0].bfloat16() + gs[1].bfloat16()); st.flat_g16 = g16
            else:
                st.flat_g.copy_(gs[0] + gs[1])
            try:
                gn = opt.clip_grad_norm_(0.1)
                opt.step()
                torch.cuda.synchronize()
                return float(gn), (st.flat_p - state[0]).double()
            finally:
                if bf16:
                    del st.flat_g16
        n16, d16 = update(True)
        n32, d32 = update(False)                 # the trajectory continues from the fp32 update
        rows.append((abs(n16 - n32) / n32, float((d16 * d32).sum() / (d16.norm() * d32.norm())), float((d16 - d32).norm() / d32.norm())))
    print("\n[bf16 vs fp32 exchange, 2 emulated ranks] per step (grad-norm rel, cos of the weight update, rel diff of the update): "
          + "  ".join(f"({a:.1e}, {b:.5f}, {c:.1e})" for a, b, c in rows) + f"  losses {['%.4f' % v for v in losses]}")
    assert losses[-1] < losses[0]
    for a, b, c in rows:
        assert a < 2e-3           # ||sum|| : the 2^-9 rounding errors average out over 10^8 elements
        assert b > 0.995 and c < 0.1


def test_rt_comm_single_rank_allreduce_through_the_c_abi(hip, single_rank_group):
    """rt_comm_* (include/reftr_hip.h): RCCL bound at run time -- the copy torch.distributed already loaded -- communicator of
    one rank, grouped in-place SUM all-reduce of fp32 and bf16 buffers on a side stream: a one-rank sum is the identity."""
    uid = hip.Comm.unique_id()
    assert len(uid) == 128
    comm = hip.Comm(uid, 0, 1)
    st = torch.cuda.Stream()
    for dt in (torch.float32, torch.bfloat16):
        a = torch.randn(1 << 20, device="cuda").to(dt); b = torch.randn(12345, device="cuda").to(dt)
        a0, b0 = a.clone(), b.clone()
        st.wait_stream(torch.cuda.current_stream())
        comm.allreduce([a, b, a[:0]], stream=st)
        torch.cuda.current_stream().wait_stream(st)
        torch.cuda.synchronize()
        assert torch.equal(a, a0) and torch.equal(b, b0)
    comm.destroy()


def test_dp_wrapper_on_the_c_abi_exchange_matches_the_torch_distributed_one(hip, single_rank_group, monkeypatch):
    """REFTR_COMM=abi: the same schedule with the library's own RCCL binding: three eager data-parallel steps give the same
    losses, gradient norm and weights as the torch.distributed exchange (single rank: both sums are the identity, bit for bit up
    to the atomics-order noise of two runs)."""
    from reftr_amd.engine_vg import train_step
    from reftr_amd.optim import FusedAdamW
    from reftr_amd.parallel import DistributedDataParallel
    samples, targets = make_inputs("e2e_single", B=2, H=96, W=128, L=12)
    s, tg = to_cuda(samples, targets)
    out = {}
    for mode in ("torch", "abi"):
        monkeypatch.setenv("REFTR_COMM", mode)
        model, crit, P, ocfg = build(small=True)
        model.eval()
        opt = FusedAdamW(model, lr=1e-4, lr_backbone=1e-5, weight_decay=1e-4)
        runner = DistributedDataParallel(model)
        assert (runner.comm is not None) == (mode == "abi")
        res, gn0 = [], None
        for it in range(3):
            res.append(train_step(runner, crit, s, tg, opt, None, max_norm=0.1))
            if it == 0:
                gn0 = float(res[0][3])        # the optimizer's device scalar: the next step overwrites it (read it now, not after step 3)
        torch.cuda.synchronize()
        out[mode] = ([r[0] for r in res], gn0, model.store.flat_p.clone())
        if runner.comm is not None:
            runner.comm.destroy()
    (lt, gt, pt), (la, ga, pa) = out["torch"], out["abi"]
    assert abs(lt[0] - la[0]) < 1e-6 * abs(lt[0]) and abs(gt - ga) < 2e-3 * gt      # atomics-order noise of two backward runs: 4e-4
    assert abs(lt[2] - la[2]) < 5e-2 * abs(lt[2]) and rel(pa, pt) < 2e-4


@pytest.mark.parametrize("kind", ["single", "multi"])
def test_bf16_exchange_twin_is_written_by_the_gradient_producers(hip, single_rank_group, monkeypatch, kind):
    """Round 4: the bf16 exchange buffer is filled by the weight-gradient launches themselves (rt_conv_wgrad_desc.g16:
    second-generation epilogues and reduction, the M <= 16 kernel, a rounding pass behind the first-generation launches) plus one
    chunk pass per slice over what they do not produce -- not by a 607 MB -> 304 MB copy.  After a backward through the wrapper (single
    rank: the all-reduce is the identity) the twin must be the bf16 image of the fp32 gradients, bit for bit, everywhere: matrices
    written once, matrices accumulated twice (multi-phrase: two BERT passes), matrices never written (cleared), biases / norms /
    embeddings."""
    from reftr_amd.engine_vg import _total, _zero_grad
    from reftr_amd.optim import FusedAdamW
    from reftr_amd.parallel import DistributedDataParallel
    monkeypatch.setenv("REFTR_DDP_DTYPE", "bf16")
    samples, targets = make_inputs("e2e_" + kind, B=2, H=96, W=128, L=12, n_phrase=3 if kind == "multi" else 0)
    s, tg = to_cuda(samples, targets)
    model, crit, P, ocfg = build(small=True)
    model.train()
    opt = FusedAdamW(model)
    runner = DistributedDataParallel(model)
    assert runner.active and runner.bf16 and runner.twin
    st = model.store
    st.flat_g.normal_(); st.flat_g16.normal_()                  # garbage from "the previous step" on both sides
    out = runner(s)
    total = _total(crit, crit(out, tg))
    _zero_grad(opt)
    total.backward()
    torch.cuda.synchronize()
    want = st.flat_g.to(torch.bfloat16)
    bad = (st.flat_g16 != want)
    assert int(bad.sum()) == 0, ("first mismatch at", int(bad.nonzero()[0]), "of", int(bad.sum()))
    assert float(st.flat_g.abs().sum()) > 0


def test_bf16_exchange_twin_in_full_clear_mode(hip, single_rank_group, monkeypatch):
    """with REFTR_OVERWRITE=0 (or optimizer.zero_grad() with fast=False) the store is never armed, so the twins of the
    matrices that no producer writes in a step (one query per image: the decoder's self-attention q / k projections) were left at
    the previous step's all-reduced values.  The full clear now clears the twins with the masters."""
    from reftr_amd.engine_vg import _total, _zero_grad
    from reftr_amd.optim import FusedAdamW
    from reftr_amd.parallel import DistributedDataParallel
    monkeypatch.setenv("REFTR_DDP_DTYPE", "bf16")
    monkeypatch.setenv("REFTR_OVERWRITE", "0")
    samples, targets = make_inputs("e2e_single", B=2, H=96, W=128, L=12)
    s, tg = to_cuda(samples, targets)
    model, crit, P, ocfg = build(small=True)
    model.train()
    opt = FusedAdamW(model)
    runner = DistributedDataParallel(model)
    assert runner.active and runner.bf16 and runner.twin
    st = model.store
    st.flat_g.normal_(); st.flat_g16.normal_()
    out = runner(s)
    total = _total(crit, crit(out, tg))
    _zero_grad(opt)
    assert not st._armed
    total.backward()
    torch.cuda.synchronize()
    bad = (st.flat_g16 != st.flat_g.to(torch.bfloat16))
    assert int(bad.sum()) == 0, ("first mismatch at", int(bad.nonzero()[0]), "of", int(bad.sum()))


def test_clip_norm_after_a_non_overlapped_fp32_exchange(hip, single_rank_group, monkeypatch):
    """DistributedDataParallel(overlap=False) with the fp32 exchange has neither stops nor phase hooks (dp_mode False) and
    the clip read the norm slots the weight-gradient epilogues filled BEFORE the all-reduce.  Any exchange now voids the slots; with
    one rank the norm must equal the plain loop's."""
    from reftr_amd.engine_vg import train_step
    from reftr_amd.optim import FusedAdamW
    from reftr_amd.parallel import DistributedDataParallel
    monkeypatch.setenv("REFTR_DDP_DTYPE", "fp32")
    samples, targets = make_inputs("e2e_single", B=2, H=96, W=128, L=12)
    s, tg = to_cuda(samples, targets)
    norms = []
    for wrap in (False, True):
        model, crit, P, ocfg = build(small=True)
        model.eval()
        opt = FusedAdamW(model, lr=1e-4, lr_backbone=1e-5, weight_decay=1e-4)
        runner = DistributedDataParallel(model, overlap=False) if wrap else model
        if wrap:
            assert runner.active and not runner.bf16 and not model.dp_mode
        _, _, _, gn = train_step(runner, crit, s, tg, opt, None, max_norm=0.1)
        torch.cuda.synchronize()
        if wrap:
            assert not model.store.norm_valid          # consumed / voided: the pass over the exchanged buffer ran
        norms.append((float(gn), model.store.flat_g.clone()))
    (n0, g0), (n1, g1) = norms
    ref = float(g1.double().pow(2).sum().sqrt())
    assert abs(n1 - ref) < 2e-5 * ref and abs(n1 - n0) < 2e-3 * n0      # two backward runs: atomics-order noise 4e-4
