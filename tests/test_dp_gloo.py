"""CPU, world_size 2 over gloo: the data-parallel path (reftr_amd/parallel.py) — parameter broadcast from rank 0,
chunked all-reduce of the flat gradient buffer, 1/world folded into the optimizer's grad_scale, num_boxes
all-reduce of the criterion (models/criterion.py:176-180)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, q, dtype):
    os.environ["REFTR_DDP_DTYPE"] = dtype
    os.environ["REFTR_DDP_SCHEDULE"] = "serial" if dtype == "fp32" else "interleave"      # both exchange schedules get a run
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from reftr_amd.models import layout as L
        from reftr_amd.models.reftr_transformer import RefTR
        from reftr_amd.parallel import DistributedDataParallel
        cfg = L.ModelConfig(enc_layers=1, dec_layers=1, bert=L.BertConfig(layers=1))
        m = RefTR(cfg, device="cpu")
        m.reset_parameters(seed=100 + rank)                    # ranks start different ...
        ddp = DistributedDataParallel(m, n_chunks=5)
        ref = [torch.empty_like(m.store.flat_p) for _ in range(world)]
        dist.all_gather(ref, m.store.flat_p)
        same = all(torch.equal(ref[0], r) for r in ref)         # ... and are identical after the wrapper's broadcast
        bounds = ddp.chunk_bounds()
        covered = bounds[0][0] == 0 and bounds[-1][1] == m.store.flat_g.numel() and all(a[1] == b[0] for a, b in zip(bounds, bounds[1:]))
        g = torch.Generator().manual_seed(7 + rank)
        local = torch.randn(m.store.flat_g.numel(), generator=g)
        m.store.flat_g.copy_(local)
        pb = ddp.phase_bounds()                                  # boundary -> chunks of the slice that is final there
        serial = m.dp_schedule == "serial"
        assert list(pb) == list(m.active_boundaries()) + ["end"] == (["main", "bert", "layer4", "end"] if serial else ["main", "pair4", "pair3", "end"])   # 1 BERT layer: no thirds
        spans = sorted(c for v in pb.values() for c in v)
        covered = covered and spans[0][0] == 0 and spans[-1][1] == m.store.flat_g.numel() and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        from reftr_amd.models import layout as Lm
        bb = m.store.group_range[Lm.GROUP_BACKBONE]
        be = m.store.group_range[Lm.GROUP_BERT]
        l4 = m.store.offset["img_backbone.0.body.layer4.0.conv1.weight"][1]
        if serial:
            covered = covered and pb["end"][0][0] == bb[0] and pb["end"][-1][1] == l4 and pb["layer4"][0][0] == l4 and pb["layer4"][-1][1] == bb[1]
            covered = covered and pb["bert"][0][0] == be[0] and pb["bert"][-1][1] == be[1] and pb["main"][0][0] == 0
        else:       # pair4 = ResNet layer4 + (1 BERT layer: no thirds) the whole BERT slice; pair3 = layer3; end = layer2
            l3 = m.store.offset["img_backbone.0.body.layer3.0.conv1.weight"][1]
            p4 = sorted(pb["pair4"]); p3 = sorted(pb["pair3"])
            covered = covered and p4[0][0] == l4 and p4[-1][1] == be[1] and any(c[0] == be[0] for c in p4) and p3[0][0] == l3 and p3[-1][1] == l4
            covered = covered and sorted(pb["end"])[0][0] == bb[0] and sorted(pb["end"])[-1][1] == l3 and pb["main"][0][0] == 0
        for name in m.active_boundaries():                       # backward passes the boundaries in this order
            for hook in m._phase_hooks[name]:
                hook()
        for hook in m._post_backward_hooks:                      # end of backward: last slice + wait for everything
            hook()
        both = [torch.randn(m.store.flat_g.numel(), generator=torch.Generator().manual_seed(7 + r)) for r in range(world)]
        if dtype == "fp32":
            ok_sum = torch.allclose(m.store.flat_g, both[0] + both[1], atol=1e-6) and getattr(m.store, "flat_g16", None) is None
        else:       # bf16 exchange: every element rounded once per rank, summed in bf16; the fp32 buffer keeps the local gradients
            want = both[0].bfloat16().float() + both[1].bfloat16().float()
            got = m.store.flat_g16.float()
            ok_sum = (m.store.flat_g16.dtype == torch.bfloat16 and torch.allclose(got, want, rtol=2 ** -7, atol=1e-30)
                      and torch.equal(m.store.flat_g, local))
        nb = torch.tensor([3.0 + rank])
        dist.all_reduce(nb)
        q.put((rank, same, covered, ok_sum, m._grad_scale, float(nb / world)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("dtype", ["bf16", "fp32"])
def test_gradient_allreduce_world2_gloo(dtype):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, dtype)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, same, covered, ok_sum, gs, nb in res:
        assert same and covered and ok_sum
        assert gs == 0.5 and nb == 3.5


def _collectives_worker(rank, world, port, q):
    """Drives the product's own collective call sites with world > 1: CriterionVGMultiPhrase.num_boxes (C4,
    models/criterion.py:176-180), util.misc.reduce_dict (C5, util/misc.py:136-160) and the collective capture / replay /
    eager decision of engine_vg.captured_train_step."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from reftr_amd.engine_vg import dp_capture_decision
        from reftr_amd.models.criterion import CriterionVGMultiPhrase
        from reftr_amd.util import misc as utils
        crit = CriterionVGMultiPhrase({"loss_bbox": 1.0, "loss_giou": 1.0}, ["boxes"])
        # rank 0: 3 + 0 boxes, rank 1: 1 + 4 boxes -> global 8 boxes / 2 ranks = 4 per rank (NOT the local 3 or 5)
        counts = [(3, 0), (1, 4)][rank]
        targets = [{"labels": torch.zeros(n, dtype=torch.long), "boxes": torch.zeros(n, 4)} for n in counts]
        nb = crit.num_boxes(targets, torch.device("cpu"))
        # a static scalar (CapturedTrainStep refreshes it with its own all-reduce) short-cuts the collective
        crit.num_boxes_static = torch.tensor([7.0])
        nb_static = float(crit.num_boxes(targets, torch.device("cpu")))
        crit.num_boxes_static = None
        # an all-empty global batch: 0 / world (the clamp(min=1) is the kernel's, criterion.py:180)
        nb0 = crit.num_boxes([{"labels": torch.zeros(0, dtype=torch.long)}], torch.device("cpu"))
        red = utils.reduce_dict({"loss_giou": torch.tensor(1.0 + rank), "loss_bbox": torch.tensor(10.0 * (rank + 1))})
        summed = utils.reduce_dict({"a": torch.tensor(2.0 + rank)}, average=False)
        # rank 0 could replay, rank 1 sees a new shape -> everyone runs the eager body; both new -> capture; both hold one -> replay
        d1 = dp_capture_decision(rank == 0, rank == 1, torch.device("cpu"))
        d2 = dp_capture_decision(False, True, torch.device("cpu"))
        d3 = dp_capture_decision(True, False, torch.device("cpu"))
        d4 = dp_capture_decision(False, rank == 0, torch.device("cpu"))       # rank 1 is out of capture budget
        # the same collective carries the box-count average of the iteration (one all-reduce instead of two per replay)
        d5, nb5 = dp_capture_decision(True, False, torch.device("cpu"), num_boxes=sum(counts))
        assert d5 == "replay" and abs(float(nb5) - 4.0) < 1e-6, (d5, nb5)
        q.put((rank, float(nb), nb_static, float(nb0), float(red["loss_giou"]), float(red["loss_bbox"]), float(summed["a"]),
               (d1, d2, d3, d4)))
    finally:
        dist.destroy_process_group()


def test_criterion_num_boxes_reduce_dict_and_capture_decision_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_collectives_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, nb, nb_static, nb0, giou, bbox, a, dec in res:
        assert nb == 4.0 and nb_static == 7.0 and nb0 == 0.0
        assert giou == 1.5 and bbox == 15.0 and a == 5.0           # averaged / summed over the two ranks, same on both
        assert dec == ("eager", "capture", "replay", "eager")
