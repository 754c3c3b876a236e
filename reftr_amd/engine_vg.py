"""Train / eval loops with the reference's signatures (engine_vg.py:22-25,82): the loop body of
train_one_epoch is the hot path this library accelerates (engine_vg.py:40-72).

Differences that are deliberate and documented in INTEGRATION.md:
  * `optimizer` is reftr_amd.optim.FusedAdamW (clip + AdamW in one kernel); a plain torch optimizer also works;
  * the prefetcher only uses a side HIP stream when the device is a GPU (the reference constructs
    torch.cuda.Stream() unconditionally, engine_vg.py:240).
"""
import math
import contextlib
import os
import sys

import torch

from .util import misc as utils
from .util.box_ops import box_cxcywh_to_xyxy, box_iou


def _to_device(samples, targets, device, non_blocking=True):
    s = {k: (v.to(device, non_blocking=non_blocking) if hasattr(v, "to") else v) for k, v in samples.items()}
    t = [{k: (v.to(device, non_blocking=non_blocking) if hasattr(v, "to") else v) for k, v in tg.items()} for tg in targets]
    return s, t


class data_prefetcher:
    """H2D copies on a side stream, one batch ahead (engine_vg.py:234-291)."""

    def __init__(self, loader, device, prefetch=True):
        self.loader = iter(loader)
        self.device = torch.device(device)
        self.prefetch = prefetch and self.device.type == "cuda"
        if self.prefetch:
            self.stream = torch.cuda.Stream()
            self.preload()

    def preload(self):
        try:
            s, t = next(self.loader)
        except StopIteration:
            self.next_samples = self.next_targets = None
            return
        with torch.cuda.stream(self.stream):
            self.next_samples, self.next_targets = _to_device(s, t, self.device)

    def next(self):
        if not self.prefetch:
            try:
                s, t = next(self.loader)
            except StopIteration:
                return None, None
            return _to_device(s, t, self.device, non_blocking=False)
        torch.cuda.current_stream().wait_stream(self.stream)
        s, t = self.next_samples, self.next_targets
        if s is not None:
            for v in s.values():
                if hasattr(v, "record_stream"):
                    v.record_stream(torch.cuda.current_stream())
            for tg in t:
                for v in tg.values():
                    if hasattr(v, "record_stream"):
                        v.record_stream(torch.cuda.current_stream())
        self.preload()
        return s, t


def _total(criterion, loss_dict):
    """engine_vg.py:43.  The HIP criterion offers the same weighted sum as one fused reduction."""
    if hasattr(criterion, "weighted_total") and os.environ.get("REFTR_FUSED_TOTAL", "1") != "0":
        return criterion.weighted_total(loss_dict)
    wd = criterion.weight_dict
    return sum(loss_dict[k] * wd[k] for k in loss_dict.keys() if k in wd)


def _zero_grad(optimizer):
    """engine_vg.py:60.  The fused optimizer clears only what backward accumulates with atomics (the weight matrices are
    overwritten by their producers); any other optimizer: the plain zero_grad()."""
    if isinstance(getattr(optimizer, "model", None), torch.nn.Module) and hasattr(optimizer, "apply_pending"):
        optimizer.zero_grad(fast=True)
    else:
        optimizer.zero_grad()


def _inner(model):
    return getattr(model, "module", model)


def _seed_state(model):
    """(device dropout step-seed word, host step count) BEFORE this iteration's forward advanced them (train_step reads them
    after the forward: one step back)."""
    inner = _inner(model)
    if not hasattr(inner, "seed_dev"):
        return None
    return (1 if inner.training else 0, 1)


def _restore_seed(model, back):
    """Takes back the dropout-seed / step advance of an iteration that is run again."""
    inner = _inner(model)
    if back is None:
        return
    if back[0]:
        inner.seed_dev.sub_(back[0])
    inner._step -= back[1]


def _cooperative_failed(model):
    """True when a cooperative decoder launch of THIS process -- or, under data parallelism, of any rank (the choice to run the
    iteration again must be collective) -- raised its failure word since it was last cleared.  One host read (eager loop only; the
    replayed loop carries the word in its per-iteration stats vector)."""
    inner = _inner(model)
    fw = inner.coop_failure_word() if hasattr(inner, "coop_failure_word") else None
    if fw is None:                                     # no cooperative launches in this build / configuration: the same on every rank
        return False
    launched = getattr(inner.net, "dec_counters", None) is not None      # rank-local (depends on this rank's batch shape)
    if utils.is_dist_avail_and_initialized() and utils.get_world_size() > 1:
        # every rank enters the collective; a rank that has not launched the cooperative decoder yet contributes 0
        f = fw.to(torch.int32).clone() if launched else torch.zeros_like(fw, dtype=torch.int32)
        torch.distributed.all_reduce(f, op=torch.distributed.ReduceOp.MAX)
        return int(f) != 0
    return launched and int(fw) != 0


_COOP_LOGGED = False


def _coop_fallback(model):
    """Switches the cooperative decoder launches off for this process (the launched chain computes the same values), clears the
    failure word, and drops every captured step / forward of the model: they contain the launches.  Pending (deferred) updates of
    the dropped captures are discarded -- the caller runs the failed iteration again."""
    global _COOP_LOGGED
    inner = _inner(model)
    net = inner.net
    net.dec_coop = net.dec_coop_bwd = False
    os.environ["REFTR_DEC_COOP"] = "0"; os.environ["REFTR_DEC_COOP_BWD"] = "0"
    if net._dec_handoff is not None:
        net._dec_handoff[1:2].zero_()
    net.dec_counters = None
    caps = inner.__dict__.get("_captured_steps") or {}
    for cap in caps.values():
        cap.reset_pending()
        if getattr(cap.optimizer, "veto", None) is not None:
            cap.optimizer.veto = None
    caps.clear()
    inner.__dict__["_staged_cap"] = None
    inner.__dict__["_coop_generation"] = inner.__dict__.get("_coop_generation", 0) + 1     # CapturedForward drops its graphs
    if not _COOP_LOGGED:
        _COOP_LOGGED = True
        print("[reftr_amd] rt_decoder_fwd / rt_decoder_bwd: a stage hand-off timed out (workgroups not co-resident?). The iteration "
              "is discarded and run again; the decoder uses the launched chain (REFTR_DEC_COOP=0) from here on.", file=sys.stderr)


def train_step(model, criterion, samples, targets, optimizer, lr_scheduler=None, max_norm=0.0):
    """The loop body, engine_vg.py:40-72.  Returns (loss_value, reduced scaled dict, reduced unscaled dict,
    grad_norm tensor)."""
    if hasattr(optimizer, "veto") and hasattr(_inner(model), "coop_failure_word"):
        optimizer.veto = _inner(model).coop_failure_word()
    outputs = model(samples)
    loss_dict = criterion(outputs, targets)
    weight_dict = criterion.weight_dict
    losses = _total(criterion, loss_dict)
    loss_dict_reduced = utils.reduce_dict(loss_dict)
    unscaled = {f"{k}_unscaled": v for k, v in loss_dict_reduced.items()}
    scaled = {k: v * weight_dict[k] for k, v in loss_dict_reduced.items() if k in weight_dict}
    loss_value = sum(scaled.values()).item()
    if not math.isfinite(loss_value):
        print("Loss is {}, stopping training".format(loss_value))
        print(loss_dict_reduced)
        sys.exit(1)
    _zero_grad(optimizer)
    seed0 = _seed_state(model)
    losses.backward()
    if _cooperative_failed(model):
        # a stage hand-off of the cooperative decoder launches timed out: this iteration's activations / gradients are void.
        # The reference stops BEFORE the update when an iteration is bad (engine_vg.py:55-58); here the launches are switched
        # off for the process and the iteration is run again on the launched chain (same dropout seeds), then updated as usual
        _coop_fallback(model)
        _restore_seed(model, seed0)
        outputs = model(samples)
        loss_dict = criterion(outputs, targets)
        losses = _total(criterion, loss_dict)
        loss_dict_reduced = utils.reduce_dict(loss_dict)
        unscaled = {f"{k}_unscaled": v for k, v in loss_dict_reduced.items()}
        scaled = {k: v * weight_dict[k] for k, v in loss_dict_reduced.items() if k in weight_dict}
        loss_value = sum(scaled.values()).item()
        _zero_grad(optimizer)
        losses.backward()
    if hasattr(optimizer, "clip_grad_norm_"):
        grad_total_norm = optimizer.clip_grad_norm_(max_norm)
    elif max_norm > 0:
        grad_total_norm = torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm)
    else:
        grad_total_norm = torch.zeros(())
    optimizer.step()
    if lr_scheduler is not None:
        lr_scheduler.step()
    return loss_value, scaled, unscaled, grad_total_norm


def _clone_batch(samples, targets):
    def c(v):
        if isinstance(v, utils.NestedTensor):
            return utils.NestedTensor(v.tensors.clone(), v.mask.clone())
        return v.clone() if torch.is_tensor(v) else v
    return {k: c(v) for k, v in samples.items()}, [{k: c(v) for k, v in t.items()} for t in targets]


def _copy_batch(dst_s, dst_t, samples, targets):
    if samples is dst_s and targets is dst_t:        # the caller filled the static buffers in place (CapturedTrainStep.batch)
        return
    dst, src = [], []
    for k, v in samples.items():
        if isinstance(v, utils.NestedTensor):
            dst += [dst_s[k].tensors, dst_s[k].mask]; src += [v.tensors, v.mask]
        elif torch.is_tensor(v):
            dst.append(dst_s[k]); src.append(v)
    for d, t in zip(dst_t, targets):
        for k, v in t.items():
            if torch.is_tensor(v):
                dst.append(d[k]); src.append(v)
    # device -> device with matching dtypes (the steady state of a loop: batches come from the prefetcher): ONE multi-tensor
    # copy for the ~20 small fields instead of a launch each -- they sit on the critical path between two replays
    big = [i for i, (a, b) in enumerate(zip(dst, src)) if a.numel() >= (1 << 20) or a.dtype != b.dtype or a.device != b.device
           or a.shape != b.shape]
    for i in big:
        dst[i].copy_(src[i], non_blocking=True)
    groups = {}
    for i in range(len(dst)):
        if i not in set(big):
            groups.setdefault(dst[i].dtype, []).append(i)
    for idx in groups.values():                       # one launch per dtype (the multi-tensor fast path wants uniform lists)
        torch._foreach_copy_([dst[i] for i in idx], [src[i] for i in idx], non_blocking=True)


# Stream capture is made thread-local: torch.distributed's RCCL watchdog thread polls hipEventQuery on its own schedule,
# which a process-global capture would turn into hipErrorStreamCaptureUnsupported (and an abort) at world_size > 1.
_CAPTURE_MODE = "thread_local"


@contextlib.contextmanager
def _capture(graph, **kw):
    """`torch.cuda.graph(...)` with Python's cyclic garbage collector switched OFF for the duration of the capture.  A step's capture
    runs ~1500 Python-level launches; a generation-2 collection in the middle of it can finalize an OLDER CapturedTrainStep (its graphs,
    its private pool, its events) -- device frees and graph destruction inside an open capture.  Seen in round 6 as `Fatal Python error:
    Aborted` with `Garbage-collecting` on the stack inside `flush_wgrads_side` during a capture, and as segfaults in later replays, once the
    GPU suite held enough dead captures; torch's own context collects BEFORE the capture but does not stop a collection during it."""
    import gc
    was = gc.isenabled()
    gc.collect()
    gc.disable()
    try:
        with torch.cuda.graph(graph, capture_error_mode=_CAPTURE_MODE, **kw):
            yield
    finally:
        if was:
            gc.enable()


class CapturedTrainStep:
    """The loop body (engine_vg.py:40-72) captured into HIP graphs for one input shape.

    A step is ~1500 kernel launches; replaying them from a hipGraph removes the per-launch host cost.  Everything
    that changes from step to step lives in device memory (dropout step-seed word, optimizer step counter, the
    input batch copied into static buffers), so replays are real training steps.  With world_size > 1 the body is
    three graphs (forward + backward phase 1 | ResNet backward | clip + AdamW) around the two eager, asynchronous
    gradient exchanges of reftr_amd.parallel.DistributedDataParallel.  Shapes other than the captured one must use
    `train_step`.

    Single-process runs use the DEFERRED optimizer schedule (one graph): iteration i ends with the gradient norm, its
    AdamW update is applied at the head of iteration i+1 -- the ResNet / transformer slice first on the main stream, the
    BERT slice (72 % of the 4.25 GB pass) on the language stream in front of the BERT operand refresh, concurrently with
    the ResNet forward.  Same arithmetic, same order of updates; whoever reads the weights or the optimizer state in
    between (`model(...)` outside the replay, `state_dict()`, `optimizer.step()`) first gets the pending update applied
    (`flush()`).  The learning rates are device words (`optimizer.lr_dev`): the update of iteration i uses the rates that
    were current when iteration i ran, and schedule changes need no re-capture.
    """

    def __init__(self, model, criterion, optimizer, max_norm, samples, targets, warmup=2, force_two_phase=False, force_phases=None):
        self.model, self.criterion, self.optimizer, self.max_norm = model, criterion, optimizer, max_norm
        self.inner = getattr(model, "module", model)
        self.ddp = model if model is not self.inner else None
        self.s, self.t = _clone_batch(samples, targets)
        self.key = self.shape_key(samples, targets)
        self._lrs = [g["lr"] for g in optimizer.param_groups]
        inner = self.inner
        # the cooperative decoder's failure word vetoes an iteration's update on the device and travels in the stats vector
        self.fail_word = inner.coop_failure_word() if hasattr(inner, "coop_failure_word") else None
        if hasattr(optimizer, "veto"):
            optimizer.veto = self.fail_word
        warmup = max(warmup, 2)          # the second step is the first one with the steady-state operand refresh
        # criterion.py:176-180 averages the box count over ranks: that collective runs eagerly before every replay and
        # the graph reads its result from a static device scalar
        self.nb = None
        if utils.is_dist_avail_and_initialized():
            self.nb = torch.zeros(1, dtype=torch.float32, device=self.s["sentence"].device)
            self._refresh_num_boxes(targets)
            criterion.num_boxes_static = self.nb
        # collectives stay outside the graphs: the engine calls the hooks between replays
        self._phase_hooks, inner._phase_hooks = inner._phase_hooks, {}
        self._post, inner._post_backward_hooks = inner._post_backward_hooks, []
        # data parallel: backward is captured as one graph per segment between the exchange points the wrapper registered
        # (RefTR.BOUNDARIES), so that each slice's all-reduce is issued the moment it is final and runs under the next graphs
        self.phases = [b for b in inner.active_boundaries() if self._phase_hooks.get(b)]
        if force_phases is not None:
            self.phases = [b for b in inner.active_boundaries() if b in force_phases]
        elif force_two_phase and not self.phases:
            # serial schedule: [everything but the ResNet | the ResNet]; interleaved: [... transformer backward | BERT || ResNet]
            self.phases = ["bert"] if inner.dp_schedule == "serial" else ["main"]
        self.two_phase = bool(self.phases)
        inner._stops = frozenset(self.phases)
        can_defer = hasattr(optimizer, "enable_deferred") and os.environ.get("REFTR_DEFER_OPT", "1") == "1"
        self.deferred = can_defer and not self.two_phase and not self._post
        # data parallel (round 3): the same deferred schedule across the segment graphs -- the AdamW pass of iteration i sits at
        # the head of iteration i+1's FIRST graph (its BERT slice on the language stream under the ResNet forward, reading the
        # all-reduced bf16 gradients of iteration i, which nothing rewrites before this iteration's first exchange boundary), the
        # last graph ends with the gradient norm.  (Optimizers without the deferred interface: clip + step in the last graph.)
        self.deferred_dp = can_defer and not self.deferred
        self._pending = False
        self._staged = None
        if self.deferred:
            self._init_deferred(warmup)
            return
        if hasattr(optimizer, "enable_device_lr"):
            optimizer.enable_device_lr()
        if self.deferred_dp:
            self._deferred_hooks()
        head = self._head_deferred if self.deferred_dp else self._fwd_bwd
        tail = self._tail_deferred if self.deferred_dp else self._opt
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(warmup):
                    head()
                    for name in self.phases:
                        self._run(self._phase_hooks.get(name, ()))
                        inner.continue_backward()
                    self._run(self._post); tail()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()     # no collective in flight while capturing (see _CAPTURE_MODE)
            self.g_fb = torch.cuda.CUDAGraph()
            with _capture(self.g_fb):
                self.out = head()
            self.g_seg = []
            for name in self.phases:                 # the segment that FOLLOWS boundary `name`
                g = torch.cuda.CUDAGraph()
                with _capture(g, pool=self.g_fb.pool()):
                    stopped = inner.continue_backward()
                self.g_seg.append(g)
            assert inner._bwd_gen is None, "backward did not run to its end during capture"
            self.g_bb = self.g_seg[-1] if self.g_seg else None
            self.g_opt = torch.cuda.CUDAGraph()
            with _capture(self.g_opt, pool=self.g_fb.pool()):
                tail()
            if self.deferred_dp:
                self.grad_norm = optimizer.grad_norm
                self._pending = True                   # the last warm-up iteration's update
                self._set_flush(True)
        finally:
            inner._phase_hooks, inner._post_backward_hooks = self._phase_hooks, self._post
            inner._stops = frozenset()
            criterion.num_boxes_static = None

    # ------------------------------------------------------------------ deferred optimizer schedule
    def _deferred_hooks(self):
        from .models import layout as L
        inner, opt = self.inner, self.optimizer
        opt.enable_deferred()
        bb, be = inner.store.group_range[L.GROUP_BERT]
        assert be == inner.store.flat_p.numel() or be > bb, "BERT is the last group of the flat buffers"
        self._hooks = ((lambda: opt.apply_pending(span=(0, bb)) if bb > 0 else None),
                       (lambda: opt.apply_pending(span=(bb, be))))

    def _head_deferred(self):
        """[AdamW of the previous iteration | forward | loss | backward up to the first boundary]"""
        inner = self.inner
        self._set_flush(False)
        inner._pre_update = self._hooks
        try:
            return self._fwd_bwd()
        finally:
            inner._pre_update = None

    def _tail_deferred(self):
        self.grad_norm = self.optimizer.finish_step(self.max_norm)
        self._pack_stats()

    def _init_deferred(self, warmup):
        inner, opt = self.inner, self.optimizer
        self._deferred_hooks()
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(warmup):
                    self._step_deferred()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self.g_fb = torch.cuda.CUDAGraph()
            with _capture(self.g_fb):
                self.out = self._step_deferred()
            self.g_bb = self.g_opt = None
            self.grad_norm = opt.grad_norm
            if getattr(opt, "_emit", False):
                opt._emit_tables(None)                 # flush() applies the update over the WHOLE buffer: its job / chunk tables are built
                                                       # now (a pageable host -> device copy later would stall the host behind queued work)
            self._pending = True                       # the last warm-up iteration's update
            self._set_flush(True)
        finally:
            inner._phase_hooks, inner._post_backward_hooks = self._phase_hooks, self._post
            inner._stops = frozenset()
            self.criterion.num_boxes_static = None

    def _set_flush(self, on):
        f = self.flush if on else None
        self.inner._flush_pending = f
        self.optimizer._flush_pending = f

    def _step_deferred(self):
        """[AdamW of the previous iteration | forward | loss | backward | gradient norm] -- eagerly or under capture."""
        inner, opt = self.inner, self.optimizer
        self._set_flush(False)
        inner._pre_update = self._hooks
        # (the gradient clear stays between loss and backward on the main stream: on the language stream under the encoder it was
        # neutral, 6.67-6.69 vs 6.66-6.68 ms, profiles/r04ai_zero_side_ab.txt and again +0.06 ms in round 5 -- option removed)
        # backward and clip norm are one unit here (nothing touches the gradient buffer in between): the BERT slice's share of the
        # norm may be taken on the language stream as soon as that slice is final (reftr_transformer._backward_gen)
        inner._norm_side = not getattr(inner.store, "fused_norm", False)
        try:
            out = self._fwd_bwd(zero=True)
            self.grad_norm = opt.finish_step(self.max_norm, loss=out[0], stats=self._stats_spec())
            from . import hip as _H
            _H.mark("gradient norm done (step end)")
        finally:
            inner._pre_update = None
            inner._norm_side = False
            inner._norm_split = None
        self._pack_stats()
        return out

    def flush(self):
        """Applies the pending update now (before an eager forward, a checkpoint, an optimizer.step())."""
        if not ((self.deferred or self.deferred_dp) and self._pending):
            return
        self._pending = False
        self._set_flush(False)
        emitted = self.optimizer.apply_pending()
        self.optimizer.clear_pending()
        if emitted:
            self.inner.operands_emitted()
        else:
            self.inner.mark_dirty()

    def reset_pending(self):
        """Forgets the pending update (after / before the caller restores weights / optimizer state by hand).  The bf16 operands are
        marked stale: the next replay rebuilds them from whatever the masters then hold (the graphs carry no operand refresh of
        their own since the optimizer's pass writes the operands)."""
        self._pending = False
        self.inner.mark_dirty()
        if self.deferred or self.deferred_dp:
            self.optimizer.clear_pending()
            self._set_flush(False)

    def _refresh_num_boxes(self, targets, reduced=None):
        if self.nb is None:
            return
        if reduced is not None:               # already averaged over the ranks by the iteration's capture decision (one collective)
            self.nb.copy_(reduced)
            return
        self.nb.fill_(float(sum(len(t["labels"]) for t in targets)))
        torch.distributed.all_reduce(self.nb)
        self.nb.div_(utils.get_world_size())

    @staticmethod
    def shape_key(samples, targets):
        k = []
        for n, v in sorted(samples.items()):
            k.append((n, tuple(v.tensors.shape) if isinstance(v, utils.NestedTensor) else tuple(v.shape)))
        # every tensor field of the targets is copied into the static batch (_copy_batch) and read by the criterion at its
        # captured shape: boxes, labels, masks (RefTRSeg), sizes ... all belong to the key
        for t in targets:
            k.append(tuple((n, tuple(v.shape)) for n, v in sorted(t.items()) if torch.is_tensor(v)))
        return tuple(k)

    def _direct_loss_ok(self):
        """The direct loss path applies when the total is the weighted sum of the box losses of THIS process's model: no wrapper
        (the data-parallel schedules drive backward themselves), no mask / CEM terms, the fused total, aux weights as usual."""
        inner, crit = self.inner, self.criterion
        # (round 5 built this path under the data-parallel wrapper too and measured it through single-rank RCCL: 7.58-7.63 ms against
        # 7.17-7.22 ms on the autograd path -- the early fork of the target preparation costs the segment graphs more than the launches it
        # removes; profiles/r05_ddp_direct_loss_negative_result.txt.  Removed.)
        dp_ok = self.model is inner and not inner.dp_mode
        return (os.environ.get("REFTR_LOSS_DIRECT", "1") == "1" and dp_ok and getattr(inner, "seg", 1) is None
                and hasattr(crit, "loss_and_grad") and tuple(crit.losses) == ("boxes",)
                and os.environ.get("REFTR_FUSED_TOTAL", "1") != "0")

    def _fwd_bwd_direct(self, zero):
        inner, crit = self.inner, self.criterion
        dev = inner.store.device
        side = inner.net.side
        main = torch.cuda.current_stream()
        prepared = side.run(lambda: crit.prepare(self.t, dev))            # reads the targets only: beside the step head, not behind the forward
        # round 5: with the fused head (rt_head_loss) the forward itself ends with the losses and the head's backward-data; it needs
        # the prepared targets (ready at the forward join, where the side stream comes in)
        inner.__dict__["_head_fused"] = (crit, prepared) if os.environ.get("REFTR_HEAD_FUSE", "1") != "0" else None
        try:
            logits = inner(self.s, _logits_only=True)                      # joins the side stream at its forward join
        finally:
            inner.__dict__["_head_fused"] = None
        if side.enabled:
            for t in prepared:
                t.record_stream(main)
        head = inner._saved.get("head")
        if head is not None:
            box, dl = head["losses"], None                                 # backward starts behind the head (RefTR._backward_phases)
            loss_dict = crit._loss_dict(box)
            losses = head["total"][0]                                      # the weighted total (engine_vg.py:43), from the same launch
        else:
            loss_dict, box, dl = crit.loss_and_grad(logits, inner._saved["phrase_mask"], prepared, inner.aux_loss)
            # the weighted total is only logged (engine_vg.py:43,46-53): the same two torch kernels as in the autograd path.  On the
            # main stream: forked to the side stream here (measured) the replayed graph's whole backward slows down by 0.25 ms
            losses = crit.weighted_total(loss_dict)
        if zero:
            _zero_grad(self.optimizer)
        inner._backward_impl(dl)                                           # ends with the side stream joined
        self._last = (losses.detach(), {k: v.detach() for k, v in loss_dict.items()})
        return self._last

    def _fwd_bwd(self, zero=True):
        if self._direct_loss_ok():
            return self._fwd_bwd_direct(zero)
        outputs = self.model(self.s)
        loss_dict = self.criterion(outputs, self.t)
        losses = _total(self.criterion, loss_dict)
        if zero:
            _zero_grad(self.optimizer)
        losses.backward()
        self._last = (losses.detach(), {k: v.detach() for k, v in loss_dict.items()})
        return self._last

    def _stats_spec(self):
        """The stats vector as rt_finish_stats writes it (deferred single-process schedule): (loss scalars sorted by name, whether the
        failure word rides along, the static output vector) -- or None when a loss is not an fp32 device scalar (then _pack_stats)."""
        ld = self._last[1]
        names = tuple(sorted(ld))
        srcs = [ld[k].reshape(1) for k in names]
        if len(srcs) > 40 or any(not (t.is_cuda and t.dtype == torch.float32) for t in srcs):
            self._stats_fused = False
            return None
        n = len(srcs) + (1 if self.fail_word is not None else 0) + 1
        if getattr(self, "stats", None) is None or self.stats.numel() != n or not self.stats.is_cuda:
            self.stats = torch.empty(n, dtype=torch.float32, device=srcs[0].device)
        self.stat_names = names
        self._stats_fused = True
        return srcs, self.fail_word is not None, self.stats

    def _pack_stats(self):
        """Every scalar the loop reads from an iteration -- the unweighted losses (sorted by name) and the gradient norm -- in
        ONE static device vector (a single small kernel at the end of the graph): the engine fetches it with one device -> host
        copy per iteration instead of one .item() per meter (the reference: engine_vg.py:46-53,69-72, util/misc.py:156-160)."""
        if getattr(self, "_stats_fused", False):      # rt_finish_stats wrote them (this iteration's finish_step)
            self._stats_fused = False
            return
        ld = self._last[1]
        self.stat_names = tuple(sorted(ld))
        # layout: [losses (sorted by name) | cooperative-launch failure word (when the model can raise one) | gradient norm]
        fw = [self.fail_word.reshape(()).float()] if self.fail_word is not None else []
        self.stats = torch.stack([ld[k].reshape(()).float() for k in self.stat_names] + fw + [self.grad_norm.reshape(()).float()])

    def queue_stats(self, stats=None):
        """Enqueues the device -> host copy of this iteration's stats vector behind the replay, into one of TWO pinned buffers used
        in turn (each with its event): a caller may launch iteration i + 1 before it reads iteration i, so the two
        copies in flight must not share a buffer.  Returns (host buffer, event)."""
        stats = self.stats if stats is None else stats
        if getattr(self, "_stats_host", None) is None:
            self._stats_host = [torch.empty(stats.numel(), dtype=torch.float32).pin_memory() for _ in range(2)]
            self._stats_event = [torch.cuda.Event(), torch.cuda.Event()]
            self._stats_flip = 0
        i = self._stats_flip
        self._stats_flip ^= 1
        self._stats_host[i].copy_(stats, non_blocking=True)
        self._stats_event[i].record()
        return self._stats_host[i], self._stats_event[i]

    @staticmethod
    def _run(hooks):
        for h in hooks:
            h()

    def _opt(self):
        self.grad_norm = self.optimizer.clip_grad_norm_(self.max_norm)
        self.optimizer.step()
        self._pack_stats()

    @property
    def batch(self):
        """The static input buffers the graphs read: (samples, targets).  An input pipeline that writes the next batch INTO
        them (reftr_amd.data's device kernels take an output tensor) and calls `step(*step.batch)` pays no staging copy;
        any other batch of the captured shape is copied in (one small copy per field) before the replay."""
        return self.s, self.t

    def stage(self, samples, targets):
        """Copies the NEXT batch into the static input buffers now -- stream-ordered behind the replay that is still running,
        while the host would otherwise idle in front of the iteration's device -> host copy -- so that the next call only has to
        launch the graph.  Returns False (and does nothing) for a batch of another shape."""
        if samples is None or self.shape_key(samples, targets) != self.key:
            return False
        _copy_batch(self.s, self.t, samples, targets)
        self._staged = (samples, targets)
        return True

    def _stage_in(self, samples, targets):
        staged, self._staged = self._staged, None
        if staged is None or staged[0] is not samples or staged[1] is not targets:
            _copy_batch(self.s, self.t, samples, targets)

    def __call__(self, samples, targets, nb_reduced=None):
        staged = self._staged is not None and self._staged[0] is samples and self._staged[1] is targets
        assert staged or (samples is self.s and targets is self.t) or self.shape_key(samples, targets) == self.key, \
            "captured for another input shape; use train_step"
        if self.inner._operands_dirty:        # the masters were changed outside an optimizer step (load_state_dict, a restored
            self.inner.refresh_now()          # snapshot): the graph no longer carries an operand refresh of its own
        if hasattr(self.optimizer, "check_sparse_state"):
            self.optimizer.check_sparse_state()      # moments restored from outside: the sparse-state bytes go back to 'unknown'
        if self.deferred:
            self._stage_in(samples, targets)
            self.g_fb.replay()                # applies iteration i-1's update with the rates synced at iteration i-1
            lrs = [g["lr"] for g in self.optimizer.param_groups]
            if lrs != self._lrs:              # this iteration's rates, for the update the NEXT replay (or flush) applies
                self._lrs = lrs
                self.optimizer.sync_lr()
            self.optimizer.step_count += 1
            self._pending = True
            self._set_flush(True)
            return self.out[0], self.out[1], self.grad_norm
        lrs = [g["lr"] for g in self.optimizer.param_groups]
        if lrs != self._lrs and not self.deferred_dp:
            if getattr(self.optimizer, "lr_dev", None) is not None:
                self._lrs = lrs
                self.optimizer.sync_lr()          # stream-ordered, in front of this iteration's optimizer graph
            else:
                self.refresh_lr()
        self._stage_in(samples, targets)
        self._refresh_num_boxes(targets, reduced=nb_reduced)
        self.g_fb.replay()                # deferred: applies iteration i-1's update with the rates synced at iteration i-1
        if self.deferred_dp and lrs != self._lrs:
            self._lrs = lrs
            self.optimizer.sync_lr()      # this iteration's rates, for the update the NEXT replay (or flush) applies
        for name, g in zip(self.phases, self.g_seg):
            self._run(self._phase_hooks.get(name, ()))       # this slice is final: its all-reduce goes out now ...
            g.replay()                                       # ... and runs under the next segment of backward
        self._run(self._post)
        self.g_opt.replay()
        self.optimizer.step_count += 1
        if self.deferred_dp:
            self._pending = True
            self._set_flush(True)
        elif not getattr(self.optimizer, "_last_emitted", False):
            self.inner.mark_dirty()      # eager forwards after a replay must rebuild the bf16 operands
        return self.out[0], self.out[1], self.grad_norm

    def refresh_lr(self):
        self._lrs = [g["lr"] for g in self.optimizer.param_groups]
        self.g_opt = torch.cuda.CUDAGraph()
        sc = self.optimizer.step_count
        with _capture(self.g_opt, pool=self.g_fb.pool()):
            self._opt()
        self.optimizer.step_count = sc


def dp_capture_decision(can_replay, can_capture, device, num_boxes=None):
    """The collective choice between replaying, capturing and the eager loop body under data parallelism: 'replay' only if
    EVERY rank can replay its batch, 'capture' only if every rank would capture, else 'eager' on every rank.  ONE all-reduce per
    iteration: with `num_boxes` (this rank's box count) the same collective also carries the criterion's box-count average
    (criterion.py:176-180), which a replay would otherwise have to all-reduce on its own right behind this one -- returns
    (decision, device tensor [1] = world-average box count) then, the decision alone otherwise."""
    v = torch.tensor([0.0 if can_replay else 1.0, 0.0 if can_capture else 1.0, float(num_boxes or 0.0)], dtype=torch.float32, device=device)
    torch.distributed.all_reduce(v)                       # SUM: a rank that cannot contributes 1
    no_replay, no_capture, _ = v.tolist()
    decision = "replay" if no_replay == 0 else ("capture" if no_capture == 0 else "eager")
    if num_boxes is None:
        return decision
    return decision, v[2:3] / utils.get_world_size()


class Lookahead:
    """The next batch of a loop, fetched at most once: by the step while the device is busy with the current batch (the
    captured path calls it between the graph launch and the iteration's host sync), else by the loop itself."""

    def __init__(self, fn):
        self.fn, self.batch, self.done = fn, None, False

    def __call__(self):
        if not self.done:
            self.batch, self.done = self.fn(), True
        return self.batch


class _EagerResult:
    """finish() of a step that ran eagerly: the values are already there."""

    def __init__(self, res):
        self.res = res

    def finish(self):
        return self.res


class _ReplayInFlight:
    """A replayed step between its launch and its host read-out (`finish`)."""

    def __init__(self, cap, criterion, retry=None, slot=None):
        self.cap, self.criterion, self.retry = cap, criterion, retry
        self.slot = slot if slot is not None else (cap._stats_host[0], cap._stats_event[0])

    def finish(self):
        cap = self.cap
        host_buf, event = self.slot
        event.synchronize()
        host = host_buf.tolist()
        k = len(cap.stat_names)
        if cap.fail_word is not None and host[k] != 0:
            # A stage hand-off of the cooperative decoder launches timed out in this iteration (under data parallelism: on any
            # rank -- the word was summed with the losses).  Its AdamW update was never armed (optimizer.finish_step's device-side
            # veto), so no parameter has changed; the launches are switched off, the captures dropped, and the SAME batch is run
            # again on the launched chain with the dropout seeds and learning rates it had.
            if self.retry is None:
                _coop_fallback(cap.model)
                raise RuntimeError("rt_decoder_fwd: a stage hand-off timed out and the iteration cannot be re-run from here "
                                   "(replayed without begin_train_step); the launches are now off (REFTR_DEC_COOP=0)")
            return self.retry()
        weight_dict = self.criterion.weight_dict
        unscaled = {f"{n}_unscaled": v for n, v in zip(cap.stat_names, host)}
        scaled = {n: v * weight_dict[n] for n, v in zip(cap.stat_names, host) if n in weight_dict}
        loss_value = sum(scaled.values())
        if not math.isfinite(loss_value):
            print("Loss is {}, stopping training".format(loss_value))
            print(unscaled)
            sys.exit(1)
        return loss_value, scaled, unscaled, host[-1]


def begin_train_step(model, criterion, samples, targets, optimizer, lr_scheduler=None, max_norm=0.0, max_shapes=4, lookahead=None):
    """Launches one iteration of the loop body and returns a handle; `handle.finish()` waits for it and returns what
    `train_step` returns.  Between the two the caller may do host work (the epoch loop books the previous iteration's meters).

    Replayed from hipGraphs: the first batch of an input shape captures a CapturedTrainStep (kept on the model), later batches
    of that shape replay it; CPU tensors, foreign optimizers and more than `max_shapes` different shapes (variable-size data)
    run the eager `train_step`.  `lookahead` (a Lookahead) is called while the replay runs: the next batch is fetched and staged
    into the graph's input buffers under the current step instead of in front of the next one.

    Everything the host does between an iteration's read-out and the next launch is device idle time, so the steady state is
    short: a batch that the previous iteration already staged is recognised by identity (no shape key, no lookups), the
    scheduler step and the stats copy are issued right behind the launch."""
    inner = getattr(model, "module", model)
    caps = inner.__dict__.setdefault("_captured_steps", {})
    dist_on = utils.is_dist_avail_and_initialized() and utils.get_world_size() > 1
    nb_reduced = None
    cap = inner.__dict__.get("_staged_cap")
    if cap is not None and not dist_on and cap._staged is not None and cap._staged[0] is samples and cap._staged[1] is targets \
            and cap.criterion is criterion and cap.optimizer is optimizer and cap.max_norm == max_norm and cap.training == model.training:
        pass                                                   # steady state: staged by the previous iteration's lookahead
    else:
        cap = None
        img = samples.get("img")
        ok = (isinstance(img, utils.NestedTensor) and img.tensors.is_cuda and hasattr(optimizer, "clip_grad_norm_")
              and os.environ.get("REFTR_TRAIN_GRAPH", "1") == "1")
        key = None
        if ok:
            key = (CapturedTrainStep.shape_key(samples, targets), id(criterion), id(optimizer), float(max_norm), model.training)
            ok = key in caps or len(caps) < max_shapes
        if dist_on:
            # Data parallel: replaying, capturing and the eager loop issue DIFFERENT collective sequences (a capture adds the
            # num_boxes all-reduce of its constructor and warm-up iterations with real gradient exchanges), and both the shape
            # key (per-image box counts, image sizes) and the capture budget are rank-local.  The choice is therefore made
            # collectively: replay only if EVERY rank holds a capture for its batch, capture only if every rank would capture,
            # otherwise every rank runs the eager step.  One 2-word MIN all-reduce per iteration, issued while the device is
            # idle behind the previous iteration's read-out.
            mine = ok
            decision, nb_reduced = dp_capture_decision(ok and key in caps, ok and key not in caps, inner.store.device,
                                                       num_boxes=sum(len(t["labels"]) for t in targets))
            ok = decision != "eager"
            if decision != "replay":
                nb_reduced = None                   # the capture / eager paths reduce the box count themselves
            if mine and not ok:
                # this rank could have replayed / captured, another one could not (its capture budget is spent, or its batch has a
                # shape this rank already holds while the other must still capture): the whole job runs this iteration eagerly.
                # Said once per shape -- a run that quietly degrades to eager launches is 2-3x slower.
                seen = inner.__dict__.setdefault("_dp_eager_logged", set())
                if key not in seen:
                    seen.add(key)
                    print(f"[reftr_amd] rank {utils.get_rank()}: data-parallel step runs eagerly (ranks disagree on replay/capture for "
                          f"this input shape; {len(caps)} of {max_shapes} captures in use)", file=sys.stderr)
        if not ok:
            for other in caps.values():             # a pending (deferred) update must land before the eager forward
                other.flush()
            return _EagerResult(train_step(model, criterion, samples, targets, optimizer, lr_scheduler, max_norm))
        cap = caps.get(key)
        if cap is None:
            for other in caps.values():             # one pending (deferred) update at a time
                other.flush()
            # the capture's warm-up iterations are real updates on this batch: take them back, so that the loop sees exactly one
            # update per batch, like the eager loop
            st = inner.store
            snap = (st.flat_p.clone(), optimizer.m.clone(), optimizer.v.clone(), optimizer.step_dev.clone(), optimizer.step_count,
                    inner.seed_dev.clone(), inner._step)
            cap = caps[key] = CapturedTrainStep(model, criterion, optimizer, max_norm, samples, targets)
            cap.training = model.training
            cap.reset_pending()
            st.flat_p.copy_(snap[0]); optimizer.m.copy_(snap[1]); optimizer.v.copy_(snap[2]); optimizer.step_dev.copy_(snap[3])
            optimizer.step_count = snap[4]
            inner.seed_dev.copy_(snap[5]); inner._step = snap[6]
            inner.mark_dirty(full=True)      # the replay's first act finds the operands dirty and rebuilds them from the restored masters
            del snap
        else:
            for other in caps.values():
                if other is not cap:
                    other.flush()
    lrs_now = [g["lr"] for g in optimizer.param_groups]
    seed_back = (1 if model.training else 0, 1)
    cap(samples, targets, nb_reduced)
    # ONE device -> host copy for everything the loop looks at (losses for the meters and the finite check, gradient norm);
    # under data parallelism the loss entries are first averaged over the ranks in one all-reduce (util/misc.py:136-160).
    # The copy is enqueued right behind the replay, into pinned memory; the host then spends the step's run time on the
    # scheduler, on fetching and staging the NEXT batch (H2D hand-over, copies into the graph's input buffers -- stream-ordered
    # BEHIND the stats copy) and only then waits for the copy's event.
    stats = cap.stats
    if dist_on:
        k = len(cap.stat_names)
        kf = k + (1 if cap.fail_word is not None else 0)      # the failure word rides along: SUM over ranks, non-zero = some rank failed
        stats = stats.clone()
        torch.distributed.all_reduce(stats[:kf])
        stats[:k] /= utils.get_world_size()
    slot = cap.queue_stats(stats)
    if lr_scheduler is not None:
        lr_scheduler.step()          # host state only; the device reads the new rates when the next launch syncs them
    inner.__dict__["_staged_cap"] = None
    if lookahead is not None:
        nxt = lookahead()
        if nxt is not None and nxt[0] is not None and cap.stage(*nxt):
            inner.__dict__["_staged_cap"] = cap
    def _retry():
        # what the failed iteration advanced: host step count (the device counter was vetoed), dropout seed / step, learning rates
        _coop_fallback(model)
        optimizer.step_count -= 1
        # The veto is a per-rank device word, the decision to run the iteration again is collective: a healthy rank's finish_step
        # has advanced its device counter, the failing rank's was vetoed.  Every rank re-bases the device counter on the host count
        # (the same on all ranks), otherwise the AdamW bias corrections of the replicas differ from here on.
        optimizer.step_dev.fill_(optimizer.step_count)
        _restore_seed(model, seed_back)
        after = [g["lr"] for g in optimizer.param_groups]
        for g, lr in zip(optimizer.param_groups, lrs_now):
            g["lr"] = lr
        try:
            return begin_train_step(model, criterion, samples, targets, optimizer, None, max_norm, max_shapes, None).finish()
        finally:
            for g, lr in zip(optimizer.param_groups, after):
                g["lr"] = lr
    return _ReplayInFlight(cap, criterion, retry=_retry, slot=slot)


def captured_train_step(model, criterion, samples, targets, optimizer, lr_scheduler=None, max_norm=0.0, max_shapes=4, lookahead=None):
    """`train_step` with the same return value, on `begin_train_step` (launch) + `finish` (read-out)."""
    return begin_train_step(model, criterion, samples, targets, optimizer, lr_scheduler, max_norm, max_shapes, lookahead).finish()


def _check_cooperative(model):
    """rt_decoder_fwd reports a consumer that gave up waiting (a workgroup that never became resident) through a device word;
    a loop that produced numbers from such a launch must not return them."""
    inner = getattr(model, "module", model)
    dc = getattr(getattr(inner, "net", None), "dec_counters", None)
    if dc is not None and int(dc[-1]) != 0:
        raise RuntimeError("rt_decoder_fwd: a stage hand-off timed out (workgroups not co-resident?) -- set REFTR_DEC_COOP=0")


def train_one_epoch(model, criterion, data_loader, optimizer, lr_scheduler, device, epoch, max_norm=0):
    model.train()
    criterion.train()
    board = utils.StatBoard()
    header = "Epoch: [{}]".format(epoch)
    prefetcher = data_prefetcher(data_loader, device, prefetch=True)
    samples, targets = prefetcher.next()
    booked = None
    # The loop reads iteration i before it launches iteration i + 1 (engine_vg.py:53-58 stops on a non-finite loss BEFORE the update).
    # Launching i + 1 first was built in round 5 on the device-side veto that is still there (rt_finish_step never arms the update of
    # an iteration whose total is not finite) and measured SLOWER: a hipGraph launched while its previous launch is still running
    # starts later than one launched from an idle stream on this stack (+0.03 ... 0.11 ms per iteration, also with two instantiated
    # graphs used in turn; profiles/r05_pipeline_negative_result.txt).  Removed.

    def _read(step):
        loss_value, scaled, unscaled, gnorm = step.finish()
        if torch.is_tensor(gnorm):               # the eager step hands back the optimizer's device scalar, which the next step overwrites
            gnorm = float(gnorm)
        return dict(loss=loss_value, **scaled, **unscaled, lr=step.lr_logged, grad_norm=gnorm)

    for _ in board.log_every(range(len(data_loader)), 50, header):
        # the loop body of engine_vg.py:40-72 -- replayed from hipGraphs for fixed-shape data (RefCOCO: 640 x 640, L = 40);
        # the replayed path hands back host numbers (one stacked copy per iteration), the eager one device scalars.
        ahead = Lookahead(prefetcher.next)
        step = begin_train_step(model, criterion, samples, targets, optimizer, lr_scheduler, max_norm, lookahead=ahead)
        step.lr_logged = optimizer.param_groups[0]["lr"]      # what the reference's meter shows: the rate after this iteration's scheduler step
        if booked is not None:
            board.add(**booked)
            booked = None
        booked = _read(step)
        samples, targets = ahead()
    if booked is not None:
        board.add(**booked)
    _check_cooperative(model)
    board.synchronize_between_processes()
    print("Averaged stats:", board)
    return board.global_avg()


class CapturedForward:
    """Inference forward replayed from one hipGraph per input shape (RefCOCO-style evaluation pads every image to the same
    square, datasets/transforms.py: NormalizeAndPad): ~600 launches become one replay.  The forward has no atomics, so the
    replayed outputs are bit-identical to the eager ones (tests/test_model_gpu.py).  Outputs are static buffers that the
    next replay overwrites."""

    def __init__(self, model, max_shapes=4):
        self.model, self.graphs, self.max_shapes = model, {}, max_shapes
        self._gen = _inner(model).__dict__.get("_coop_generation", 0)

    @staticmethod
    def shape_key(samples):
        return tuple((n, tuple(v.tensors.shape) if isinstance(v, utils.NestedTensor) else tuple(v.shape))
                     for n, v in sorted(samples.items()) if isinstance(v, utils.NestedTensor) or torch.is_tensor(v))

    @torch.no_grad()
    def __call__(self, samples):
        if not isinstance(samples.get("img"), utils.NestedTensor):
            return self.model(samples)
        gen = _inner(self.model).__dict__.get("_coop_generation", 0)
        if gen != self._gen:                     # the cooperative launches were switched off: the graphs that contain them go
            self.graphs, self._gen = {}, gen
        key = self.shape_key(samples)
        ent = self.graphs.get(key)
        if ent is None and len(self.graphs) >= self.max_shapes:      # variable-size data: eager launches
            return self.model(samples)
        if ent is None:
            s, _ = _clone_batch(samples, [])
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                self.model(s)                        # warm-up: operand refresh, workspaces, descriptor tables
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with _capture(g):
                out = self.model(s)
            ent = self.graphs[key] = (g, s, out)
        g, s, out = ent
        inner = getattr(self.model, "module", self.model)
        if inner._flush_pending is not None:
            inner._flush_pending()
        if inner._operands_dirty:                    # weights changed since the capture: rebuild the bf16 operands eagerly
            inner.refresh_operands()
            if inner._lin_refresh_pending:
                inner.net.refresh(); inner._lin_refresh_pending = False
        _copy_batch(s, [], samples, [])
        g.replay()
        return out


def _vis_dirs(output_dir, split):
    """output_dir/vis/<split>/{mask,bbox,att,gt} (engine_vg.py:86-94)."""
    from pathlib import Path
    if output_dir is None:
        raise ValueError("evaluate(visualize=True) needs an output_dir")
    root = Path(output_dir) / "vis" / str(split)
    for sub in ("mask", "bbox", "att", "gt"):
        (root / sub).mkdir(parents=True, exist_ok=True)
    return root


_VIS_PURPLE = (128, 0, 128)
_VIS_YELLOW = (255, 255, 0)


def _dump_visuals(root, dataset, dataset_id, mask_origin, pred_box, mask_att):
    """The four image dumps of one evaluated sample (engine_vg.py:157-192): `mask_origin` uint8 / bool [H, W] at the original
    image size (PostProcessSegm's 'masks_origin'), `pred_box` xyxy in original pixels, `mask_att` [heads, h, w] (the
    segmentation head's attention maps).  Host-side, off the hot path; PIL / matplotlib are imported here only."""
    import numpy as np
    from PIL import Image, ImageDraw
    import matplotlib
    matplotlib.use("Agg", force=False)
    import matplotlib.pyplot as plt
    img, mask, _phrase, tgt_box, img_file = dataset.pull_item(dataset_id)
    pm = (mask_origin.detach().cpu().numpy() != 0)
    mask = np.asarray(mask)
    assert pm.shape == mask.shape[:2], (pm.shape, mask.shape)
    stem = str(img_file).split("/")[-1].split(".")[0]
    tag = f"{stem}_{dataset_id:05d}"
    purple = np.array(_VIS_PURPLE, dtype=np.uint8); yellow = np.array(_VIS_YELLOW, dtype=np.uint8)
    Image.fromarray(np.where(pm[..., None], yellow, purple)).save(root / "mask" / f"{tag}.jpg")
    Image.fromarray(np.where((mask != 0)[..., None], yellow, purple)).save(root / "gt" / f"{tag}.jpg")
    canvas = Image.fromarray(np.asarray(img))
    draw = ImageDraw.Draw(canvas)
    draw.rectangle([float(v) for v in pred_box.detach().cpu().reshape(-1)], outline="blue", width=5)
    draw.rectangle([float(v) for v in np.asarray(tgt_box).reshape(-1)], outline="red", width=5)
    canvas.save(root / "bbox" / f"{tag}.jpg")
    # attention maps: resized to 320 x 320 and cropped to the image's half size, heads 0, 1, 2 and 7 (:181-186)
    att = torch.nn.functional.interpolate(mask_att.detach().float().cpu()[None], size=(320, 320), mode="bilinear")[0].numpy()
    h, w = mask.shape[:2]
    for head in (0, 1, 2, 7):
        if head < att.shape[0]:
            plt.imsave(root / "att" / f"{tag}_{head}.jpg", att[head, :h // 2, :w // 2], cmap="viridis")


@torch.no_grad()
def evaluate(model, criterion, postprocessors, data_loader, device, output_dir=None, visualize=False):
    """engine_vg.evaluate (engine_vg.py:82-225): losses, Acc@0.5 / mean IoU of the predicted boxes (:127-140), mask IoU when a
    'segm' post-processor is present (:143-152), boxes scaled to the original image size in the returned results dict (:141,203).
    `visualize` (needs the 'segm' post-processor, `output_dir` and a dataset with `split` / `pull_item`, as in the reference):
    per sample the predicted mask, the ground-truth mask, the image with both boxes and four attention maps under
    output_dir/vis/<split>/{mask,gt,bbox,att} (:86-96,157-192)."""
    from .util.box_ops import mask_iou
    model.eval()
    criterion.eval()
    board = utils.StatBoard()
    sum_accu = torch.zeros((), device=device); sum_iou = torch.zeros((), device=device); cnt = torch.zeros((), device=device)
    seg_iou = torch.zeros((), device=device); cnt_seg = 0.0
    results_dict = {}
    vis_dir = _vis_dirs(output_dir, data_loader.dataset.split) if visualize else None
    prefetcher = data_prefetcher(data_loader, device, prefetch=True)
    samples, targets = prefetcher.next()
    # the forward is replayed from a hipGraph per input shape (fixed-size evaluation sets; bit-identical outputs; up to four
    # shapes, then eager launches); REFTR_EVAL_GRAPH=0: always eager
    fwd = CapturedForward(model) if (os.environ.get("REFTR_EVAL_GRAPH", "1") == "1" and torch.device(device).type == "cuda") else model
    wvec = None
    for _ in board.log_every(range(len(data_loader)), 50, "Test:"):
        outputs = fwd(samples)
        if _cooperative_failed(model):           # a hand-off timed out: the batch is run again on the launched chain
            _coop_fallback(model)
            outputs = fwd(samples)
        loss_dict = criterion(outputs, targets)
        weight_dict = criterion.weight_dict
        red = utils.reduce_dict(loss_dict)
        # meters stay on the device (no .item() per loss): [unscaled..., scaled..., total] of this iteration in one vector
        names = sorted(red)
        wn = [k for k in names if k in weight_dict]
        if wvec is None or wvec[0] != names:
            wvec = (names, torch.tensor([weight_dict[k] for k in wn], dtype=torch.float32, device=red[names[0]].device))
        un = torch.stack([red[k].reshape(()).float() for k in names])
        sc = torch.stack([red[k].reshape(()).float() for k in wn]) * wvec[1]
        board.add_device([f"{k}_unscaled" for k in names] + wn + ["loss"], torch.cat([un, sc, sc.sum().reshape(1)]))
        key = "orig_size" if "orig_size" in targets[0] else "size"
        orig_sizes = torch.stack([t[key] for t in targets], dim=0)
        results = postprocessors["bbox"](outputs, orig_sizes)
        for res, tg in zip(results, targets):
            gt = box_cxcywh_to_xyxy(tg["boxes"])
            assert gt.size(0) == res["boxes"].size(0), (res, gt)
            iou = torch.diag(box_iou(gt, res["boxes"])[0])
            sum_accu += (iou > 0.5).float().sum(); sum_iou += iou.sum(); cnt += len(tg["boxes"])
        results_scaled = postprocessors["bbox"](outputs, orig_sizes, scale_to_original_shape=True)
        if "segm" in postprocessors:
            target_sizes = torch.stack([t["size"] for t in targets], dim=0)
            results = postprocessors["segm"](results, outputs, orig_sizes, target_sizes)
            for i, (res, tg) in enumerate(zip(results, targets)):
                seg_iou += mask_iou(res["masks"][0][0], tg["masks"])
                cnt_seg += 1
                if vis_dir is not None:
                    _dump_visuals(vis_dir, data_loader.dataset, int(tg["dataset_id"]), res["masks_origin"][0, 0],
                                  results_scaled[i]["boxes"][0], outputs["mask_att"][i])
        for tg, res in zip(targets, results_scaled):
            if "image_id" in tg:
                results_dict[int(tg["image_id"])] = res["boxes"].cpu().numpy().tolist()
        samples, targets = prefetcher.next()
    _check_cooperative(model)
    board.synchronize_between_processes()
    stats = board.global_avg()
    if utils.is_dist_avail_and_initialized():
        for t in (sum_accu, sum_iou, cnt):
            torch.distributed.all_reduce(t)
    stats["accuracy_iou0.5"] = float(sum_accu / cnt.clamp(min=1))
    stats["miou"] = float(sum_iou / cnt.clamp(min=1))
    if "segm" in postprocessors:
        if utils.is_dist_avail_and_initialized():
            torch.distributed.all_reduce(seg_iou)
            cnt_seg = utils.get_world_size() * cnt_seg
        stats["seg_miou"] = float(seg_iou / max(cnt_seg, 1.0))
    stats = {k: v for k, v in stats.items() if k.split("_")[-1] not in ("unscaled", "0", "1", "2")}      # engine_vg.py:221
    return stats, results_dict
