"""FusedAdamW: torch.optim.AdamW + clip_grad_norm_ of the reference loop (main_vg.py:234-268,
engine_vg.py:62-66) as two kernels over the model's flat parameter / gradient buffers.

It is a torch.optim.Optimizer so `lr_scheduler.step()` (StepLR / LambdaLR, main_vg.py:269-287) and
`optimizer.param_groups[0]["lr"]` (engine_vg.py:71) keep working; the four param groups are the
reference's (default lr / lr_backbone names 'img_backbone.0' / lr_bert names 'lang_backbone', the latter
also at args.lr_backbone — main_vg.py:251-255).
"""
import torch

from . import hip as H
from .models import layout as L


def cover_span(mats, b, e):
    """`mats`: (element offset, N, T, C) of weight matrices inside [b, e), sorted by offset.  Returns (jobs, total tiles, chunks):
    jobs = (offset, N, T, C, first tile) with ceil(N/32) * ceil(T*C/256) tiles each (rt_adamw_mat's tiling); chunks = flat [offset, count, ...] pairs
    (count <= 16384) covering every element of [b, e) that no matrix covers.  Together they tile [b, e) exactly once."""
    jobs, tiles, pos, chunks = [], 0, b, []

    def fill(a, to):
        while a < to:
            c = min(16384, to - a)
            chunks.extend((a, c)); a += c
    for off, N, T, C in mats:
        assert off >= pos and off + N * T * C <= e and off % 4 == 0 and (N * T * C) % 4 == 0, \
            "weight matrices overlap, cross the span's end or are not 16-byte aligned"
        fill(pos, off)
        jobs.append((off, N, T, C, tiles))
        tiles += ((N + 31) // 32) * ((T * C + 255) // 256)
        pos = off + N * T * C
    fill(pos, e)
    return jobs, tiles, chunks


class FusedAdamW(torch.optim.Optimizer):
    SGD = False

    def __init__(self, model, lr=1e-4, lr_backbone=1e-5, lr_bert=None, weight_decay=1e-4, betas=(0.9, 0.999), eps=1e-8,
                 lr_mask_branch_proj=1.0):
        self.model = getattr(model, "module", model)
        st = self.model.store
        named = dict(self.model.named_parameters())
        groups = []
        # the reference's four param groups, in its order, each in its named_parameters() order (main_vg.py:234-262)
        order = L.reference_param_order(self.model.cfg)
        assert sorted(order) == sorted(n for n, _, k in st.table if k == "param")
        self._names = []
        for grp, glr in ((L.GROUP_MAIN, lr), (L.GROUP_BACKBONE, lr_backbone),
                         (L.GROUP_BERT, lr_backbone if lr_bert is None else lr_bert),
                         (L.GROUP_MASK, lr * lr_mask_branch_proj)):
            ns = [n for n in order if L.lr_group(n) == grp]
            self._names.append(ns)
            groups.append({"params": [named[n] for n in ns], "lr": glr, "group_id": grp})
        super().__init__(groups, dict(lr=lr, weight_decay=weight_decay, betas=betas, eps=eps))
        dev = st.device
        self.m = torch.zeros_like(st.flat_p)
        self.v = torch.zeros_like(st.flat_p)
        self.sq = torch.zeros(1, dtype=torch.float32, device=dev)
        self.grad_norm = torch.zeros(1, dtype=torch.float32, device=dev)
        self.step_count = 0
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=dev)    # device-resident step counter (graph replay)
        self._max_norm = 0.0
        self._have_sq = False
        # deferred mode (engine_vg.CapturedTrainStep): the update of step i is applied at the head of step i+1
        self.active = None            # device word: != 0 while an update is pending
        self.lr_dev = None            # device learning rates (one per non-empty param group) read by the kernel
        self._flush_pending = None    # set by the engine: applies a pending update before anyone reads weights / state
        # round 4: the AdamW pass walks the weight matrices in tiles and writes their bf16 GEMM operands itself (rt_adamw_mat +
        # rt_adamw_chunks) -- no separate operand refresh after an update.  REFTR_OPT_EMIT=0: the flat pass + rt_weight_prep_batched.
        import os
        self._emit = os.environ.get("REFTR_OPT_EMIT", "1") != "0" and not self.SGD
        self._emit_cache = {}
        self._last_emitted = False
        # sparse optimizer state (rt_adamw_mat, matrices without bf16 operands = the embedding tables): one byte per KB row piece,
        # 0 = "m and v of the piece are zero"; such a piece with an all-zero gradient is skipped without reading p / m / v (exact: see
        # the kernel).  The bytes describe m / v as THIS optimizer's kernels left them: any write to them from outside (a restored
        # snapshot, load_state_dict) is noticed through the tensors' version counters and resets every byte to "unknown" (1).
        self._sparse = True
        self._sparse_flags = {}
        self._mv_version = None
        # device word that vetoes an iteration's update (the cooperative decoder's failure word): see finish_step / step
        self.veto = None

    def zero_grad(self, set_to_none=False, fast=False):
        """One memset of the flat gradient buffer (param.grad views are kept).  fast=True (the training loops of
        reftr_amd.engine_vg, right in front of backward): only the atomically-accumulated tensors are cleared and the weight
        matrices -- 96 % of the 607 MB -- are overwritten by their first weight-gradient launch (ParamStore.arm_overwrite);
        REFTR_OVERWRITE=0 keeps the full clear."""
        self.model.store.zero_for_backward(fast)

    def _grad_buffer(self):
        """The buffer the update reads: the fp32 gradients, or -- in a data-parallel run that exchanges bf16 -- the bf16 copy
        the all-reduce produced (reftr_amd.parallel.DistributedDataParallel sets store.flat_g16)."""
        g16 = getattr(self.model.store, "flat_g16", None)
        return g16 if g16 is not None else self.model.store.flat_g

    def _sqnorm_all(self):
        """Squared norm of the whole gradient buffer into self.sq.  When the backward already reduced the BERT slice on the
        language stream (model._norm_split, single-process fp32 gradients only), only the rest is read here."""
        g = self._grad_buffer()
        split = getattr(self.model, "_norm_split", None)
        st = self.model.store
        if getattr(st, "norm_valid", False) and g is st.flat_g and not getattr(self.model, "dp_mode", False):
            # the weight-gradient launches of this backward have left |dw|^2 of every registered matrix in the accumulator slots
            # (hip._SQACC_MAP); what they do not produce -- biases, norm parameters, embeddings: 4 % of the buffer -- is read here
            table, n = st._complement_table()
            H.sqnorm_finish(st.flat_g, table, n, st.sq_slots, self.sq)
            st.norm_valid = False
            self.model._norm_split = None
        elif split is not None and g is st.flat_g and split[1] == st.flat_g.numel() and split[0] > 0:
            H.sqnorm(g[:split[0]], self.sq)
            self.sq.add_(split[2])
            self.model._norm_split = None
        else:
            H.sqnorm(g, self.sq)

    def clip_grad_norm_(self, max_norm):
        """Launches the global-norm reduction; the clip coefficient itself is applied inside the AdamW kernel.
        Returns the device scalar that holds the total norm after step()."""
        self._sqnorm_all()
        self._max_norm, self._have_sq = float(max_norm), True
        return self.grad_norm

    def _ranges(self):
        st = self.model.store
        ranges = []
        for g in self.param_groups:
            b, e = st.group_range[g["group_id"]]
            if e > b:
                ranges.append((b, e, g["lr"], g["weight_decay"]))
        return ranges

    def _emit_tables(self, span):
        """((job table, njobs, tiles), (chunk table, nchunks)) covering `span` of the flat buffers exactly once: the model's weight
        matrices as tile jobs (with the bf16 operand tensors they feed), everything else in <= 16384-element chunks.  None while
        the model has not built its operands yet (before the first forward) or is not on a GPU."""
        model = self.model
        st = model.store
        jobs = model.operand_jobs() if hasattr(model, "operand_jobs") else None
        if not jobs or not st.flat_p.is_cuda:
            return None
        key = (span, getattr(model, "_operand_version", 0))
        ent = self._emit_cache.get(key)
        if ent is None:
            b, e = span if span is not None else (0, st.flat_p.numel())
            mine = sorted((j for j in jobs if b <= j[0] < e), key=lambda j: j[0])
            geo, tiles, chunks = cover_span([j[:4] for j in mine], b, e)
            rows = [[off, H._p(j[4]) or 0, H._p(j[5]) or 0, H._p(j[6]) or 0, N, T, C, first] for (off, N, T, C, first), j in zip(geo, mine)]
            if self._sparse:
                for row, j in zip(rows, mine):
                    if j[4] is None and j[5] is None and j[6] is None:          # no operands: the scale slot carries the state bytes
                        row[1] = self._sparse_flag(row[0], row[4], row[5] * row[6]).data_ptr()
            dev = st.device
            mat = (torch.tensor(rows, dtype=torch.int64).to(dev) if rows else None, len(rows), tiles)
            chk = (torch.tensor(chunks, dtype=torch.int64).to(dev) if chunks else None, len(chunks) // 2)
            ent = self._emit_cache[key] = (mat, chk, [j[4:] for j in mine])      # the tensors behind the raw pointers stay referenced
        return ent[0], ent[1]

    def _sparse_flag(self, off, N, K):
        f = self._sparse_flags.get(off)
        if f is None:
            f = self._sparse_flags[off] = torch.ones(N * ((K + 255) // 256), dtype=torch.uint8, device=self.model.store.device)
        return f

    def check_sparse_state(self):
        """Host-side, before a launch / a graph replay: m or v written from outside since the last look -> every state byte back to
        'unknown'.  (The kernels write through raw pointers and do not move the version counters.)"""
        if not self._sparse_flags:
            return
        ver = (self.m._version, self.v._version)
        if ver != self._mv_version:
            if not torch.cuda.is_current_stream_capturing():
                for f in self._sparse_flags.values():
                    f.fill_(1)
                self._mv_version = ver

    def _launch(self, span=None):
        """One AdamW pass over `span` (default: everything).  Returns True when the pass also wrote the bf16 operands of the
        matrices it updated (the caller then skips the operand refresh)."""
        st = self.model.store
        b1, b2 = self.defaults["betas"]
        tabs = self._emit_tables(span) if self._emit else None
        kw = dict(mat=tabs[0], chunks=tabs[1]) if tabs is not None else {}
        self.check_sparse_state()
        H.adamw_flat(st.flat_p, st.flat_g, self.m, self.v, step=max(self.step_count, 1), ranges=self._ranges(), gnorm_sq=self.sq,
                     g16=getattr(st, "flat_g16", None),
                     gnorm_out=self.grad_norm, grad_scale=getattr(self.model, "_grad_scale", 1.0),
                     max_norm=self._max_norm, beta1=b1, beta2=b2, eps=self.defaults["eps"], step_dev=self.step_dev,
                     active=self.active, lr_dev=self.lr_dev, span=span, sgd=self.SGD, **kw)
        self._last_emitted = tabs is not None
        if tabs is None and self._sparse_flags and not torch.cuda.is_current_stream_capturing():
            # the flat pass wrote m / v of the embedding tables through raw pointers: their "m and v are zero here" bytes are void
            for f in self._sparse_flags.values():
                f.fill_(1)
        return self._last_emitted

    @torch.no_grad()
    def step(self, closure=None):
        if self._flush_pending is not None:
            self._flush_pending()
        st = self.model.store
        self.step_count += 1
        H.counter_add(self.step_dev, 1, unless=self.veto)      # a vetoed iteration does not advance the device counter either
        if not self._have_sq:
            H.sqnorm(self._grad_buffer(), self.sq)
        act, self.active = self.active, None          # an immediate step is never conditional ...
        if self.veto is not None:                     # ... except on the cooperative decoder's failure word (see finish_step)
            if getattr(self, "_ok", None) is None:
                self._ok = torch.zeros(1, dtype=torch.int32, device=st.device)
            self._ok.zero_()
            H.counter_add(self._ok, 1, unless=self.veto)
            self.active = self._ok
        try:
            emitted = self._launch()
        finally:
            self.active = act
        self._have_sq = False
        if emitted:
            self.model.operands_emitted()
        else:
            self.model.mark_dirty()

    # ---- deferred mode: [finish_step at the end of iteration i] ... [apply_pending at the head of iteration i+1] ----
    def enable_device_lr(self):
        """The kernel reads the learning rates from device words (kept current with `sync_lr`): a captured optimizer
        launch follows the schedule without re-capture."""
        if self.lr_dev is None:
            self.lr_dev = torch.zeros(8, dtype=torch.float32, device=self.model.store.device)
            self.sync_lr()

    def enable_deferred(self):
        self.enable_device_lr()
        if self.active is None:
            self.active = torch.zeros(1, dtype=torch.int32, device=self.model.store.device)

    def sync_lr(self):
        """Copies the param groups' current learning rates to the device words the kernel reads (stream-ordered: a
        replay enqueued before this call still sees the old values)."""
        lrs = [r[2] for r in self._ranges()]
        if lrs == getattr(self, "_lr_last", None):
            return lrs                # nothing moved since the last copy (StepLR between drops): no traffic at all
        # two persistent pinned staging buffers, alternated; an event per buffer says its last copy has run, so the
        # per-iteration schedules (LambdaLR warm-up / cosine, main_vg.py:272-287) cost no host allocation
        if getattr(self, "_lr_stage", None) is None:
            self._lr_stage = [torch.zeros(8, dtype=torch.float32).pin_memory() for _ in range(2)]
            self._lr_event = [torch.cuda.Event(), torch.cuda.Event()]
            self._lr_used = [False, False]
            self._lr_flip = 0
        i = self._lr_flip
        self._lr_flip ^= 1
        if self._lr_used[i]:
            self._lr_event[i].synchronize()
        host = self._lr_stage[i]
        host.zero_()
        host[:len(lrs)] = torch.tensor(lrs, dtype=torch.float32)
        self.lr_dev.copy_(host, non_blocking=True)
        self._lr_event[i].record()
        self._lr_used[i] = True
        self._lr_last = list(lrs)
        return lrs

    @torch.no_grad()
    def finish_step(self, max_norm, loss=None, stats=None):
        """End of an iteration in deferred mode: total gradient norm, step counter, 'update pending' flag.  The weights
        are NOT touched; grad_norm holds this iteration's (pre-clip) norm like clip_grad_norm_'s return value.
        `loss` (device fp32 scalar, optional): the iteration's weighted total -- a non-finite value vetoes the update like the
        cooperative decoder's failure word does (engine_vg.py:53-58: the reference stops BEFORE the update)."""
        self._sqnorm_all()
        self._max_norm = float(max_norm)
        # `veto` (the cooperative decoder's failure word): an iteration whose launches reported a hand-off timeout neither advances
        # the step counter nor arms its update -- decided on the device, so a replayed graph can never apply such an update; the
        # same launch checks the loss.  `active` != 0 already (earlier iterations): a veto clears it
        # a device tensor of another dtype is converted (it used to be dropped silently, and with it the on-device
        # "non-finite loss never arms the update" check); a host number cannot be checked on the device -- the loop's own
        # math.isfinite exit (engine_vg.py:53-58) covers it, as it does for the data-parallel schedules, which pass no loss
        if loss is not None:
            loss = loss.detach().float() if (torch.is_tensor(loss) and loss.is_cuda) else None
        loss1 = loss.reshape(1) if loss is not None and loss.dim() == 0 else loss
        gs = getattr(self.model, "_grad_scale", 1.0)
        if stats is not None:
            # (srcs, with_veto_word, out): the loop's stats vector [losses | failure word | norm] written by the same launch
            # (rt_finish_stats) -- the counters, the square root and the packing were four launches at the end of every replay
            srcs, with_veto, out = stats
            H.finish_stats(self.step_dev, self.active, self.veto, loss1, self.sq, gs, self.grad_norm, srcs=srcs,
                           cond_in_stats=with_veto and self.veto is not None, stats=out)
            return self.grad_norm
        H.finish_step(self.step_dev, self.active, self.veto, loss1)
        torch.sqrt(self.sq, out=self.grad_norm)
        if gs != 1.0:
            self.grad_norm.mul_(gs)
        return self.grad_norm

    @torch.no_grad()
    def apply_pending(self, span=None):
        """The AdamW pass over `span` (default: everything) for the pending update; a no-op kernel while nothing is
        pending.  May be issued as several spans on different streams.  Returns True when the pass wrote the bf16 operands too."""
        return self._launch(span)

    def clear_pending(self):
        if self.active is not None:
            self.active.zero_()

    def state_dict(self):
        """torch.optim.AdamW's format (what the reference writes into checkpoint['optimizer'], main_vg.py:377-384):
        per-parameter `step`, `exp_avg`, `exp_avg_sq` keyed by the parameter's index in the reference's group order, so a
        checkpoint written here resumes under the reference and vice versa."""
        if self._flush_pending is not None:
            self._flush_pending()
        st = self.model.store
        state, groups, idx = {}, [], 0
        for g, ns in zip(self.param_groups, self._names):
            ids = []
            for n in ns:
                if self.step_count > 0:
                    state[idx] = {"step": torch.tensor(float(self.step_count)),
                                  "exp_avg": st.view_of(self.m, n).detach().clone(memory_format=torch.contiguous_format),
                                  "exp_avg_sq": st.view_of(self.v, n).detach().clone(memory_format=torch.contiguous_format)}
                ids.append(idx); idx += 1
            pg = {k: v for k, v in g.items() if k not in ("params", "group_id")}
            pg.update(amsgrad=False, maximize=False, foreach=None, capturable=False, differentiable=False, fused=None, params=ids)
            groups.append(pg)
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        """Accepts a torch.optim.AdamW state_dict with the reference's grouping (as written by the reference or by
        state_dict() above); hyper-parameters of the groups are taken over like torch does."""
        if self._flush_pending is not None:
            self._flush_pending()
        st = self.model.store
        flat = [n for ns in self._names for n in ns]
        groups = sd["param_groups"]
        # the reference leaves its 4th (mask branch) group empty for REC models; older checkpoints may have 3 groups
        assert sum(len(g["params"]) for g in groups) == len(flat), "optimizer state does not match this model's parameters"
        steps = set()
        with torch.no_grad():
            for i, s_ in sd["state"].items():
                n = flat[int(i)]
                st.view_of(self.m, n).copy_(s_["exp_avg"]); st.view_of(self.v, n).copy_(s_["exp_avg_sq"])
                steps.add(int(float(s_["step"])))
        assert len(steps) <= 1, "per-parameter step counts differ: not representable by the fused optimizer"
        self.step_count = steps.pop() if steps else 0
        self.step_dev.fill_(self.step_count)
        for g, s_ in zip(self.param_groups, groups):
            for k in ("lr", "weight_decay", "betas", "eps", "initial_lr"):
                if k in s_:
                    g[k] = s_[k]


class FusedSGD(FusedAdamW):
    """torch.optim.SGD(param_dicts, lr, momentum=0.9, weight_decay) -- the reference's `--sgd` optimizer (main_vg.py:263-265) --
    on the same flat buffers, schedules (deferred update under graph replay, device learning rates, data-parallel bf16
    gradients) and clip path as FusedAdamW: `m` is the momentum buffer, `v` is not allocated, rt_sgd_flat is the kernel."""
    SGD = True

    def __init__(self, model, lr=1e-4, lr_backbone=1e-5, lr_bert=None, weight_decay=1e-4, momentum=0.9, lr_mask_branch_proj=1.0):
        super().__init__(model, lr=lr, lr_backbone=lr_backbone, lr_bert=lr_bert, weight_decay=weight_decay, betas=(momentum, 0.0),
                         lr_mask_branch_proj=lr_mask_branch_proj)
        self.v = self.m[:4]                      # never read by rt_sgd_flat (a 16-byte placeholder keeps the shared call sites simple)
        for g in self.param_groups:
            g["momentum"] = momentum

    def state_dict(self):
        """torch.optim.SGD's format: per-parameter `momentum_buffer` keyed by the parameter's index in the reference's group order."""
        if self._flush_pending is not None:
            self._flush_pending()
        st = self.model.store
        state, groups, idx = {}, [], 0
        for g, ns in zip(self.param_groups, self._names):
            ids = []
            for n in ns:
                if self.step_count > 0:
                    state[idx] = {"momentum_buffer": st.view_of(self.m, n).detach().clone(memory_format=torch.contiguous_format)}
                ids.append(idx); idx += 1
            pg = {k: v for k, v in g.items() if k not in ("params", "group_id", "betas", "eps")}
            pg.update(dampening=0, nesterov=False, maximize=False, foreach=None, differentiable=False, fused=None, params=ids)
            groups.append(pg)
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        if self._flush_pending is not None:
            self._flush_pending()
        st = self.model.store
        flat = [n for ns in self._names for n in ns]
        assert sum(len(g["params"]) for g in sd["param_groups"]) == len(flat), "optimizer state does not match this model's parameters"
        with torch.no_grad():
            for i, s_ in sd["state"].items():
                if s_.get("momentum_buffer") is not None:
                    st.view_of(self.m, flat[int(i)]).copy_(s_["momentum_buffer"])
        self.step_count = 1 if sd["state"] else 0
        self.step_dev.fill_(self.step_count)
        for g, s_ in zip(self.param_groups, sd["param_groups"]):
            for k in ("lr", "weight_decay", "momentum", "initial_lr"):
                if k in s_:
                    g[k] = s_[k]


def build_optimizer(model, args):
    """The optimizer main_vg.py:234-268 builds, on the fused kernels: AdamW, or SGD(momentum 0.9) with --sgd."""
    if getattr(args, "sgd", False):
        return FusedSGD(model, lr=args.lr, lr_backbone=args.lr_backbone, weight_decay=args.weight_decay,
                        lr_mask_branch_proj=getattr(args, "lr_mask_branch_proj", 1.0))
    return FusedAdamW(model, lr=args.lr, lr_backbone=args.lr_backbone, weight_decay=args.weight_decay,
                      lr_mask_branch_proj=getattr(args, "lr_mask_branch_proj", 1.0))
