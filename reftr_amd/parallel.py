"""Data parallelism for the RefTR step (reference: torch DistributedDataParallel, main_vg.py:290-296).

One process per GPU, full replica, the per-GPU batch fixed (weak scaling).  The only bandwidth-carrying
exchange is the gradient all-reduce.  Because every gradient lives in ONE flat fp32
buffer, the exchange is a few LARGE all-reduces over contiguous slices of it (RCCL over xGMI when the
backend is 'nccl'; gloo on CPU in the tests) instead of ~470 per-tensor buckets; the 1/world average is
folded into the fused AdamW kernel (grad_scale), and the clip norm is computed after the exchange, as DDP +
clip_grad_norm_ do in the reference loop (engine_vg.py:61-63).
"""
import os

import torch
import torch.distributed as dist
from torch import nn


class DistributedDataParallel(nn.Module):
    """Exchange schedule: the flat gradient buffer is laid out [main | mask head | ResNet | BERT] (models/store.py).  In data-
    parallel mode backward finishes the slices in this order (RefTR.BOUNDARIES) and the asynchronous all-reduce of a slice is
    launched the moment it is final, on RCCL's stream, under the backward that is still running:

      REFTR_DDP_SCHEDULE=interleave (default): BERT's backward stays on the language stream, a third beside each ResNet stage
        main   decoder / encoder / heads / input_proj (+ mask head)        under everything that follows
        pair4  BERT layers 11-8 + pooler and ResNet layer4                  under BERT 7-0 / layer3-2
        pair3  BERT layers 7-4 and ResNet layer3                            under BERT 3-0 / layer2
        (end)  BERT layers 3-0 + embeddings and ResNet layer2, then everything is waited for (exposed: ~120 MB in bf16)
      REFTR_DDP_SCHEDULE=serial: BERT's backward on the main stream in front of the ResNet's (+1.6 ms of serialised compute)
        main   decoder / encoder / heads / input_proj (+ mask head)        under the BERT backward
        bert_hi, bert_mid, bert   BERT layers 11-8 + pooler | 7-4 | 3-0 + embeddings (the layers' flat order is 0..11,
                                  backward walks 11..0)                    under the rest of BERT and the ResNet backward
        layer4 ResNet layer4 (64 % of the ResNet bytes)                    under layer3 / layer2
        (end)  ResNet layer2-3, then everything is waited for.

    `reduce_phase(name)` / `reduce_late` are the hook points; CapturedTrainStep calls them between its graphs, the eager loop
    gets them from the model's backward."""

    def __init__(self, module, n_chunks=None, broadcast=True, overlap=True):
        super().__init__()
        self.module = module
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        # REFTR_DDP_FORCE=1: run the exchange schedule (single-rank RCCL all-reduces) even with one process, so the
        # multi-GPU code path can be exercised on a one-GPU box
        self.active = self.world > 1 or (dist.is_initialized() and os.environ.get("REFTR_DDP_FORCE") == "1")
        # pieces per exchange are at most 1/n_chunks of the buffer: fewer, larger all-reduces cost fewer launches beside the
        # backward and suit the ring's per-message latency; more pieces start moving earlier
        self.n_chunks = 2 if n_chunks is None else n_chunks
        self.overlap = overlap
        self._works = []
        self.timing = None        # a list: reduce_late appends (event before the waits, event after) = exposed exchange time
        # Exchange dtype.  bf16 (default) halves the bytes on the xGMI links -- the exchange is link-bound at 2 and 4 GPUs
        # and a third of the step at 8: every slice is rounded to bf16 once it is final, summed by RCCL in
        # bf16, and clip + AdamW read the bf16 sums directly (no unpack pass).  The rounding (2^-9 relative per element) is
        # below the noise the bf16 GEMM operands already put on every gradient.  REFTR_DDP_DTYPE=fp32: the reference's
        # fp32 exchange.
        self.bf16 = self.active and os.environ.get("REFTR_DDP_DTYPE", "bf16") == "bf16"
        if self.bf16:
            module.store.flat_g16 = torch.zeros_like(module.store.flat_g, dtype=torch.bfloat16)
            # Round 4: the weight-gradient launches write the bf16 twin of every registered matrix themselves (rt_conv_wgrad_desc.g16),
            # so a slice's rounding pass only touches what they do not produce (biases, norm parameters, embeddings: 4 % of the
            # buffer) instead of reading 607 MB and writing 304 MB per step (CPU tensors: the full rounding copy).
            self.twin = module.store.flat_g.is_cuda
            if self.twin:
                from . import hip as H
                st = module.store
                import weakref
                for ptr, (off, n) in st._ow.items():
                    H._G16_MAP[ptr] = (weakref.ref(st), st.flat_g16.data_ptr() + 2 * off)
                self._round_tables = {}
        else:
            self.twin = False
        # REFTR_COMM=abi: the all-reduces go through the library's own RCCL binding (rt_comm_*, include/reftr_hip.h) on a
        # dedicated exchange stream instead of through torch.distributed's process group; the rendezvous that main_vg.py set up
        # (util/misc.py:392-431) only carries the 128-byte communicator id from rank 0.  Default: torch.distributed (the path
        # the CPU/gloo tests and the single-rank GPU tests exercise on every build).
        self.comm = None
        if self.active and os.environ.get("REFTR_COMM", "torch") == "abi" and module.store.flat_g.is_cuda:
            from . import hip as H
            uid = [H.Comm.unique_id() if dist.get_rank() == 0 else None]
            if self.world > 1:
                dist.broadcast_object_list(uid, src=0)
            self.comm = H.Comm(uid[0], dist.get_rank(), self.world)
            self._cstream = torch.cuda.Stream()
            self._cpending = False
        module._grad_scale = 1.0 / self.world
        if self.world > 1 and broadcast:      # DDP constructor: parameters + buffers from rank 0 (C2)
            for buf in module.store.flat.values():
                dist.broadcast(buf, src=0)
            module.mark_dirty(full=True)
        self.phases = list(module.active_boundaries()) if overlap else []
        if not self.active:
            return
        if overlap:
            module._slice_bounds = self.slice_bounds()     # backward clears never-written matrices of a slice in front of its exchange
        if overlap:
            # REFTR_DDP_PHASES: comma list of the boundaries to exchange at (default: all); a boundary that is left out hands
            # its slice to the next one that is kept (the end if none)
            want = os.environ.get("REFTR_DDP_PHASES")
            self.phases = [b for b in module.active_boundaries() if want is None or b in want.split(",")]
            for name in self.phases:
                module._phase_hooks.setdefault(name, []).append(lambda name=name: self.reduce_phase(name))
            module._post_backward_hooks.append(self.reduce_late)
        else:
            self.phases = []
            module._post_backward_hooks.append(self.allreduce_gradients)

    def forward(self, samples):
        return self.module(samples)

    def _split(self, a, b, max_elems):
        if b <= a:
            return []
        n = max(1, -(-(b - a) // max_elems))
        step = -(-(b - a) // n)
        step = (step + 1023) // 1024 * 1024
        return [(x, min(x + step, b)) for x in range(a, b, step)]

    def chunk_bounds(self):
        n = self.module.store.flat_g.numel()
        step = (n + self.n_chunks - 1) // self.n_chunks
        step = (step + 1023) // 1024 * 1024
        return [(a, min(a + step, n)) for a in range(0, n, step)]

    def slice_bounds(self):
        """{boundary or 'end': (a, b) or [(a, b), (c, d)]} -- the piece(s) of flat_g that become final at each boundary (the
        interleaved schedule pairs a BERT third with a ResNet stage)."""
        from .models import layout as L
        st = self.module.store
        cfg = self.module.cfg
        off = lambda n: st.offset[n][1]                                   # noqa: E731
        ma, mb = st.group_range[L.GROUP_MAIN]
        ka, kb = st.group_range[L.GROUP_MASK]
        ra, rb = st.group_range[L.GROUP_BACKBONE]
        ba, bb = st.group_range[L.GROUP_BERT]
        assert ma == 0 and ka == mb and ra == kb and ba == rb, "flat layout [main | mask | ResNet | BERT]"
        nl = cfg.bert.layers
        lay = lambda i: off(f"lang_backbone.encoder.layer.{i}.attention.self.query.weight")   # noqa: E731
        hi, mid = (lay((2 * nl) // 3), lay(nl // 3)) if nl >= 3 else (ba, ba)
        frozen_bb = rb == ra                                                 # --lr_backbone 0: no ResNet slice at all
        l4 = ra if frozen_bb else off("img_backbone.0.body.layer4.0.conv1.weight")
        assert ba <= mid <= hi <= bb and ra <= l4 <= rb
        if self.module.dp_schedule == "interleave":
            l3 = ra if frozen_bb else off("img_backbone.0.body.layer3.0.conv1.weight")
            assert ra <= l3 <= l4
            cuts = self.module.bert_cuts()
            if len(cuts) == 1:              # halves: BERT is complete (embeddings included) at 'pair3'; only ResNet layer2 is left
                half = lay(next(iter(cuts)))
                assert ba <= half <= bb
                return {"main": (ma, kb), "pair4": [(l4, rb), (half, bb)], "pair3": [(l3, l4), (ba, half)], "end": [(ra, l3)]}
            return {"main": (ma, kb), "pair4": [(l4, rb), (hi, bb)], "pair3": [(l3, l4), (mid, hi)], "end": [(ra, l3), (ba, mid)]}
        return {"main": (ma, kb), "bert_hi": (hi, bb), "bert_mid": (mid, hi), "bert": (ba, mid),
                "layer4": (l4, rb), "end": (ra, l4)}

    def phase_bounds(self):
        """{boundary or 'end': [chunk, ...]} with the slices of the boundaries that are not exchanged at (REFTR_DDP_PHASES)
        merged into the next kept one; chunks are at most 1/n_chunks of the buffer."""
        sl = self.slice_bounds()
        total = self.module.store.flat_g.numel()
        per = max(1, -(-total // self.n_chunks))
        out, carry = {}, []
        for name in list(self.module.BOUNDARIES) + ["end"]:
            carry += sl[name] if isinstance(sl[name], list) else [sl[name]]
            if name in self.phases or name == "end":
                out[name] = [c for a, b in carry for c in self._split(a, b, per)]
                carry = []
        return out

    def _launch(self, bounds):
        st = self.module.store
        g = st.flat_g
        # an exchange touches the gradient buffer between backward and the clip: what the weight-gradient epilogues collected in
        # the norm slots (and what the backward pre-reduced of the BERT slice) describes THIS rank's gradients, not the sums the
        # buffer holds afterwards -- every schedule (overlap or not, bf16 or fp32) falls back to the pass over the exchanged buffer
        st.norm_valid = False
        self.module._norm_split = None
        if self.bf16:
            g16 = st.flat_g16
            if self.twin:
                self._round_complement(bounds)        # the matrices' twins are already there (written by their producers)
            else:
                for a, b in bounds:
                    if b > a:
                        g16[a:b].copy_(g[a:b])        # round the final slice; the all-reduce is ordered behind it
            g = g16
        if self.comm is not None:
            self._cstream.wait_stream(torch.cuda.current_stream())       # behind the slice's producers (and its rounding)
            self.comm.allreduce([g[a:b] for a, b in bounds if b > a], stream=self._cstream)
            self._cpending = True
            return
        self._works += [dist.all_reduce(g[a:b], op=dist.ReduceOp.SUM, async_op=True) for a, b in bounds if b > a]

    def _round_complement(self, bounds):
        """bf16 twins of everything inside `bounds` that is NOT a registered weight matrix (one launch over a cached chunk table)."""
        from . import hip as H
        st = self.module.store
        key = tuple(bounds)
        ent = self._round_tables.get(key)
        if ent is None:
            mats = sorted(st._ow.values())
            chunks = []
            for a, b in bounds:
                pos = a
                for off, n in mats + [(b, 0)]:
                    if off + n <= pos or off >= b:
                        if off >= b:
                            off = b
                        else:
                            continue
                    lo, hi = pos, min(off, b)
                    while lo < hi:
                        c = min(16384, hi - lo)
                        chunks += [lo, c]; lo += c
                    pos = max(pos, min(off + n, b))
                    if pos >= b:
                        break
            ent = self._round_tables[key] = (torch.tensor(chunks or [0, 0], dtype=torch.int64, device=st.device), len(chunks) // 2)
        H.round_chunks(st.flat_g, st.flat_g16, ent[0], ent[1])

    def reduce_phase(self, name):
        self._launch(self.phase_bounds()[name])

    def exchange_bytes(self):
        """bytes this rank hands to the all-reduces of one step (the whole flat gradient buffer, in the exchange dtype)"""
        n = sum(b - a for chunks in self.phase_bounds().values() for a, b in chunks) if self.overlap else self.module.store.flat_g.numel()
        return n * (2 if self.bf16 else 4)

    def reduce_late(self):
        self._launch(self.phase_bounds()["end"])
        ev = None
        if self.timing is not None and self.module.store.flat_g.is_cuda:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        for w in self._works:
            w.wait()
        if self.comm is not None and self._cpending:
            torch.cuda.current_stream().wait_stream(self._cstream)
            self._cpending = False
        if ev is not None:
            ev[1].record()
            self.timing.append(ev)
        self._works = []

    def allreduce_gradients(self):
        self._launch(self.chunk_bounds())
        for w in self._works:
            w.wait()
        if self.comm is not None and self._cpending:
            torch.cuda.current_stream().wait_stream(self._cstream)
            self._cpending = False
        self._works = []
