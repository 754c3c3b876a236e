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
    """Exchange schedule: the flat gradient buffer is laid out [main | mask head | ResNet | BERT] (models/store.py).  Backward
    finishes the main and BERT ranges first (phase 1); their all-reduces are launched asynchronously at that point and
    run on RCCL's stream while the ResNet backward (phase 2) computes; the ResNet range follows, then everything is
    waited for.  `reduce_early` / `reduce_late` are the two hook points; CapturedTrainStep calls them between its
    graphs, the eager loop gets them from the model's backward."""

    def __init__(self, module, n_chunks=8, broadcast=True, overlap=True):
        super().__init__()
        self.module = module
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        # REFTR_DDP_FORCE=1: run the exchange schedule (single-rank RCCL all-reduces) even with one process, so the
        # multi-GPU code path can be exercised on a one-GPU box
        self.active = self.world > 1 or (dist.is_initialized() and os.environ.get("REFTR_DDP_FORCE") == "1")
        self.n_chunks = n_chunks
        self.overlap = overlap
        self._works = []
        module._grad_scale = 1.0 / self.world
        if self.world > 1 and broadcast:      # DDP constructor: parameters + buffers from rank 0 (C2)
            for buf in module.store.flat.values():
                dist.broadcast(buf, src=0)
            module.mark_dirty(full=True)
        if not self.active:
            return
        if overlap:
            module._mid_backward_hooks.append(self.reduce_early)
            module._post_backward_hooks.append(self.reduce_late)
        else:
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

    def phase_bounds(self):
        """(early, late) lists of [a, b) slices of flat_g: early = main + BERT groups, late = ResNet group."""
        from .models import layout as L
        st = self.module.store
        total = st.flat_g.numel()
        per = max(1, -(-total // self.n_chunks))
        early = []
        for grp in (L.GROUP_MAIN, L.GROUP_MASK, L.GROUP_BERT):
            a, b = st.group_range[grp]
            early += self._split(a, b, per)
        a, b = st.group_range[L.GROUP_BACKBONE]
        return early, self._split(a, b, per)

    def _launch(self, bounds):
        g = self.module.store.flat_g
        self._works += [dist.all_reduce(g[a:b], op=dist.ReduceOp.SUM, async_op=True) for a, b in bounds if b > a]

    def reduce_early(self):
        self._launch(self.phase_bounds()[0])

    def reduce_late(self):
        self._launch(self.phase_bounds()[1])
        for w in self._works:
            w.wait()
        self._works = []

    def allreduce_gradients(self):
        self._launch(self.chunk_bounds())
        for w in self._works:
            w.wait()
        self._works = []
