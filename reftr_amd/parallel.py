"""Data parallelism for the RefTR step (reference: torch DistributedDataParallel, main_vg.py:290-296).

One process per GPU, full replica, the per-GPU batch fixed (weak scaling).  The only bandwidth-carrying
exchange is the gradient all-reduce.  Because every gradient lives in ONE flat fp32
buffer, the exchange is a few LARGE all-reduces over contiguous slices of it (RCCL over xGMI when the
backend is 'nccl'; gloo on CPU in the tests) instead of ~470 per-tensor buckets; the 1/world average is
folded into the fused AdamW kernel (grad_scale), and the clip norm is computed after the exchange, as DDP +
clip_grad_norm_ do in the reference loop (engine_vg.py:61-63).
"""
import torch
import torch.distributed as dist
from torch import nn


class DistributedDataParallel(nn.Module):
    def __init__(self, module, n_chunks=8, broadcast=True):
        super().__init__()
        self.module = module
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.n_chunks = n_chunks
        module._grad_scale = 1.0 / self.world
        if self.world > 1 and broadcast:      # DDP constructor: parameters + buffers from rank 0 (C2)
            for buf in module.store.flat.values():
                dist.broadcast(buf, src=0)
            module.mark_dirty(full=True)
        module._post_backward_hooks.append(self.allreduce_gradients)

    def forward(self, samples):
        return self.module(samples)

    def chunk_bounds(self):
        n = self.module.store.flat_g.numel()
        step = (n + self.n_chunks - 1) // self.n_chunks
        step = (step + 1023) // 1024 * 1024
        return [(a, min(a + step, n)) for a in range(0, n, step)]

    def allreduce_gradients(self):
        if self.world < 2:
            return
        g = self.module.store.flat_g
        works = [dist.all_reduce(g[a:b], op=dist.ReduceOp.SUM, async_op=True) for a, b in self.chunk_bounds()]
        for w in works:
            w.wait()
