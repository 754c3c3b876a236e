"""Host-side utilities with the reference's names and behaviour (util/misc.py of ubc-vision/RefTR):
NestedTensor (:308-333), nested_tensor_from_tensor_list (:288-305), distributed helpers (:351-431),
reduce_dict (:136-160); the loops' meters are reftr_amd's own StatBoard (one table, one device -> host copy per iteration)
instead of the reference's SmoothedValue / MetricLogger objects (:31-90,163-250)."""
import datetime
import os
import time
from collections import deque

import torch
import torch.distributed as dist


class NestedTensor(object):
    def __init__(self, tensors, mask):
        self.tensors = tensors
        self.mask = mask

    def to(self, device, non_blocking=False):
        mask = self.mask.to(device, non_blocking=non_blocking) if self.mask is not None else None
        return NestedTensor(self.tensors.to(device, non_blocking=non_blocking), mask)

    def record_stream(self, *args, **kwargs):
        self.tensors.record_stream(*args, **kwargs)
        if self.mask is not None:
            self.mask.record_stream(*args, **kwargs)

    def decompose(self):
        return self.tensors, self.mask

    def __repr__(self):
        return str(self.tensors)


def nested_tensor_from_tensor_list(tensor_list):
    """Pad [3, h, w] images to the batch maximum; mask is True on padding."""
    if tensor_list[0].ndim != 3:
        raise ValueError("not supported")
    c = tensor_list[0].shape[0]
    hmax = max(t.shape[1] for t in tensor_list)
    wmax = max(t.shape[2] for t in tensor_list)
    b = len(tensor_list)
    dtype, device = tensor_list[0].dtype, tensor_list[0].device
    tensor = torch.zeros((b, c, hmax, wmax), dtype=dtype, device=device)
    mask = torch.ones((b, hmax, wmax), dtype=torch.bool, device=device)
    for img, pad_img, m in zip(tensor_list, tensor, mask):
        pad_img[:, : img.shape[1], : img.shape[2]].copy_(img)
        m[: img.shape[1], : img.shape[2]] = False
    return NestedTensor(tensor, mask)


def is_dist_avail_and_initialized():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if is_dist_avail_and_initialized() else 1


def get_rank():
    return dist.get_rank() if is_dist_avail_and_initialized() else 0


def is_main_process():
    return get_rank() == 0


def save_on_master(*args, **kwargs):
    """util/misc.py:387-389."""
    if is_main_process():
        torch.save(*args, **kwargs)


def init_distributed_mode(args):
    """env:// rendezvous, one process per GPU; backend 'nccl' is RCCL on ROCm (util/misc.py:392-431)."""
    if "RANK" in os.environ and "WORLD_SIZE" in os.environ:
        args.rank = int(os.environ["RANK"])
        args.world_size = int(os.environ["WORLD_SIZE"])
        args.gpu = int(os.environ.get("LOCAL_RANK", 0))
    else:
        args.distributed = False
        return
    args.distributed = True
    backend = getattr(args, "dist_backend", "nccl")
    if backend == "nccl":
        torch.cuda.set_device(args.gpu)
    dist.init_process_group(backend=backend, init_method=getattr(args, "dist_url", "env://"),
                            world_size=args.world_size, rank=args.rank)
    dist.barrier()


def reduce_dict(input_dict, average=True):
    """All-reduce a dict of 0-d tensors in one stacked call (util/misc.py:136-160)."""
    world_size = get_world_size()
    if world_size < 2:
        return input_dict
    with torch.no_grad():
        names = sorted(input_dict.keys())
        values = torch.stack([input_dict[k] for k in names], dim=0)
        dist.all_reduce(values)
        if average:
            values /= world_size
        return {k: v for k, v in zip(names, values)}


class StatBoard:
    """The meters of a training / evaluation loop as ONE table instead of one object per scalar (the reference keeps a
    SmoothedValue per loss and calls .item() on each of them every iteration, util/misc.py:31-90,163-250: ~25 host syncs per
    iteration, which were free at 700 ms a step and are not at 7 ms).

    Two feeds, both sync-free by themselves:
      add(**scalars)            host numbers of one iteration (the engine reads ALL of an iteration's scalars with one stacked
                                device -> host copy, the same one that serves the reference's `math.isfinite(loss)` check);
      add_device(names, vec)    a 1-D device tensor of one iteration's scalars: accumulated on the device (one add), read back
                                only when something is printed or summarised.
    Statistics per name: last value, mean of the last `window` iterations, global mean (= the reference's `global_avg`,
    what train_one_epoch / evaluate return)."""

    def __init__(self, window=20, delimiter="  "):
        self.window, self.delimiter = int(window), delimiter
        self.total, self.count, self.last, self.recent = {}, {}, {}, {}
        self._dev = None                   # [names, fp64 sums on the device, iterations, last vector]

    # ---- feeds
    def add(self, **scalars):
        tens = [k for k, v in scalars.items() if torch.is_tensor(v)]
        if tens:                           # tensors (the eager loop body returns them): ONE stacked copy for all of them
            vals = torch.stack([scalars[k].detach().reshape(()).float() for k in tens]).tolist()
            scalars = dict(scalars, **dict(zip(tens, vals)))
        for k, v in scalars.items():
            v = float(v)
            self.total[k] = self.total.get(k, 0.0) + v
            self.count[k] = self.count.get(k, 0) + 1
            self.last[k] = v
            self.recent.setdefault(k, deque(maxlen=self.window)).append(v)

    def add_device(self, names, vec):
        names = tuple(names)
        if self._dev is not None and self._dev[0] != names:
            self._drain()
        if self._dev is None:
            self._dev = [names, torch.zeros(len(names), dtype=torch.float64, device=vec.device), 0, None]
        self._dev[1] += vec.detach().to(torch.float64)
        self._dev[2] += 1
        self._dev[3] = vec.detach()

    def _drain(self):
        if self._dev is None:
            return
        names, sums, n, last = self._dev
        self._dev = None
        host = torch.cat([sums, last.to(torch.float64)]).tolist()          # the only device -> host copy of this feed
        for i, k in enumerate(names):
            self.total[k] = self.total.get(k, 0.0) + host[i]
            self.count[k] = self.count.get(k, 0) + n
            self.last[k] = host[len(names) + i]
            self.recent.setdefault(k, deque(maxlen=self.window)).append(host[len(names) + i])

    # ---- read-outs
    def global_avg(self):
        self._drain()
        return {k: self.total[k] / max(self.count[k], 1) for k in self.total}

    def synchronize_between_processes(self):
        """sums and counts of every meter over the ranks, one all-reduce (the reference: one barrier + all-reduce per meter)"""
        self._drain()
        if not is_dist_avail_and_initialized():
            return
        names = sorted(self.total)
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([self.total[k] for k in names] + [float(self.count[k]) for k in names], dtype=torch.float64, device=dev)
        dist.all_reduce(t)
        t = t.tolist()
        for i, k in enumerate(names):
            self.total[k], self.count[k] = t[i], int(t[len(names) + i])

    def __str__(self):
        self._drain()
        parts = []
        for k in self.total:
            r = self.recent[k]
            parts.append("{}: {:.4g} (last {}: {:.4g}, all: {:.4g})".format(k, self.last[k], len(r), sum(r) / max(len(r), 1),
                                                                            self.total[k] / max(self.count[k], 1)))
        return self.delimiter.join(parts)

    def log_every(self, iterable, print_freq, header=""):
        start = time.time()
        n = len(iterable)
        for i, obj in enumerate(iterable):
            yield obj
            if print_freq and (i % print_freq == 0 or i == n - 1) and is_main_process():
                el = time.time() - start
                eta = datetime.timedelta(seconds=int(el / (i + 1) * (n - i - 1)))
                print(f"{header} [{i}/{n}] eta: {eta} {self} time: {el / (i + 1):.4f}")
