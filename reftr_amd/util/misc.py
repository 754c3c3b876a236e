"""Host-side utilities with the reference's names and behaviour (util/misc.py of ubc-vision/RefTR):
NestedTensor (:308-333), nested_tensor_from_tensor_list (:288-305), distributed helpers (:351-431),
reduce_dict (:136-160), SmoothedValue / MetricLogger (:31-90,163-250)."""
import datetime
import os
import time
from collections import defaultdict, deque

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


class SmoothedValue(object):
    def __init__(self, window_size=20, fmt=None):
        self.deque = deque(maxlen=window_size)
        self.total = 0.0
        self.count = 0
        self.fmt = fmt or "{median:.4f} ({global_avg:.4f})"

    def update(self, value, n=1):
        self.deque.append(value)
        self.count += n
        self.total += value * n

    def synchronize_between_processes(self):
        if not is_dist_avail_and_initialized():
            return
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([self.count, self.total], dtype=torch.float64, device=dev)
        dist.barrier()
        dist.all_reduce(t)
        t = t.tolist()
        self.count, self.total = int(t[0]), t[1]

    @property
    def median(self):
        return torch.tensor(list(self.deque)).median().item()

    @property
    def avg(self):
        return torch.tensor(list(self.deque), dtype=torch.float32).mean().item()

    @property
    def global_avg(self):
        return self.total / max(self.count, 1)

    @property
    def max(self):
        return max(self.deque)

    @property
    def value(self):
        return self.deque[-1]

    def __str__(self):
        return self.fmt.format(median=self.median, avg=self.avg, global_avg=self.global_avg, max=self.max, value=self.value)


class MetricLogger(object):
    def __init__(self, delimiter="\t"):
        self.meters = defaultdict(SmoothedValue)
        self.delimiter = delimiter

    def update(self, **kwargs):
        for k, v in kwargs.items():
            if isinstance(v, torch.Tensor):
                v = v.item()
            self.meters[k].update(float(v))

    def add_meter(self, name, meter):
        self.meters[name] = meter

    def synchronize_between_processes(self):
        for meter in self.meters.values():
            meter.synchronize_between_processes()

    def __str__(self):
        return self.delimiter.join("{}: {}".format(n, str(m)) for n, m in self.meters.items())

    def log_every(self, iterable, print_freq, header=None):
        header = header or ""
        start = time.time()
        n = len(iterable)
        for i, obj in enumerate(iterable):
            yield obj
            if print_freq and (i % print_freq == 0 or i == n - 1) and is_main_process():
                el = time.time() - start
                eta = datetime.timedelta(seconds=int(el / (i + 1) * (n - i - 1)))
                print(f"{header} [{i}/{n}] eta: {eta} {self} time: {el / (i + 1):.4f}")
