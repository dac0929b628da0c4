"""Multi-GPU plumbing for the matching path: one process per GPU, ``torch.distributed`` ("nccl" = RCCL on
ROCm, over xGMI).

The forward of the layer has NO exchange step: every (video, frame) is independent, so the
batch of frames is simply partitioned across ranks (``shard_range``), like the reference's
``DistributedSampler`` (train.py:88-95) / per-GPU eval processes (eval.py:57-61).

Training needs exactly one collective: the mean of the gradients across ranks.  The reference gets it from
DDP's 25 MB buckets (train.py:178-184) and then repeats it with one un-awaited async all-reduce PER
PARAMETER TENSOR (``average_gradients``, train.py:62-68 -- ~230-390 tiny collectives whose handles are never
waited on).  ``GradBucketer`` is the semantic equivalent done once: gradients are packed into a few large
flat buckets (xGMI is point-to-point, a ring all-reduce is bound by one ~153 GB/s link, so fewer/larger
messages win), each bucket is all-reduced asynchronously, then waited, scaled by 1/world and scattered back.
The matching layer itself owns no parameters (checkpoint neutral), so it contributes nothing to it.
"""
from __future__ import annotations

import os
from typing import Dict, Iterable, List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced [begin, end) of ``total`` frames for ``rank`` (first ``total % world`` ranks get one more)."""
    assert 0 <= rank < world
    base, rem = divmod(total, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def init_from_env(backend: Optional[str] = None, device: Optional[torch.device] = None):
    """env:// rendezvous like the reference (train.py:407-409); 127.0.0.1 unless MASTER_ADDR is set."""
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    kw = {}
    if backend == "nccl" and device is not None:
        kw["device_id"] = device
    dist.init_process_group(backend=backend, init_method="env://", **kw)
    return dist.get_rank(), dist.get_world_size()


class GradBucketer:
    """Bucketed mean all-reduce of ``param.grad`` over the default process group."""

    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_mb: float = 64.0):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.bucket_bytes = int(bucket_mb * (1 << 20))
        self.buckets: List[List[torch.nn.Parameter]] = []
        cur, cur_bytes, key = [], 0, None
        for p in self.params:
            k = (p.dtype, p.device)
            nbytes = p.numel() * p.element_size()
            if cur and (k != key or cur_bytes + nbytes > self.bucket_bytes):
                self.buckets.append(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
            key = k
        if cur:
            self.buckets.append(cur)
        self._flat = [None] * len(self.buckets)

    def num_collectives(self) -> int:
        return len(self.buckets)

    @torch.no_grad()
    def all_reduce_mean(self):
        """grad <- mean over ranks (params whose grad is None are treated as zeros, like DDP with
        find_unused_parameters=True, train.py:181)."""
        world = dist.get_world_size()
        handles = []
        for i, bucket in enumerate(self.buckets):
            n = sum(p.numel() for p in bucket)
            flat = self._flat[i]
            if flat is None or flat.numel() != n:
                flat = torch.empty(n, dtype=bucket[0].dtype, device=bucket[0].device)
                self._flat[i] = flat
            off = 0
            for p in bucket:
                k = p.numel()
                if p.grad is None:
                    flat[off:off + k].zero_()
                else:
                    flat[off:off + k].copy_(p.grad.reshape(-1))
                off += k
            handles.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True))
        for i, bucket in enumerate(self.buckets):
            handles[i].wait()
            flat = self._flat[i]
            flat.div_(world)
            off = 0
            for p in bucket:
                k = p.numel()
                if p.grad is None:
                    p.grad = flat[off:off + k].view_as(p).clone()
                else:
                    p.grad.copy_(flat[off:off + k].view_as(p))
                off += k


@torch.no_grad()
def reduce_loss_dict(loss_dict: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Mean of the scalar losses on rank 0 for logging (reference: maskrcnn_benchmark reduce_loss_dict,
    train.py:38, :310-311): one stacked [k] reduce instead of k scalars."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world < 2:
        return loss_dict
    names = sorted(loss_dict.keys())
    vals = torch.stack([loss_dict[k].detach().reshape(()) for k in names], 0)
    dist.reduce(vals, dst=0)
    if dist.get_rank() == 0:
        vals /= world
    return {k: v for k, v in zip(names, vals)}
