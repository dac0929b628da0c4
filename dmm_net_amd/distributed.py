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
messages win), each bucket is all-reduced asynchronously -- with ``overlap=True`` as soon as autograd has produced
its last gradient, i.e. under the rest of the backward pass -- then waited, scaled by 1/world and scattered back.
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
    """Bucketed mean all-reduce of ``param.grad`` over the default process group.

    ``overlap=False``: call ``all_reduce_mean()`` after ``backward()``.
    ``overlap=True``: buckets are laid out in REVERSE parameter order (gradients arrive last layer first) and
    post-accumulate-grad hooks copy each gradient into its bucket as autograd produces it; the moment a bucket is
    complete its all-reduce is issued asynchronously, so the collectives of the late layers run under the backward of
    the early ones (what DDP does for the reference, train.py:178-184).  Collectives are ALWAYS issued in bucket-index
    order (a complete bucket waits for its predecessors), because RCCL pairs collectives across ranks by issue order
    and per-rank autograd graphs may differ (unused parameters, videos without live templates).  ``finish()`` after
    ``backward()`` issues the buckets that never completed (parameters without a gradient this step count as zeros,
    like DDP with ``find_unused_parameters=True``, train.py:181), waits, scales by 1/world and scatters back;
    parameters without a gradient on EVERY rank keep ``grad = None`` (``track_unused``: one [n_params] MAX all-reduce)
    and, being known idle on every rank alike, no longer hold their bucket back in the next step; should such a
    parameter produce a gradient after its bucket went out (graphs that change from step to step), ``finish()`` reduces
    it in one extra collective every rank derives from the used-mask alike -- no rank raises while the others wait.
    A second ``backward()`` before ``finish()`` raises (the in-flight buckets would drop the accumulated part).
    """

    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_mb: float = 64.0, overlap: bool = False,
                 track_unused: bool = True):
        self.track_unused = bool(track_unused)
        self.launch_log: List[int] = []           # bucket indices in the order their collectives were issued
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.bucket_bytes = int(bucket_mb * (1 << 20))
        self.overlap = bool(overlap)
        order = list(reversed(self.params)) if self.overlap else self.params
        self.buckets: List[List[torch.nn.Parameter]] = []
        cur, cur_bytes, key = [], 0, None
        for p in order:
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
        self._flat: List[Optional[torch.Tensor]] = [None] * len(self.buckets)
        self._slot = {}
        for bi, bucket in enumerate(self.buckets):
            off = 0
            for p in bucket:
                self._slot[id(p)] = (bi, off)
                off += p.numel()
        self._handles: List[Optional[object]] = [None] * len(self.buckets)
        self._filled = [set() for _ in self.buckets]
        self._ready = [False] * len(self.buckets)
        self._launched = [False] * len(self.buckets)
        self._next = 0                            # lowest bucket index not launched yet (fixed issue order)
        # parameters that received no gradient on ANY rank in the previous step (from the all-reduced used-mask, hence
        # identical on every rank -- e.g. the ResNet's unused fc): a bucket does not wait for them, so one idle
        # parameter in bucket 0 does not hold back every collective until finish()
        self._idle = [set() for _ in self.buckets]
        self._hooks = []
        self._zero_cache = {}
        if self.overlap:
            for p in self.params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))

    def num_collectives(self) -> int:
        return len(self.buckets)

    def remove_hooks(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []

    def _zeros_like_flat(self, p: torch.nn.Parameter) -> torch.Tensor:
        z = self._zero_cache.get(id(p))
        if z is None or z.device != p.device:
            z = self._zero_cache[id(p)] = torch.zeros(p.numel(), dtype=p.dtype, device=p.device)
        return z

    def _flat_of(self, bi: int) -> torch.Tensor:
        bucket = self.buckets[bi]
        n = sum(p.numel() for p in bucket)
        flat = self._flat[bi]
        if flat is None or flat.numel() != n:
            flat = torch.empty(n, dtype=bucket[0].dtype, device=bucket[0].device)
            self._flat[bi] = flat
        return flat

    @torch.no_grad()
    def _on_grad(self, p: torch.nn.Parameter):
        bi, off = self._slot[id(p)]
        if self._launched[bi] and id(p) in self._idle[bi] and id(p) not in self._filled[bi]:
            # idle on every rank in the previous step, so its bucket went out without waiting for it (carrying zeros in
            # its slot) -- and now it has a gradient after all.  Raising HERE would stop one rank while the others sit
            # in the collectives until the RCCL timeout; instead the gradient stays in p.grad and finish() reduces every
            # such parameter in one extra collective that all ranks derive from the all-reduced used-mask alike.
            return
        if self._launched[bi]:
            # a second backward() before finish() (gradient accumulation): the bucket already in flight holds only
            # the first backward's gradient and finish() would overwrite p.grad with it -- refuse instead of
            # silently dropping the accumulated part
            raise RuntimeError("GradBucketer(overlap=True): backward() ran again before finish(); call finish() "
                               "after every backward, or use overlap=False for gradient accumulation")
        self._filled[bi].add(id(p))
        if len(self._filled[bi] | self._idle[bi]) == len(self.buckets[bi]):
            # the bucket is complete: pack its gradients with ONE launch (a copy per parameter from the hook was ~340
            # small launches per ResNet-101 step on the autograd thread); known-idle parameters contribute zeros
            flat = self._flat_of(bi)
            parts = []
            for q in self.buckets[bi]:
                if id(q) in self._filled[bi]:
                    parts.append(q.grad.reshape(-1))
                else:
                    parts.append(self._zeros_like_flat(q))
            torch.cat(parts, out=flat)
            self._ready[bi] = True
            self._launch_ready_prefix()

    @torch.no_grad()
    def _launch_ready_prefix(self):
        """Collectives are matched across ranks by ISSUE ORDER (NCCL/RCCL), so buckets are always launched in index
        order: a complete bucket waits until every bucket before it has been launched (DDP does the same).  Ranks
        whose autograd graphs differ (unused parameters, skipped videos) then still pair bucket k with bucket k."""
        while self._next < len(self.buckets) and self._ready[self._next]:
            bi = self._next
            self._handles[bi] = dist.all_reduce(self._flat[bi], op=dist.ReduceOp.SUM, async_op=True)
            self._launched[bi] = True
            self.launch_log.append(bi)
            self._next += 1

    @torch.no_grad()
    def _scatter_back(self, bi: int, world: int, used: Optional[torch.Tensor] = None, used_off: int = 0):
        flat = self._flat[bi]
        flat.div_(world)
        off = 0
        dsts, srcs = [], []
        for k_i, p in enumerate(self.buckets[bi]):
            k = p.numel()
            if p.grad is None:
                # unused on THIS rank: materialise the mean only if some rank produced a gradient (a parameter
                # unused everywhere keeps grad None, so optimisers with weight decay leave it alone like under DDP)
                if used is None or bool(used[used_off + k_i]):
                    p.grad = flat[off:off + k].view_as(p).clone()
            else:
                dsts.append(p.grad)
                srcs.append(flat[off:off + k].view_as(p))
            off += k
        if dsts:
            torch._foreach_copy_(dsts, srcs)       # one multi-tensor launch per bucket, not one copy per parameter

    @torch.no_grad()
    def _used_mask(self) -> Optional[torch.Tensor]:
        """One tiny MAX all-reduce telling which parameters received a gradient on ANY rank.  Only issued when
        this rank has parameters without a gradient... which the other ranks cannot know, so it is issued whenever
        ``track_unused`` is set (default) -- [n_params] uint8, issued FIRST in the fixed collective order."""
        if not self.track_unused:
            return None
        dev = self.params[0].device if self.params else torch.device("cpu")
        m = torch.tensor([0 if p.grad is None else 1 for bucket in self.buckets for p in bucket], dtype=torch.int32,
                         device=dev)
        dist.all_reduce(m, op=dist.ReduceOp.MAX)
        m = m.cpu()
        k = 0
        for bi, bucket in enumerate(self.buckets):
            self._idle[bi] = {id(p) for j, p in enumerate(bucket) if not bool(m[k + j])}
            k += len(bucket)
        return m

    @torch.no_grad()
    def finish(self):
        """After ``backward()`` with ``overlap=True``: complete (in bucket order), wait, average, scatter back."""
        world = dist.get_world_size()
        for bi in range(self._next, len(self.buckets)):
            bucket = self.buckets[bi]
            flat = self._flat_of(bi)               # (re)fill in one launch: missing gradients are zeros
            torch.cat([self._zeros_like_flat(p) if p.grad is None else p.grad.reshape(-1) for p in bucket], out=flat)
            self._ready[bi] = True
        self._launch_ready_prefix()
        assert self._next == len(self.buckets)
        idle_before = [set(s) for s in self._idle]
        used = self._used_mask()                   # issued after all bucket collectives on every rank: same order
        uo = 0
        late = []                                  # went out as "known idle" (zeros) but got a gradient on some rank
        for bi in range(len(self.buckets)):
            self._handles[bi].wait()
            if used is not None:
                # (criterion and order are the same on every rank; the LOCAL gradient is taken before the scatter)
                late += [(p, None if p.grad is None else p.grad.clone()) for k_i, p in enumerate(self.buckets[bi])
                         if id(p) in idle_before[bi] and bool(used[uo + k_i])]
            self._scatter_back(bi, world, used, uo)
            uo += len(self.buckets[bi])
            self._handles[bi] = None
            self._filled[bi] = set()
            self._ready[bi] = False
            self._launched[bi] = False
        self._next = 0
        if late:
            # the same list on every rank (it comes from the MAX-reduced used-mask and the previous step's idle sets,
            # both identical everywhere): one more flat collective, in parameter order
            flat = torch.cat([(g if g is not None else torch.zeros_like(p)).reshape(-1) for p, g in late])
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
            flat.div_(world)
            off = 0
            for p, _ in late:
                p.grad = flat[off:off + p.numel()].view_as(p).clone()
                off += p.numel()

    @torch.no_grad()
    def all_reduce_mean(self):
        """grad <- mean over ranks (params whose grad is None are treated as zeros, like DDP with
        find_unused_parameters=True, train.py:181).  With ``overlap=True`` this is ``finish()``."""
        if self.overlap:
            return self.finish()
        world = dist.get_world_size()
        handles = []
        for i, bucket in enumerate(self.buckets):
            flat = self._flat_of(i)
            torch.cat([self._zeros_like_flat(p) if p.grad is None else p.grad.reshape(-1) for p in bucket], out=flat)
            handles.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True))
            self.launch_log.append(i)
        used = self._used_mask()
        uo = 0
        for i in range(len(self.buckets)):
            handles[i].wait()
            self._scatter_back(i, world, used, uo)
            uo += len(self.buckets[i])


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
