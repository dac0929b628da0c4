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


def gpu_local_cpus(index: int):
    """-> (NUMA node, set of CPUs local to GPU ``index``) from its PCI device's sysfs entry, or (None, None) when the
    topology is not visible (a container without sysfs, node -1)."""
    try:
        pr = torch.cuda.get_device_properties(index)
        bdf = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        base = f"/sys/bus/pci/devices/{bdf}"
        node = int(open(base + "/numa_node").read().strip())
        cpus = set()
        for part in open(base + "/local_cpulist").read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        return (node, cpus) if node >= 0 and cpus else (None, None)
    except Exception:
        return None, None


def bind_host_threads_to_gpu(index: int, all_threads: bool = True):
    """Pin the host threads of this process to the CPUs local to GPU ``index`` -- call it once at the top of a trainer /
    evaluator process (one process per GPU, scripts/train/train_101.sh:27-28).  Why it matters on a two-socket MI355X host:
    a per-frame call of the layer is ~10 launches from the Python thread plus, in training, a hand-off to autograd's
    device thread and back; with the two threads on different sockets the same call measured 237 us against 183 us
    (``bench.py --config dropin``, round 5: unpinned runs were bimodal, pinned ones were not).
    ``all_threads``: also re-pin the threads that already exist (autograd's worker is created at the first backward and keeps
    the mask it was born with).  Returns (numa_node | None, restore) -- ``restore()`` puts the previous masks back; every
    failure is silent (the process stays unpinned)."""
    node, cpus = gpu_local_cpus(index)
    if node is None:
        return None, (lambda: None)
    saved = {}
    try:
        tids = [int(t) for t in os.listdir("/proc/self/task")] if all_threads else [0]
    except OSError:
        tids = [0]
    for tid in tids:
        try:
            old = os.sched_getaffinity(tid)
            allowed = cpus & old
            if allowed:
                os.sched_setaffinity(tid, allowed)
                saved[tid] = old
        except OSError:
            pass                                     # a thread that ended meanwhile, or no permission

    def restore():
        for tid, old in saved.items():
            try:
                os.sched_setaffinity(tid, old)
            except OSError:
                pass
    return (node if saved else None), restore


class GradBucketer:
    """Bucketed mean all-reduce of ``param.grad`` over the default process group, gradients kept as VIEWS of the flat
    buckets (DDP's ``gradient_as_bucket_view``; the reference gets the mean from DDP, train.py:178-184, and repeats it
    per tensor, train.py:62-68).

    Data path per step: ONE pack launch per bucket (``torch.cat(out=flat)``; skipped for gradients that already live in
    the bucket), one collective per bucket that also divides (``ReduceOp.AVG`` on RCCL; SUM + one in-place divide on
    backends without AVG, e.g. gloo), then ``p.grad`` is re-pointed at its slice of the reduced bucket -- no scatter
    copy.  The flat buffers are allocated once.  Until the next ``zero_grad`` the gradients alias the buckets: do not
    hold on to them across steps (same contract as DDP's bucket views).  With ``overlap=True`` the re-pointing happens in
    the hook that completes a bucket, so when ``backward()`` returns a gradient may already be (or be turning into) the
    reduced value -- read gradients after ``finish()``.

    ``overlap=False``: call ``all_reduce_mean()`` after ``backward()``.
    ``overlap=True``: buckets are laid out in REVERSE parameter order (gradients arrive last layer first);
    post-accumulate-grad hooks mark a bucket complete the moment its last gradient exists and its all-reduce is issued
    asynchronously, under the rest of the backward.  Collectives are ALWAYS issued in bucket-index order (a complete
    bucket waits for its predecessors), because RCCL pairs collectives across ranks by issue order and per-rank
    autograd graphs may differ (unused parameters, videos without live templates).  ``finish()`` after ``backward()``
    issues the buckets that never completed (parameters without a gradient count as zeros, like DDP with
    ``find_unused_parameters=True``, train.py:181) and waits.  A second ``backward()`` before ``finish()`` raises.

    Which parameters are used (``track_unused``).  Parameters without a gradient on EVERY rank keep ``grad = None`` (no
    zero gradients for Adam's weight decay) and do not hold their bucket back.  That is a cross-rank fact:
      * VERIFY mode (the first steps, and whenever the set changes): one [n_params] MAX all-reduce per step, read back
        on the host -- exact in the same step, but the read drains the stream (what DDP pays every step with
        ``find_unused_parameters``).  A parameter that was idle everywhere and produces a gradient after its bucket
        went out is reduced in one extra collective every rank derives from the mask alike.
      * STEADY mode (OPT-IN: ``steady_after=k``, after k consecutive identical masks; the default ``None`` stays in VERIFY
        mode like the reference's DDP with ``find_unused_parameters=True``, train.py:178-184 -- DMM's autograd graphs are data
        dependent: skipped videos, frames without live templates): the same reduction is issued asynchronously
        into pinned memory and looked at ONE STEP LATER, so nothing on the step waits for the host.  The used set is
        assumed unchanged (DDP's ``static_graph``); if the delayed mask disagrees, every rank falls back to VERIFY
        mode at the same step.  In the one step in between a parameter that was idle everywhere keeps ``grad = None``
        on every rank even if a rank produced a gradient for it (dropped consistently, never applied on one rank
        only), and a parameter that went idle everywhere receives a zero gradient instead of ``None``.  Every such
        fall-back is counted in ``steady_fallbacks`` and logged (``logging.getLogger("dmm_net_amd.distributed")``) --
        each time, not once per process.  Callers whose used set is fixed (bench.py's synthetic clips, a trainer that
        never skips a video) switch it on explicitly.
    """

    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_mb: float = 64.0, overlap: bool = False,
                 track_unused: bool = True, steady_after: Optional[int] = None, verify_layout: bool = True):
        self.track_unused = bool(track_unused)
        self.steady_after = steady_after if (steady_after and track_unused) else None
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
        # flat buckets, allocated ONCE, and every parameter's slice of its bucket in the parameter's shape
        self._flat: List[torch.Tensor] = []
        self._views: List[List[torch.Tensor]] = []
        self._slot = {}
        for bi, bucket in enumerate(self.buckets):
            flat = torch.zeros(sum(p.numel() for p in bucket), dtype=bucket[0].dtype, device=bucket[0].device)
            views, off = [], 0
            for k_i, p in enumerate(bucket):
                self._slot[id(p)] = (bi, k_i)
                views.append(flat[off:off + p.numel()].view(p.shape))
                off += p.numel()
            self._flat.append(flat)
            self._views.append(views)
        self._n = sum(len(b) for b in self.buckets)
        self._handles: List[Optional[object]] = [None] * len(self.buckets)
        self._filled = [set() for _ in self.buckets]
        self._ready = [False] * len(self.buckets)
        self._launched = [False] * len(self.buckets)
        self._adopted = [False] * len(self.buckets)   # views already adopted by the hook that completed the bucket
        self._next = 0                            # lowest bucket index not launched yet (fixed issue order)
        # parameters that received no gradient on ANY rank (from the all-reduced used-mask, hence identical on every
        # rank -- e.g. the ResNet's unused fc): a bucket does not wait for them
        self._idle = [set() for _ in self.buckets]
        self._hooks = []
        self._avg: Optional[bool] = None          # does the backend divide inside the collective?
        # used-mask protocol state
        self.mode = "verify"
        self.host_syncs = 0                       # blocking mask read-backs so far (tests / bench look at it)
        self.steady_fallbacks = 0                 # times a delayed mask disagreed and sent every rank back to VERIFY mode
        self._stable = 0
        self._last_mask: Optional[torch.Tensor] = None
        self._pending = None                      # steady mode: (pinned host mask, event) of the previous step
        self._pin_in = self._pin_out = None
        self.layout_verified = False
        if self.overlap:
            for p in self.params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
        if verify_layout and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            self.verify_layout()

    def layout_fingerprint(self) -> List[int]:
        """Two int64 words that identify what this rank will put on the wire and in which order: the number of buckets, every
        bucket's element count and dtype, every parameter's place and shape in its bucket, the issue order (overlap or not)
        and the used-mask settings."""
        import hashlib
        desc = [("overlap", self.overlap, "track", self.track_unused, "steady", self.steady_after, "n", self._n)]
        for bi, bucket in enumerate(self.buckets):
            desc.append((bi, str(bucket[0].dtype), int(self._flat[bi].numel()), tuple(tuple(p.shape) for p in bucket)))
        h = hashlib.sha256(repr(desc).encode()).digest()
        return [int.from_bytes(h[:8], "little", signed=True), int.from_bytes(h[8:16], "little", signed=True)]

    @torch.no_grad()
    def verify_layout(self):
        """ONE all_gather at start-up: every rank's bucket layout + issue order must be the same, or the per-bucket
        collectives would pair buffers of different sizes across ranks -- RCCL then hangs (or corrupts) instead of failing.
        A mismatch raises on EVERY rank, naming the ranks that differ (nobody can rehearse the 8-GPU run;
        it should diagnose itself)."""
        world, rank = dist.get_world_size(), dist.get_rank()
        dev = self.params[0].device if self.params and dist.get_backend() == "nccl" else torch.device("cpu")
        mine = torch.tensor(self.layout_fingerprint() + [len(self.buckets), self._n], dtype=torch.int64, device=dev)
        got = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(got, mine)
        rows = [g.cpu().tolist() for g in got]
        bad = [r for r in range(world) if rows[r] != rows[0]]
        if bad:
            raise RuntimeError(
                f"GradBucketer: rank {rank}: the gradient bucket layout differs between ranks (ranks {bad} differ from rank 0: "
                f"buckets / parameters per rank = {[(r[2], r[3]) for r in rows]}).  Every rank must build the bucketer from the "
                "same parameter list in the same order with the same bucket_mb / overlap / steady_after; the collectives "
                "would otherwise pair different buffers (an RCCL hang, not an error).")
        self.layout_verified = True

    def num_collectives(self) -> int:
        return len(self.buckets)

    def remove_hooks(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []

    # ------------------------------------------------------------------------------------------------ data path
    def _aliases(self, g: Optional[torch.Tensor], v: torch.Tensor) -> bool:
        return g is not None and g.data_ptr() == v.data_ptr() and g.is_contiguous() and g.shape == v.shape

    @torch.no_grad()
    def _pack(self, bi: int):
        """Bring bucket ``bi``'s gradients into its flat buffer: one ``cat`` when every gradient exists and lives
        elsewhere (the usual case after ``zero_grad(set_to_none=True)``: autograd hands fresh tensors), otherwise one
        multi-tensor copy for those that live elsewhere and one multi-tensor zero for the missing ones; gradients that
        already ARE their bucket slice (``zero_grad(set_to_none=False)``: autograd accumulated in place) cost nothing."""
        bucket, views = self.buckets[bi], self._views[bi]
        grads = [p.grad for p in bucket]
        alias = [self._aliases(g, v) for g, v in zip(grads, views)]
        if not any(alias) and all(g is not None for g in grads):
            torch.cat([g.reshape(-1) for g in grads], out=self._flat[bi])
            return
        src = [g for g, a in zip(grads, alias) if g is not None and not a]
        dst = [v for g, a, v in zip(grads, alias, views) if g is not None and not a]
        zero = [v for g, v in zip(grads, views) if g is None]
        if src:
            torch._foreach_copy_(dst, src)
        if zero:
            torch._foreach_zero_(zero)

    def _collective(self, flat: torch.Tensor):
        if self._avg is None:
            self._avg = dist.get_backend() == "nccl"            # RCCL divides in the collective; gloo has no AVG
        return dist.all_reduce(flat, op=dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM, async_op=True)

    @torch.no_grad()
    def _on_grad(self, p: torch.nn.Parameter):
        bi, _ = self._slot[id(p)]
        if self._launched[bi] and id(p) in self._idle[bi] and id(p) not in self._filled[bi]:
            # idle on every rank so far, so its bucket went out without waiting for it -- and now it has a gradient after
            # all.  Raising HERE would stop one rank while the others sit in the collectives until the RCCL timeout; the
            # gradient stays in p.grad and finish() deals with it (verify mode: one extra collective every rank derives
            # from the all-reduced used-mask alike; steady mode: dropped on every rank, then back to verify mode).
            self._filled[bi].add(id(p))
            return
        if self._launched[bi]:
            # a second backward() before finish() (gradient accumulation): the bucket already in flight holds only
            # the first backward's gradient -- refuse instead of silently dropping the accumulated part
            raise RuntimeError("GradBucketer(overlap=True): backward() ran again before finish(); call finish() "
                               "after every backward, or use overlap=False for gradient accumulation")
        self._filled[bi].add(id(p))
        if len(self._filled[bi] | self._idle[bi]) == len(self.buckets[bi]):
            self._pack(bi)
            # adopt the bucket views NOW, on the autograd thread while the GPU is busy with the rest of the backward: the
            # slices will hold the mean once the (in-place) collective has run, and nothing reads a gradient before
            # finish() has waited for it.  (Done in finish(), the ~340 ``p.grad = view`` assignments of a ResNet-101 were
            # ~1 ms of host time AFTER backward() -- exposed whenever the host is not far ahead of the GPU.)
            # (not the parameters that were idle everywhere so far: should one wake up, finish() needs its LOCAL gradient)
            idle = self._idle[bi]
            for q, v in zip(self.buckets[bi], self._views[bi]):
                if q.grad is not None and q.grad is not v and id(q) not in idle:
                    q.grad = v
            self._adopted[bi] = True
            self._ready[bi] = True
            self._launch_ready_prefix()

    @torch.no_grad()
    def _launch_ready_prefix(self):
        """Collectives are matched across ranks by ISSUE ORDER (NCCL/RCCL), so buckets are always launched in index
        order: a complete bucket waits until every bucket before it has been launched (DDP does the same).  Ranks
        whose autograd graphs differ (unused parameters, skipped videos) then still pair bucket k with bucket k."""
        while self._next < len(self.buckets) and self._ready[self._next]:
            bi = self._next
            self._handles[bi] = self._collective(self._flat[bi])
            self._launched[bi] = True
            self.launch_log.append(bi)
            self._next += 1

    @torch.no_grad()
    def _adopt(self, bi: int, world: int, used: Optional[List[bool]], used_off: int, keep_none=()):
        """After bucket ``bi``'s collective: divide where the backend did not, then point every gradient at its slice of
        the reduced bucket (no copy).  A parameter without a local gradient gets the mean only if some rank produced
        one (``used``); one that is unused everywhere keeps ``grad = None``."""
        if not self._avg and world > 1:
            self._flat[bi].div_(world)
        for k_i, (p, v) in enumerate(zip(self.buckets[bi], self._views[bi])):
            if id(p) in keep_none:
                p.grad = None
            elif p.grad is None:
                if used is None or used[used_off + k_i]:
                    p.grad = v
            elif p.grad is not v:
                p.grad = v

    # ------------------------------------------------------------------------------------------------ used-mask
    def _local_mask(self) -> List[int]:
        return [0 if p.grad is None else 1 for bucket in self.buckets for p in bucket]

    def _set_idle(self, m: List[bool]):
        k = 0
        for bi, bucket in enumerate(self.buckets):
            self._idle[bi] = {id(p) for j, p in enumerate(bucket) if not m[k + j]}
            k += len(bucket)

    @torch.no_grad()
    def _used_mask_sync(self) -> Optional[List[bool]]:
        """VERIFY mode: one tiny MAX all-reduce telling which parameters received a gradient on ANY rank, read back on
        the host (drains the stream).  Issued after all bucket collectives on every rank: same order everywhere."""
        if not self.track_unused:
            return None
        dev = self.params[0].device if self.params else torch.device("cpu")
        m = torch.tensor(self._local_mask(), dtype=torch.int32, device=dev)
        dist.all_reduce(m, op=dist.ReduceOp.MAX)
        m = m.cpu()
        self.host_syncs += 1
        used = [bool(x) for x in m.tolist()]
        self._set_idle(used)
        if self._last_mask is not None and used == self._last_mask:
            self._stable += 1
        else:
            self._stable = 1
        self._last_mask = used
        if self.steady_after is not None and self._stable >= self.steady_after:
            self.mode = "steady"                   # the same decision on every rank: it follows the reduced mask
            self._pending = None
        return used

    @torch.no_grad()
    def _used_mask_async(self):
        """STEADY mode: the same reduction, issued without a host wait (pinned staging both ways, an event marks the
        read-back); ``_check_pending`` looks at it one step later."""
        dev = self.params[0].device
        if dev.type != "cuda":
            m = torch.tensor(self._local_mask(), dtype=torch.int32)
            dist.all_reduce(m, op=dist.ReduceOp.MAX)
            self._pending = (m, None)
            return
        if self._pin_in is None:
            self._pin_in = torch.empty(self._n, dtype=torch.int32).pin_memory()
            self._pin_out = [torch.empty(self._n, dtype=torch.int32).pin_memory() for _ in range(2)]
            self._pin_k = 0
            self._pin_free = torch.cuda.Event()
        else:
            self._pin_free.synchronize()           # the previous step's upload has left the staging buffer (long ago)
        self._pin_in.copy_(torch.tensor(self._local_mask(), dtype=torch.int32))
        m = self._pin_in.to(dev, non_blocking=True)
        self._pin_free.record()
        dist.all_reduce(m, op=dist.ReduceOp.MAX)
        out = self._pin_out[self._pin_k]
        self._pin_k ^= 1
        out.copy_(m, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._pending = (out, ev)

    def _check_pending(self):
        """STEADY mode, at the start of ``finish()``: the PREVIOUS step's reduced mask (complete long ago -- the wait does
        not drain this step's work).  A change sends every rank back to VERIFY mode at the same step."""
        if self._pending is None:
            return
        m, ev = self._pending
        self._pending = None
        if ev is not None:
            ev.synchronize()
        used = [bool(x) for x in m.tolist()]
        if used != self._last_mask:
            import logging
            self.steady_fallbacks += 1
            logging.getLogger("dmm_net_amd.distributed").warning(
                "GradBucketer: the set of parameters that receive gradients changed (fall-back #%d); the step in between "
                "dropped / zero-filled the gradients of the parameters that changed sides; back to the per-step used-mask "
                "exchange (one host read per step) until it is stable again", self.steady_fallbacks)
            self.mode, self._stable, self._last_mask = "verify", 0, None

    # ------------------------------------------------------------------------------------------------ entry points
    @torch.no_grad()
    def _reduce_late(self, late, world):
        # the same list on every rank (it comes from the MAX-reduced used-mask and the previous idle sets, both
        # identical everywhere): one more flat collective, in parameter order
        flat = torch.cat([(g if g is not None else torch.zeros_like(p)).reshape(-1) for p, g in late])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.div_(world)
        off = 0
        for p, _ in late:
            p.grad = flat[off:off + p.numel()].view_as(p).clone()
            off += p.numel()

    @torch.no_grad()
    def _complete(self, world: int):
        """Shared tail of ``finish()`` / ``all_reduce_mean()``: every bucket collective has been issued."""
        idle_before = [set(s) for s in self._idle]
        steady = self.mode == "steady"
        if steady:
            self._used_mask_async()
            used = None
        else:
            used = self._used_mask_sync()          # after all bucket collectives on every rank: same order
        uo, late = 0, []
        for bi in range(len(self.buckets)):
            self._handles[bi].wait()
            keep_none = ()
            if steady:
                keep_none = idle_before[bi]        # assumed idle everywhere: None on every rank, whatever happened here
            elif used is not None:
                # went out as "known idle" (zeros) but got a gradient on some rank; criterion and order are the same
                # on every rank; the LOCAL gradient is taken before the views are adopted
                late += [(p, p.grad) for k_i, p in enumerate(self.buckets[bi])
                         if id(p) in idle_before[bi] and used[uo + k_i]]
            if steady and self._adopted[bi]:
                # the hook adopted every gradient that exists; what is left is to keep the known-idle parameters at None
                if not self._avg and world > 1:
                    self._flat[bi].div_(world)
                if keep_none:
                    for p in self.buckets[bi]:
                        if id(p) in keep_none and p.grad is not None:
                            p.grad = None
            else:
                self._adopt(bi, world, used, uo, keep_none)
            self._adopted[bi] = False
            uo += len(self.buckets[bi])
            self._handles[bi] = None
            self._filled[bi] = set()
            self._ready[bi] = False
            self._launched[bi] = False
        self._next = 0
        if late:
            self._reduce_late(late, world)

    @torch.no_grad()
    def finish(self):
        """After ``backward()`` with ``overlap=True``: complete (in bucket order), wait, adopt the bucket views."""
        world = dist.get_world_size()
        if self.mode == "steady":
            self._check_pending()
        for bi in range(self._next, len(self.buckets)):
            self._pack(bi)                         # missing gradients are zeros
            self._ready[bi] = True
        self._launch_ready_prefix()
        assert self._next == len(self.buckets)
        self._complete(world)

    @torch.no_grad()
    def all_reduce_mean(self):
        """grad <- mean over ranks (params whose grad is None are treated as zeros, like DDP with
        find_unused_parameters=True, train.py:181).  With ``overlap=True`` this is ``finish()``."""
        if self.overlap:
            return self.finish()
        world = dist.get_world_size()
        if self.mode == "steady":
            self._check_pending()
        for i in range(len(self.buckets)):
            self._pack(i)
            self._handles[i] = self._collective(self._flat[i])
            self.launch_log.append(i)
        self._complete(world)


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
