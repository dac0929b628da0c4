"""Multi-process path on CPU: gloo, world_size 2 (the GPU runs use the same code over RCCL)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dmm_net_amd.distributed import GradBucketer, init_from_env, reduce_loss_dict, shard_range


def test_shard_range_partitions_frames():
    for total in (0, 1, 7, 8, 1024, 1025):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0]
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    r, w = init_from_env("gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(0)
    # a small "model": several tensors, a bucket limit that forces more than one bucket, one unused parameter
    params = [torch.nn.Parameter(torch.zeros(s)) for s in ((300, 7), (5,), (1000,), (64, 64), (3,))]
    for i, p in enumerate(params[:-1]):
        p.grad = torch.full_like(p, float((rank + 1) * (i + 1)))
    params[-1].grad = None if rank == 0 else torch.ones(3)
    gb = GradBucketer(params, bucket_mb=0.01)
    assert 2 <= gb.num_collectives() < len(params) + 1
    gb.all_reduce_mean()
    ok = True
    for i, p in enumerate(params[:-1]):
        exp = (i + 1) * sum(range(1, world + 1)) / world
        ok &= bool(torch.allclose(p.grad, torch.full_like(p, exp)))
    ok &= bool(torch.allclose(params[-1].grad, torch.full((3,), (world - 1) / world)))
    red = reduce_loss_dict({"loss": torch.tensor(float(rank + 1)), "match_loss": torch.tensor(2.0 * (rank + 1))})
    if rank == 0:
        ok &= abs(float(red["loss"]) - sum(range(1, world + 1)) / world) < 1e-6
        ok &= abs(float(red["match_loss"]) - 2 * sum(range(1, world + 1)) / world) < 1e-6
    # frame sharding + a max-over-ranks timing reduction as bench.py does
    b, e = shard_range(11, rank, world)
    t = torch.tensor([float(e - b)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ok &= float(t) == 6.0
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, ok))


def test_gloo_world2_grad_bucketer_and_loss_reduce():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]


def _overlap_worker(rank, world, port, q):
    """Real backward passes: overlapped bucketer (hooks fire during backward) == plain mean of per-rank gradients."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    init_from_env("gloo")
    torch.manual_seed(0)                                      # identical initial weights on every rank
    net = torch.nn.Sequential(torch.nn.Linear(16, 64), torch.nn.ReLU(), torch.nn.Linear(64, 64), torch.nn.ReLU(),
                              torch.nn.Linear(64, 4))
    unused = torch.nn.Parameter(torch.ones(5))                # never reaches the loss (find_unused_parameters case)
    params = list(net.parameters()) + [unused]
    gb = GradBucketer(params, bucket_mb=0.004, overlap=True)
    ok = gb.num_collectives() >= 3
    ok &= gb.buckets[0][0] is unused and gb.buckets[-1][-1] is params[0]      # reverse order: last layer first
    for step in range(2):                                     # two steps: state resets between them
        x = torch.randn(8, 16, generator=torch.Generator().manual_seed(100 * step + rank))
        for p in params:
            p.grad = None
        # this rank's own gradients WITHOUT the hooks (autograd.grad does not accumulate): with bucket views a gradient may
        # already hold the reduced value when backward() returns
        local = list(torch.autograd.grad(net(x).pow(2).sum(), params[:-1])) + [None]
        net(x).pow(2).sum().backward()
        launched = sum(h is not None for h in gb._handles)
        # step 0: the idle parameter sits in bucket 0 and nothing may overtake it (fixed issue order); from step 1 on
        # it is known idle on every rank and the collectives are in flight when backward() returns
        ok &= launched == 0 if step == 0 else launched >= 1
        gb.finish()
        for p, g in zip(params, local):
            if p is unused:                                   # unused on every rank: stays None (no zero grads for Adam)
                ok &= p.grad is None
                continue
            g = torch.zeros_like(p) if g is None else g
            ref = g.clone()
            dist.all_reduce(ref)
            ok &= bool(torch.allclose(p.grad, ref / world, rtol=1e-6, atol=1e-7))
        ok &= gb.launch_log == sorted(gb.launch_log)          # fixed issue order
        gb.launch_log.clear()
    # gradient accumulation under overlap is refused, not silently dropped
    for p in params:
        p.grad = None
    x = torch.randn(8, 16)
    net(x).pow(2).sum().backward()
    try:
        net(x).pow(2).sum().backward()
        ok = False
    except RuntimeError as e:
        ok &= "finish()" in str(e)
    gb.finish()
    gb.remove_hooks()
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, bool(ok)))


def test_gloo_world2_overlapped_grad_bucketer():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_overlap_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]


def _diverging_worker(rank, world, port, q):
    """Per-rank autograd graphs DIFFER: rank 1 never uses a parameter that sits in a MIDDLE bucket, so its
    buckets complete in another order than rank 0's.  Collectives must still be issued in bucket-index order on
    both ranks (RCCL pairs by issue order) and the result must be the plain mean with zeros for the missing grad."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    init_from_env("gloo")
    torch.manual_seed(0)
    a, b, c = (torch.nn.Linear(8, 300), torch.nn.Linear(8, 500), torch.nn.Linear(8, 700))   # different bucket sizes
    params = list(a.parameters()) + list(b.parameters()) + list(c.parameters())
    gb = GradBucketer(params, bucket_mb=0.003, overlap=True)
    ok = gb.num_collectives() >= 3
    x = torch.randn(4, 8, generator=torch.Generator().manual_seed(7 + rank))
    def fwd():
        loss = a(x).sum() + c(x).pow(2).sum()
        return loss + b(x).sum() if rank == 0 else loss       # the middle layer only exists in rank 0's graph
    used = [p for p in params if rank == 0 or all(p is not q for q in b.parameters())]
    own = dict(zip(map(id, used), torch.autograd.grad(fwd(), used)))       # this rank's gradients, without the hooks
    local = [own.get(id(p)) for p in params]
    fwd().backward()
    gb.finish()
    ok &= gb.launch_log == list(range(gb.num_collectives()))
    for p, g in zip(params, local):
        ref = torch.zeros_like(p) if g is None else g.clone()
        dist.all_reduce(ref)
        ok &= p.grad is not None and bool(torch.allclose(p.grad, ref / world, rtol=1e-6, atol=1e-7))
    gb.remove_hooks()
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, bool(ok)))


def test_gloo_world2_bucket_order_is_fixed_when_graphs_differ():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_diverging_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]


def _late_grad_worker(rank, world, port, q):
    """a parameter that was idle on EVERY rank in the previous step lets its bucket go out early; if it then
    produces a gradient (on one rank only, after the bucket was issued) no rank may raise or hang, and every rank must
    end up with the plain mean."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    init_from_env("gloo")
    torch.manual_seed(0)
    l1, l2, l3 = torch.nn.Linear(16, 64), torch.nn.Linear(64, 64), torch.nn.Linear(64, 4)
    shift = torch.nn.Parameter(torch.zeros(16))               # added to the INPUT: its gradient is the last to arrive
    params = list(l1.parameters()) + list(l2.parameters()) + list(l3.parameters()) + [shift]
    gb = GradBucketer(params, bucket_mb=0.004, overlap=True)
    ok = gb.buckets[0][0] is shift                             # reverse order: it sits in the FIRST bucket to go out
    for step in range(3):
        x = torch.randn(8, 16, generator=torch.Generator().manual_seed(50 * step + rank))
        for p in params:
            p.grad = None
        use = step == 1 and rank == 1                          # idle everywhere in step 0; one rank uses it in step 1
        fwd = lambda: l3(torch.relu(l2(torch.relu(l1(x + shift if use else x))))).pow(2).sum()
        local = list(torch.autograd.grad(fwd(), params if use else params[:-1])) + ([] if use else [None])
        fwd().backward()
        if step == 1:
            ok &= gb._launched[0]                              # bucket 0 left before shift's gradient existed
        gb.finish()
        for p, g in zip(params, local):
            ref = torch.zeros_like(p) if g is None else g.clone()
            dist.all_reduce(ref)
            if p is shift and step != 1:
                ok &= p.grad is None
            else:
                ok &= p.grad is not None and bool(torch.allclose(p.grad, ref / world, rtol=1e-6, atol=1e-7))
    gb.remove_hooks()
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, bool(ok)))


def _wakeup_worker(rank, world, port, q):
    """A parameter that was idle on every rank wakes up on ONE rank, IN TIME (its gradient exists before its bucket goes
    out -- it sits in the last bucket): the bucket carries it, finish() still reduces it from the LOCAL gradients (views
    adopted early must not be reduced twice), every rank ends up with the plain mean."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    init_from_env("gloo")
    torch.manual_seed(0)
    l1, l2 = torch.nn.Linear(16, 64), torch.nn.Linear(64, 4)
    extra = torch.nn.Parameter(torch.ones(4))                  # multiplies the OUTPUT: its gradient is the first to arrive
    params = [extra] + list(l1.parameters()) + list(l2.parameters())
    gb = GradBucketer(params, bucket_mb=0.004, overlap=True, steady_after=None)
    ok = gb.buckets[-1][-1] is extra                           # reverse order: it sits in the LAST bucket to go out
    for step in range(3):
        x = torch.randn(8, 16, generator=torch.Generator().manual_seed(70 * step + rank))
        for p in params:
            p.grad = None
        use = step == 1 and rank == 0
        fwd = lambda: (l2(torch.relu(l1(x))) * (extra if use else 1.0)).pow(2).sum()
        used = params if use else params[1:]
        own = dict(zip(map(id, used), torch.autograd.grad(fwd(), used)))
        fwd().backward()
        gb.finish()
        for p in params:
            ref = own[id(p)].clone() if id(p) in own else torch.zeros_like(p)
            dist.all_reduce(ref)
            if p is extra and step != 1:
                ok &= p.grad is None
            else:
                ok &= p.grad is not None and bool(torch.allclose(p.grad, ref / world, rtol=1e-6, atol=1e-7))
    gb.remove_hooks()
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, bool(ok)))


def test_gloo_world2_idle_parameter_waking_up_in_time_on_one_rank():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_wakeup_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]


def test_gloo_world2_late_gradient_of_a_previously_idle_parameter():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_late_grad_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]


def _steady_worker(rank, world, port, q):
    """once the set of used parameters has been the same for ``steady_after`` steps the per-step
    host read of the used-mask stops (STEADY mode: the mask is reduced asynchronously and looked at one step later);
    gradients are views of the flat buckets; a change of the set is noticed one step late by EVERY rank alike, the
    parameter that woke up is dropped on every rank in that one step (never applied on one rank only), and the
    bucketer is back in the exact per-step exchange from the next step on."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    init_from_env("gloo")
    torch.manual_seed(0)
    l1, l2, l3 = torch.nn.Linear(16, 64), torch.nn.Linear(64, 64), torch.nn.Linear(64, 4)
    shift = torch.nn.Parameter(torch.zeros(16))
    params = list(l1.parameters()) + list(l2.parameters()) + list(l3.parameters()) + [shift]
    gb = GradBucketer(params, bucket_mb=0.004, overlap=True, steady_after=2)
    ok, modes, syncs = True, [], []
    for step in range(8):
        x = torch.randn(8, 16, generator=torch.Generator().manual_seed(50 * step + rank))
        if step % 2:                                           # both zero_grad flavours
            for p in params:
                p.grad = None
        else:
            for p in params:
                if p.grad is not None:
                    p.grad.zero_()
        use = step == 4 and rank == 1                          # the idle parameter wakes up on ONE rank in step 4
        fwd = lambda: l3(torch.relu(l2(torch.relu(l1(x + shift if use else x))))).pow(2).sum()
        # this rank's own gradients, taken WITHOUT the hooks: once gradients are bucket views, a bucket whose collective
        # is already in flight when backward() returns no longer holds the local values
        local = list(torch.autograd.grad(fwd(), params[:-1])) + [None]
        fwd().backward()
        before = gb.steady_fallbacks
        gb.finish()
        modes.append(gb.mode)
        syncs.append(gb.host_syncs)
        ok &= (gb.steady_fallbacks - before == 1) == (step == 5)   # the change is noticed one step late, and COUNTED
        for p, g, v in zip(params, local, [gb._views[gb._slot[id(p)][0]][gb._slot[id(p)][1]] for p in params]):
            if p is shift:
                ok &= p.grad is None                           # also in step 4, on BOTH ranks
                continue
            ref = g.clone()
            dist.all_reduce(ref)
            ok &= p.grad is v                                  # the gradient IS the bucket slice: no scatter copy
            ok &= bool(torch.allclose(p.grad, ref / world, rtol=1e-6, atol=1e-7))
    # steps 0, 1 verify (2 identical masks) -> steady from step 2; step 5 sees step 4's mask -> verify again for 5, 6
    ok &= modes == ["verify", "steady", "steady", "steady", "steady", "verify", "steady", "steady"]
    ok &= syncs == [1, 2, 2, 2, 2, 3, 4, 4]
    gb.remove_hooks()
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, bool(ok), modes, syncs))


def test_gloo_world2_steady_mode_takes_the_host_read_off_the_step_and_heals():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_steady_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[:2] for r in res) == [(0, True), (1, True)], res


def _layout_worker(rank, world, port, q):
    """A rank that builds its bucketer from a different parameter list (here: another bucket size on rank 1, then another
    parameter order) must be caught at construction, on EVERY rank, by one all_gather -- not by a hang in the first
    per-bucket collective."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    init_from_env("gloo")
    params = [torch.nn.Parameter(torch.zeros(s)) for s in ((300, 7), (5,), (1000,), (64, 64))]
    ok = True
    same = GradBucketer(params, bucket_mb=0.01)
    ok &= same.layout_verified
    for kw, plist in ((dict(bucket_mb=0.01 if rank == 0 else 0.02), params),
                      (dict(bucket_mb=0.01), params if rank == 0 else params[::-1]),
                      (dict(bucket_mb=0.01, overlap=(rank == 1)), params)):
        try:
            GradBucketer(plist, **kw)
            ok = False
        except RuntimeError as e:
            ok &= "differs between ranks" in str(e) and "[1]" in str(e)
    quiet = GradBucketer(params if rank == 0 else params[::-1], bucket_mb=0.01, verify_layout=False)   # opt-out: no collective
    ok &= not quiet.layout_verified
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, bool(ok)))


def test_gloo_world2_a_bucket_layout_mismatch_fails_loudly_on_every_rank():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_layout_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]


def test_host_thread_pinning_is_silent_without_a_gpu_topology():
    """``bind_host_threads_to_gpu``: no visible GPU / sysfs topology -> (None, no-op restore), masks untouched."""
    import os
    from dmm_net_amd.distributed import bind_host_threads_to_gpu
    before = os.sched_getaffinity(0)
    node, restore = bind_host_threads_to_gpu(0)
    assert node is None
    restore()
    assert os.sched_getaffinity(0) == before
