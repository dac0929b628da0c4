"""Multi-GPU plumbing ON THE DEVICE: RCCL ("nccl" backend) is initialised on cuda:0 and the
bucketed gradient mean runs over a real ResNet-101 encoder's device gradients; BASELINE config 4's per-GPU share
(4 videos x clip 3 of 255x448, ResNet-101, 50 proposals, 5 template slots, forward + backward + Adam through
DMM_Model) is a test, not a tool; the multi-rank control flow of bench.py is launched through torch.distributed.run.

A 1-GPU box cannot host two RCCL ranks (RCCL refuses two ranks on one device), so the collective itself runs at
world_size 1 here -- the launch sequence, stream semantics and bucket layout are the ones the 8-GPU job uses -- and
the 2-rank numerics stay covered by the gloo tests (tests/test_distributed_gloo.py)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from dmm_net_amd import synth
from dmm_net_amd.distributed import GradBucketer, init_from_env

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.fixture
def rccl_world1():
    import torch.distributed as dist
    old = {k: os.environ.get(k) for k in ("MASTER_ADDR", "MASTER_PORT", "RANK", "WORLD_SIZE")}
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1")
    torch.cuda.set_device(0)
    rank, world = init_from_env("nccl", torch.device(DEV))
    assert (rank, world) == (0, 1) and dist.get_backend() == "nccl"
    yield dist
    dist.destroy_process_group()
    for k, v in old.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


def test_rccl_grad_bucketer_over_resnet101_gradients(rccl_world1):
    """train.py:178-184 + :62-68: the gradient mean of the encoder (ResNet-101 body + sk/prop heads, > 200 MB of fp32
    gradients) as a few bucketed RCCL all-reduces issued from autograd hooks during backward, on device tensors."""
    from dmm_net_amd.encoder import FeatureEncoder
    dist = rccl_world1
    torch.manual_seed(0)
    enc = FeatureEncoder("resnet101").to(DEV).train()
    params = list(enc.get_skip_params()) + list(enc.get_backbone_para())
    nbytes = sum(p.numel() * 4 for p in params)
    assert nbytes > 200e6, nbytes                                    # 44.5 M body + 8.2 M heads
    x = torch.randn(2, 3, 128, 160, device=DEV)

    def loss_of(f):
        return sum(t.float().pow(2).mean() for t in f["backbone_feature"]) + \
            sum(t.float().abs().mean() for t in f["refine_input_feat"])

    # a plain backward first: which parameters get a gradient at all.  (Values are compared within ONE backward below:
    # MIOpen's weight-gradient kernels accumulate with atomics and a random-init 101-layer net in train-mode BatchNorm at
    # batch 2 amplifies that to ~1e-2 relative between two backward passes.)
    loss_of(enc(x)).backward()
    ref = [None if p.grad is None else p.grad.clone() for p in params]
    idle = [p for p, g in zip(params, ref) if g is None]
    assert {id(p) for p in idle} == {id(p) for p in enc.base.fc.parameters()}   # the ResNet's unused classifier
    # the same through the overlapped bucketer over RCCL, two steps (step 2 overlaps: the idle fc is known by then)
    gb = GradBucketer(params, bucket_mb=64.0, overlap=True)
    assert 3 <= gb.num_collectives() <= 5
    for step in range(2):
        for p in params:
            p.grad = None
        gb.launch_log.clear()
        loss_of(enc(x)).backward()
        in_flight = sum(h is not None for h in gb._handles)
        local = [None if p.grad is None else p.grad.clone() for p in params]   # what autograd produced this step
        gb.finish()
        torch.cuda.synchronize()
        assert gb.launch_log == list(range(gb.num_collectives()))    # one all-reduce per bucket, fixed order
        if step == 1:
            assert in_flight == gb.num_collectives()                  # all issued under backward()
        for p, g, r in zip(params, local, ref):
            if g is None:
                assert r is None and p.grad is None                   # idle everywhere: no zero gradient for Adam
            else:
                assert p.grad.is_cuda and torch.equal(p.grad, g)      # world 1: mean == the local gradient, bit exact
    gb.remove_hooks()
    # non-overlapped path + the loss-dict reduce + a device barrier as bench.py's fence
    local = [None if p.grad is None else p.grad.clone() for p in params]
    gb2 = GradBucketer(params, bucket_mb=64.0, overlap=False)
    gb2.all_reduce_mean()
    for p, g in zip(params, local):
        assert (p.grad is None) if g is None else torch.equal(p.grad, g)
    t = torch.tensor([3.5], dtype=torch.float64, device=DEV)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.barrier()
    assert float(t) == 3.5


def _config4_inputs(B, F, P, H, W, g):
    from dmm_net_amd.proposals import SimpleBoxList

    def boxes(n):
        x1 = torch.rand(n, generator=g, device=DEV) * (W - 60)
        y1 = torch.rand(n, generator=g, device=DEV) * (H - 60)
        return torch.stack([x1, y1, x1 + 10 + torch.rand(n, generator=g, device=DEV) * 150,
                            y1 + 10 + torch.rand(n, generator=g, device=DEV) * 100], 1).clamp(max=W - 1)
    props = []
    for b in range(B):
        bl = SimpleBoxList(boxes(P), (W, H))
        bl.add_field("mask", torch.rand((P, 1, H, W), generator=g, device=DEV))
        bl.add_field("scores", torch.rand(P, generator=g, device=DEV))
        props.append(bl)
    return props, [SimpleBoxList(boxes(F), (W, H)) for _ in range(B)]


@pytest.mark.parametrize("form", ["fp32", "bf16"])
def test_config4_per_gpu_share_trains_through_dmm_model(rccl_world1, form):
    """(``form`` = "bf16": the same step with the encoder as ``train_encoder.TrainEncoder`` -- bf16 channels-last, HIP-graph
    replays, three forwards in flight per backward, its gradient hand-over feeding the same bucketer.)
    BASELINE configs[3], the share of ONE GPU (scripts/train/train_101.sh:12-17: batch 4 videos, clip 3, 255x448,
    ResNet-101): per frame encoder -> ROI features -> DMM_Model.forward (ragged batched HIP layer with targets) ->
    soft-IoU + match loss; backward through the HIP layer, the ROI scatter and MIOpen; bucketed RCCL gradient mean
    issued under backward; Adam.  The loss must go down on a fixed batch, and the ragged batched step must equal
    per-video calls."""
    from dmm_net_amd.dmm_model import DMM_Model
    from dmm_net_amd.encoder import FeatureEncoder
    from dmm_net_amd.roi_features import FeatureExtractor
    NV, T, F, P, H, W = 4, 3, 5, 50, 255, 448
    torch.manual_seed(0)
    g = torch.Generator(device=DEV).manual_seed(4)
    enc = FeatureEncoder("resnet101").to(DEV).train()
    run_enc = enc
    if form == "bf16":
        from dmm_net_amd.train_encoder import TrainEncoder
        run_enc = TrainEncoder(enc, skips_need_grad=False)
    cfgs = {"matching": {"algo": "relax"}, "relax_max_iter": 10, "relax_proj_iter": 5, "relax_learning_rate": 0.1,
            "score_weight": 0.3}
    model = DMM_Model(cfgs, is_test=0, feature_extractor=FeatureExtractor())
    params = list(enc.get_skip_params()) + list(enc.get_backbone_para())
    opt = torch.optim.Adam(params, lr=1e-4)
    gb = GradBucketer(params, bucket_mb=64.0, overlap=True)
    frames = torch.randn(NV, T, 3, H, W, device=DEV)
    n_tplt = [5, 3, 0, 2]                                             # ragged: one video without live templates
    valid = torch.zeros(NV, F, device=DEV)
    for b, o in enumerate(n_tplt):
        valid[b, :o] = 1
    targets = (torch.rand((NV, T, F, H, W), generator=g, device=DEV) > 0.6).float() * valid[:, None, :, None, None]
    per_frame = [_config4_inputs(NV, F, P, H, W, g) for _ in range(T)]

    def clip_loss(check_ragged=False):
        total, tplt = 0.0, None
        mask_last = targets[:, 0]
        for t in range(T):
            props, tboxes = per_frame[t]
            feats = run_enc(frames[:, t])
            if t == 0:
                tplt = model.fill_template_dict(None, tboxes, feats, None, valid)        # trainer.py:308-327
            out, _, match_loss, last = model(None, props, feats["backbone_feature"], mask_last, tplt, valid,
                                             targets[:, t])
            assert out.shape == (NV, F, H, W) and len(match_loss) == NV
            assert float(out[2].abs().max()) == 0.0 and torch.equal(last[2], mask_last[2])   # O == 0: zeros / carry over
            if check_ragged and t == 1:
                with torch.no_grad():
                    for b in range(NV):
                        one, _, ml1, _ = model(None, [props[b]], tuple(f[b:b + 1] for f in feats["backbone_feature"]),
                                               mask_last[b:b + 1], {0: tplt[b]}, valid[b:b + 1], targets[b:b + 1, t])
                        assert torch.allclose(one[0], out[b], rtol=0, atol=1e-6), b
                        assert abs(float(ml1[0]) - float(match_loss[b])) < 1e-6, b
            tg = targets[:, t]
            inter = (out * tg).flatten(2).sum(2)
            union = (out + tg - out * tg).flatten(2).sum(2)
            soft = (1.0 - inter / (union + 1e-6)) * valid                                # trainer.py:205-208
            total = total + soft.sum() / valid.sum() + sum(match_loss) / NV
            mask_last = last.detach()
        return total / T

    losses = []
    for step in range(4):
        opt.zero_grad(set_to_none=True)
        gb.launch_log.clear()
        loss = clip_loss(check_ragged=(step == 0))
        loss.backward()
        gb.finish()                                                   # RCCL mean of > 200 MB of gradients
        assert gb.launch_log == list(range(gb.num_collectives()))
        gnorm = sum(float(p.grad.float().pow(2).sum()) for p in params if p.grad is not None) ** 0.5
        assert np.isfinite(gnorm) and gnorm > 0
        opt.step()
        losses.append(float(loss))
    gb.remove_hooks()
    assert all(np.isfinite(losses)), losses
    assert losses[-1] < losses[0], losses


def test_bench_multi_rank_control_flow_is_launched_like_the_driver():
    """bench.py under ``python -m torch.distributed.run --nproc-per-node 2`` exactly as the driver launches it for
    N > 1 (barrier + synchronize fences, MAX over ranks, rank 0 prints one JSON line); ``--backend gloo`` because two
    RCCL ranks cannot share this box's single GPU."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps",
           "3", "--warmup", "1", "--frames", "128", "--backend", "gloo", "--no-extras"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                          # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["scaling"] == "weak" and out["unit"] == "frames/s"
    assert out["config"]["frames_per_gpu_per_step"] == 128
    assert abs(out["value"] - 2 * 128 * 3 / (out["ms_per_step"] * 3e-3)) <= 1e-3 * out["value"]
    assert 0 < out["roofline"]["frac"] < 1.0


@pytest.mark.parametrize("form", [[], ["--bf16"]])
def test_bench_config4_two_ranks_train_with_the_bucketed_gradient_mean(form):
    """(``--bf16``: the same rehearsal with ``TrainEncoder`` handing its gradients to the bucketer on both ranks: the layout
    check at construction, the hub-driven hand-over and the overlapped collectives with a real peer.)
    ``bench.py --config 4 --gpus 2`` launched like the driver would for the 8-GPU line (BASELINE configs[3]): every
    rank trains its own frames through DMM_Model, gradients are averaged by the overlapped bucketer, rank 0 prints one
    JSON line with the all-reduce accounting.  gloo, because two RCCL ranks cannot share this box's one GPU."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--config", "4", "--gpus", "2",
           "--steps", "2", "--warmup", "1", "--frames", "2", "--backend", "gloo", "--settle", "3"] + form
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    gm = out["config"]["gradient_mean"]
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["config"]["frames_per_gpu_per_step"] == 2
    assert gm["buckets"] >= 3 and gm["bytes"] > 150e6 and gm["allreduce_alone_ms"] > 0
    assert gm["used_mask_mode"] == "steady" and gm["host_reads_in_timed_steps"] == 0 and gm["gradients_are_bucket_views"]
    assert out["per_rank"]["backend"] == "gloo" and out["rccl_ranks"] == 0
    assert abs(gm["busbw_GBps"] - gm["algbw_GBps"]) <= 0.06 * gm["algbw_GBps"] + 0.1        # 2 (N-1) / N = 1 at N = 2
    assert abs(out["value"] - 2 * 2 / (out["ms_per_step"] * 1e-3)) <= 1e-2 * out["value"] + 0.06   # (value has one decimal)


def test_bench_plain_invocation_with_gpus_2_launches_two_ranks_itself():
    """``python bench.py --gpus 2`` with NO launcher around it (WORLD_SIZE unset) used to run on one
    GPU and print ``"n_gpus": 1``.  It now re-executes itself under torch.distributed.run; the line says 2 ranks, and a
    world size other than --gpus is an error instead of a number."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR",
                                                             "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--frames",
           "128", "--backend", "gloo", "--no-extras"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["frames_per_gpu_per_step"] == 128
    assert out["per_rank"]["frames_per_s_min"] <= out["per_rank"]["frames_per_s_max"]
    assert 0 < out["per_rank"]["roofline_frac_min"] <= out["per_rank"]["roofline_frac_max"] < 1
    # over RCCL two ranks cannot share this box's one GPU: refused loudly, no 1-GPU number
    r = subprocess.run(cmd[:-3] + ["--no-extras"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert "visible GPUs" in r.stderr
    # a launcher that formed another world size than --gpus: an error too
    r = subprocess.run(cmd[:2] + ["--gpus", "1", "--no-extras"], cwd=ROOT, env=dict(env, WORLD_SIZE="2", RANK="0"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


@pytest.mark.gpu
@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs in one process (the nn.DataParallel caller)")
def test_two_devices_two_threads_like_nn_dataparallel():
    """The single-process multi-device caller (reference train.py:185-186 / eval.py:64-65: ``nn.DataParallel`` runs one host
    thread per GPU through the same module): two threads, each on ITS device, drive the drop-in's evaluator call, the fused
    training call (forward + backward) and the batched fused forward at the same time; every result equals what the same
    device computes alone.  Workspaces are keyed (device, stream); the launchers run under the tensors' device."""
    import threading
    import numpy as np
    from dmm_net_amd import autograd, ops, synth
    from dmm_net_amd.match_model import MatchModel
    cfgs = {"matching": {"algo": "relax"}, "relax_max_iter": 10, "relax_proj_iter": 5, "relax_learning_rate": 0.1,
            "score_weight": 0.3}
    frs = [synth.make_frame(30 + 5 * k, 4 + k, 48, 56, 512, seed=70 + k, kind="structured", with_targets=True) for k in range(2)]

    def run_all(k, dev):
        fr = frs[k]
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        with torch.cuda.device(dev):
            with torch.no_grad():
                ev = MatchModel(cfgs, is_test=1)(t(fr.proposed_feature), t(fr.proposed_mask), [t(fr.template_feature)],
                                                 t(fr.mask_last_occurence), t(fr.proposal_score))
            pf = t(fr.proposed_feature).requires_grad_(True)
            tf = t(fr.template_feature).requires_grad_(True)
            fo, ms, ds, _, loss = MatchModel(cfgs, is_test=0)(pf, t(fr.proposed_mask), [tf], t(fr.mask_last_occurence),
                                                             t(fr.proposal_score), t(fr.targets))
            (fo.sum() + ms.sum() + 2.0 * loss["cost_loss"]).backward()
            bat = ops.match_forward(t(fr.proposed_mask)[None].repeat(3, 1, 1, 1), t(fr.mask_last_occurence)[None].repeat(3, 1, 1, 1),
                                    t(fr.proposed_feature)[None].repeat(3, 1, 1), t(fr.template_feature)[None].repeat(3, 1, 1),
                                    t(fr.proposal_score)[None].repeat(3, 1), score_weight=0.3, max_iter=10, proj_iter=5,
                                    lr=0.1, is_test=1)
            torch.cuda.synchronize(dev)
            return [x.detach().cpu().numpy() for x in (ev[0], ev[1], ev[2], fo, ms, ds, loss["cost_loss"], pf.grad, tf.grad,
                                                       bat[0], bat[1], bat[3])]
    devs = [torch.device("cuda", 0), torch.device("cuda", 1)]
    alone = [run_all(k, devs[k]) for k in range(2)]
    got, errors = [None, None], []

    def worker(k):
        try:
            for _ in range(4):
                got[k] = run_all(k, devs[k])
        except Exception as e:                                    # noqa: BLE001
            errors.append((k, repr(e)))
    threads = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    for k in range(2):
        for i, (a, b) in enumerate(zip(got[k], alone[k])):
            if i in (7, 8):                                       # gradients: float atomics across workgroups (arrival order)
                assert float(np.abs(a - b).max()) <= 1e-5 * (float(np.abs(b).max()) or 1.0), (k, i)
            else:
                assert np.array_equal(a, b), (k, i)


def test_host_threads_pin_to_the_gpus_numa_node_and_back():
    """``distributed.bind_host_threads_to_gpu``: every thread of the process lands on the CPUs local to the GPU (when the
    box shows the topology) and ``restore()`` gives the old masks back."""
    import os
    import threading
    from dmm_net_amd.distributed import bind_host_threads_to_gpu, gpu_local_cpus
    node, cpus = gpu_local_cpus(0)
    stop = threading.Event()
    worker_tid = []
    t = threading.Thread(target=lambda: (worker_tid.append(threading.get_native_id()), stop.wait(30)))
    t.start()
    while not worker_tid:
        pass
    before = {tid: os.sched_getaffinity(tid) for tid in (0, worker_tid[0])}
    got, restore = bind_host_threads_to_gpu(0)
    try:
        if node is None:
            assert got is None
        else:
            assert got == node
            for tid in before:
                assert os.sched_getaffinity(tid) <= cpus and os.sched_getaffinity(tid)
    finally:
        restore()
        for tid, m in before.items():
            assert os.sched_getaffinity(tid) == m
        stop.set()
        t.join()
