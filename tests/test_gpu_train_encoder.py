"""The bf16 channels-last TRAINING form of the encoder (``train_encoder.TrainEncoder``, BASELINE configs[3]) on the GPU:
the two-launch BatchNorm kernels against torch's fp32 BatchNorm on the same bf16 values, the HIP-graph plumbing against the
plain fp32 ``FeatureEncoder`` (fp32 mode: same arithmetic, so a tight bound -- and REPLAYS, not only the capturing run), the
bf16 step against the fp32 step (bounded, achieved values recorded), and the gradient hand-over to ``GradBucketer``."""
import copy
import math

import pytest
import torch
import torch.nn.functional as F

from dmm_net_amd import _lib
from dmm_net_amd.encoder import FeatureEncoder
from dmm_net_amd.train_encoder import TrainEncoder, _BNActFn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _bn_reference(x, w, b, relu, res, dy, eps=1e-5):
    """torch's training-mode BatchNorm (+ residual) (+ ReLU) in fp32 on the SAME bf16 values, and its gradients."""
    xf = x.float().requires_grad_(True)
    wf, bf = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    rf = res.float().requires_grad_(True) if res is not None else None
    rm, rv = torch.zeros_like(w), torch.ones_like(w)
    y = F.batch_norm(xf, rm, rv, wf, bf, True, 0.1, eps)
    if rf is not None:
        y = y + rf
    if relu:
        y = F.relu(y)
    y.backward(dy.float())
    return y.detach(), xf.grad, wf.grad, bf.grad, (rf.grad if rf is not None else None), rm, rv


@pytest.mark.parametrize("B,C,H,W", [(12, 64, 32, 56), (3, 256, 17, 23), (2, 2048, 8, 14), (5, 32, 9, 7), (1, 1024, 3, 5),
                                     (4, 128, 64, 112), (2, 512, 1, 1)])
@pytest.mark.parametrize("relu,has_res", [(True, False), (True, True), (False, False), (False, True)])
def test_bn_kernels_against_torch_fp32(B, C, H, W, relu, has_res):
    from conftest import record_achieved
    g = torch.Generator(device=DEV).manual_seed(C * 7 + H)
    cl = torch.channels_last
    x = (torch.randn((B, C, H, W), generator=g, device=DEV) * 1.7 + 0.4).to(torch.bfloat16).contiguous(memory_format=cl)
    res = torch.randn((B, C, H, W), generator=g, device=DEV).to(torch.bfloat16).contiguous(memory_format=cl) if has_res else None
    dy = torch.randn((B, C, H, W), generator=g, device=DEV).to(torch.bfloat16).contiguous(memory_format=cl)
    w = torch.rand(C, generator=g, device=DEV) + 0.5
    b = torch.randn(C, generator=g, device=DEV) * 0.3
    rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    xg = x.clone().requires_grad_(True)
    wg, bg = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    rg = res.clone().requires_grad_(True) if has_res else None
    y = _BNActFn.apply(xg, wg, bg, rm, rv, 0.1, 1e-5, relu, rg)
    y.backward(dy)
    ry, rdx, rdw, rdb, rdres, rrm, rrv = _bn_reference(x, w, b, relu, res, dy)
    ulp = 2.0 ** -8                                               # bf16: 8 significant bits
    tol = lambda ref: ulp * ref.abs() + 1e-3 * float(ref.abs().max())
    assert bool(((y.float() - ry).abs() <= tol(ry)).all())
    if B * H * W > 1:
        assert float((rm - rrm).abs().max()) <= 1e-4 and float((rv - rrv).abs().max()) <= 1e-3 * float(rrv.abs().max())
    # where the reference's output is within a rounding of zero the mask may differ: compare gradients where both agree
    agree = ((y > 0) == (ry.to(torch.bfloat16) > 0)) if relu else torch.ones_like(y, dtype=torch.bool)
    assert float(agree.float().mean()) > 0.999
    scale = float(rdx.abs().max()) or 1.0
    err_dx = float(((xg.grad.float() - rdx).abs() * agree).max()) / scale
    assert err_dx <= 2e-2, err_dx
    rel = lambda a, r: float((a - r).abs().max()) / (float(r.abs().max()) or 1.0)
    assert rel(wg.grad, rdw) <= 5e-3 and rel(bg.grad, rdb) <= 5e-3, (rel(wg.grad, rdw), rel(bg.grad, rdb))
    if has_res:
        assert bool(((rg.grad.float() - rdres).abs() * agree).max() == 0)
    record_achieved(f"bn_train/{B}x{C}x{H}x{W}_relu{int(relu)}_res{int(has_res)}/dx_rel", err_dx)


def test_bn_entries_refuse_what_the_kernels_do_not_take():
    L = _lib.load()
    t = torch.zeros(64, device=DEV)
    s = torch.cuda.current_stream().cuda_stream
    assert L.dmm_bn_stats_bf16(t.data_ptr(), 4, 12, t.data_ptr(), s) == 2          # C % 8 != 0
    assert L.dmm_bn_stats_bf16(t.data_ptr(), 4, 24, t.data_ptr(), s) == 2          # 256 % (C / 8) != 0
    assert L.dmm_bn_stats_bf16(None, 4, 64, t.data_ptr(), s) == 1
    assert L.dmm_bn_stats_bf16(None, 0, 64, None, s) == 0


def _grads(m):
    return {n: (None if p.grad is None else p.grad.detach().clone()) for n, p in m.named_parameters()}


def _rel(a, b):
    num = sum(float((a[k] - b[k]).square().sum()) for k in b if b[k] is not None)
    den = sum(float(b[k].square().sum()) for k in b if b[k] is not None)
    return math.sqrt(num / max(den, 1e-30))


def _target(p, k):
    # a fixed pseudo-random direction per output (NOT sum(p^2): the heads end in a BatchNorm, whose output has a fixed mean
    # and variance per channel -- the gradient of sum(p^2) through it is zero up to eps and a comparison of it is noise)
    i = torch.arange(p.numel(), device=p.device, dtype=torch.float32).view(p.shape)
    return torch.sin(i * 0.37 + k)


def _loss(f, skips=True):
    outs = f["backbone_feature"] + (f["refine_input_feat"] if skips else ())
    return sum((p.float() * _target(p, k)).mean() for k, p in enumerate(outs))


@pytest.mark.parametrize("model", ["resnet34", "resnet50"])
def test_graphed_fp32_mode_equals_the_plain_encoder_on_replays(model):
    """``TrainEncoder(dtype=float32)`` computes what ``FeatureEncoder`` computes (stock BatchNorm, fp32 GEMMs): outputs,
    every parameter gradient and the BatchNorm buffers must agree over THREE steps with different images -- the first call
    captures, the later ones replay (a mis-ordered memset node or a stale static buffer shows up only there)."""
    torch.manual_seed(3)
    ref = FeatureEncoder(model, hidden_size=32).to(DEV).train()
    enc = copy.deepcopy(ref)
    te = TrainEncoder(enc, dtype=torch.float32)
    for step in range(3):
        img = torch.randn(4, 3, 96, 128, device=DEV)
        for m in (ref, enc):
            m.zero_grad(set_to_none=True)
        fr, ft = ref(img), te(img)
        for a, b in zip(ft["backbone_feature"] + ft["refine_input_feat"], fr["backbone_feature"] + fr["refine_input_feat"]):
            assert float((a.float() - b).abs().max()) <= 2e-3 * float(b.abs().max()), step
        _loss(fr).backward()
        _loss(ft).backward()
        gr, gt = _grads(ref), _grads(enc)
        assert all((gr[k] is None) == (gt[k] is None) for k in gr), step
        assert _rel(gt, gr) <= 5e-3, (step, _rel(gt, gr))
        for (n, a), (_, b) in zip(enc.named_buffers(), ref.named_buffers()):
            assert float((a.float() - b.float()).abs().max()) <= 1e-3 * (float(b.float().abs().max()) + 1.0), (step, n)


def test_graphed_bf16_step_equals_the_eager_bf16_step_and_tracks_fp32():
    """The shipped form (bf16, fused BatchNorm, GEMM 1x1s, graphs) against (a) the same functions run eagerly -- same
    arithmetic up to the arrival order of the statistics' atomics -- over replays, and (b) the fp32 encoder: the loss within
    2 %, the gradients of the heads (a few bf16 layers from the loss) within 10 %, all gradients positively aligned."""
    from conftest import record_achieved
    torch.manual_seed(5)
    ref = FeatureEncoder("resnet50").to(DEV).train()
    a, b = copy.deepcopy(ref), copy.deepcopy(ref)
    graph, eager = TrainEncoder(a, skips_need_grad=False), TrainEncoder(b, graphs=False, skips_need_grad=False)
    loss = lambda f: _loss(f, skips=False)
    for step in range(3):
        img = torch.randn(6, 3, 128, 224, device=DEV)
        for m in (ref, a, b):
            m.zero_grad(set_to_none=True)
        lg, le, lr = loss(graph(img)), loss(eager(img)), loss(ref(img))
        lg.backward(), le.backward(), lr.backward()
        gg, ge, gr = _grads(a), _grads(b), _grads(ref)
        assert all((gg[k] is None) == (ge[k] is None) for k in ge)
        assert all(gg[k] is None for k in gg if k.startswith(("sk", "bn")))       # skips_need_grad=False: like autograd
        e1 = _rel(gg, ge)
        assert e1 <= 0.05, (step, e1)
        assert abs(float(lg) - float(lr)) <= 0.02 * abs(float(lr)), (float(lg), float(lr))
        heads = lambda g: {k: v for k, v in g.items() if k.startswith("prop") and v is not None}
        e2 = _rel(heads(gg), heads(gr))
        dot = sum(float((gg[k] * gr[k]).sum()) for k in gr if gr[k] is not None and gg[k] is not None)
        cos = dot / math.sqrt(sum(float(gg[k].square().sum()) for k in gg if gg[k] is not None) *
                              sum(float(gr[k].square().sum()) for k in gr if gr[k] is not None and gg[k] is not None))
        assert e2 <= 0.10 and cos >= 0.5, (step, e2, cos)
        record_achieved(f"train_encoder/step{step}/graph_vs_eager_rel", e1)
        record_achieved(f"train_encoder/step{step}/heads_grad_vs_fp32_rel", e2)
        record_achieved(f"train_encoder/step{step}/all_grad_vs_fp32_cos", cos)


def test_gradient_hand_over_feeds_the_bucketer_and_accumulates():
    """Parameter gradients are handed over like AccumulateGrad does: ``p.grad`` set (or added to) and the post-accumulate
    hooks called, segment by segment, last segment first -- ``GradBucketer(overlap=True)`` launches its buckets in order and
    the reduced gradients (world 1: unchanged) equal a plain backward's; a second backward accumulates."""
    import os
    import torch.distributed as dist
    from dmm_net_amd.distributed import GradBucketer
    torch.manual_seed(9)
    enc = FeatureEncoder("resnet34", hidden_size=32).to(DEV).train()
    te = TrainEncoder(enc, skips_need_grad=False)
    img = torch.randn(2, 3, 96, 128, device=DEV)
    loss = lambda f: _loss(f, skips=False)
    img2 = torch.randn(2, 3, 96, 128, device=DEV)
    loss(te(img2)).backward()
    other = _grads(enc)
    enc.zero_grad(set_to_none=True)
    loss(te(img)).backward()
    plain = _grads(enc)
    loss(te(img2)).backward()                                     # accumulation: the first gradient aliased a static buffer
    both = _grads(enc)
    assert _rel(both, {k: (None if v is None else v + other[k]) for k, v in plain.items()}) <= 2e-2
    own = not dist.is_initialized()
    if own:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29547")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        params = [p for p in enc.parameters() if p.requires_grad]
        seen = []
        for p in params:
            p.register_post_accumulate_grad_hook(lambda q: seen.append(id(q)))
        bucketer = GradBucketer(params, bucket_mb=4.0, overlap=True)
        enc.zero_grad(set_to_none=True)
        loss(te(img)).backward()
        bucketer.finish()
        got = _grads(enc)
        assert bucketer.launch_log == sorted(bucketer.launch_log) and len(bucketer.launch_log) == bucketer.num_collectives()
        assert _rel(got, plain) <= 2e-2
        used = {id(p) for p in params if p.grad is not None}
        assert set(seen) == used and len(seen) == len(used)        # one hook call per parameter that got a gradient
        bucketer.remove_hooks()
    finally:
        if own:
            dist.destroy_process_group()
