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


@pytest.mark.parametrize("R,co,ci", [(5376, 1024, 256), (86016, 256, 64), (1344, 2048, 512), (37, 64, 64), (1, 64, 128),
                                     (21504, 128, 512), (4099, 192, 320), (86016, 64, 256), (777, 64, 192), (300, 384, 64)])
def test_wgrad_1x1_against_the_fp32_product(R, co, ci):
    """dmm_wgrad_bf16: dW = dY^T X of bf16 operands with fp32 accumulation (products of bf16 values are exact in fp32: what
    differs from torch's fp32 product is the order of the sums) -- incl. row counts that are not a multiple of the 16-row
    step or of the slabs, strided operands (ldy / ldx > width); the result overwrites dW and is reproducible bit for bit."""
    from conftest import record_achieved
    L = _lib.load()
    g = torch.Generator(device=DEV).manual_seed(R + co)
    dyb = torch.randn((R, co + 64), generator=g, device=DEV).to(torch.bfloat16)
    xb = torch.randn((R, ci + 128), generator=g, device=DEV).to(torch.bfloat16)
    dy, x = dyb[:, :co], xb[:, 64:64 + ci]
    s = torch.cuda.current_stream().cuda_stream
    outs = []
    for fill in (float("nan"), 7.0):                              # dW is OVERWRITTEN whatever it (or the workspace) held
        dw = torch.full((co, ci), fill, device=DEV)
        ws = torch.full((max(int(L.dmm_wgrad_workspace_bytes(R, co, ci)), 16) // 4,), fill, device=DEV)
        rc = L.dmm_wgrad_bf16(dy.data_ptr(), x.data_ptr(), R, co, ci, dyb.stride(0), xb.stride(0), dw.data_ptr(),
                              ws.data_ptr(), ws.numel() * 4, s)
        assert rc == 0
        outs.append(dw)
    assert torch.equal(outs[0], outs[1])                          # no atomics: two runs agree bit for bit
    ref = dy.double().t() @ x.double()
    err = float((outs[0].double() - ref).abs().max()) / max(float(ref.abs().max()), 1.0)
    assert err <= 1e-5, err
    record_achieved(f"wgrad1x1/{R}x{co}x{ci}/rel", err)
    assert L.dmm_wgrad_bf16(dy.data_ptr(), x.data_ptr(), R, 96, ci, dyb.stride(0), xb.stride(0), dw.data_ptr(), None, 0, s) == 2
    if int(L.dmm_wgrad_workspace_bytes(R, co, ci)) > 16:
        assert L.dmm_wgrad_bf16(dy.data_ptr(), x.data_ptr(), R, co, ci, dyb.stride(0), xb.stride(0), dw.data_ptr(),
                                ws.data_ptr(), 16, s) in (0, 4)  # 4 = workspace too small (0: this shape needs none)


@pytest.mark.parametrize("B,H,W,ci,co,stride", [(2, 17, 23, 64, 128, 1), (3, 16, 28, 256, 256, 1), (2, 33, 56, 128, 128, 2),
                                                (12, 8, 14, 512, 512, 1), (1, 1, 1, 64, 64, 1), (2, 5, 3, 64, 64, 2),
                                                (12, 64, 112, 64, 64, 1), (3, 19, 31, 128, 64, 1), (2, 33, 41, 64, 64, 2)])
def test_wgrad_3x3_against_torch_fp32(B, H, W, ci, co, stride):
    """dmm_wgrad3x3_bf16 (implicit patch matrix) against torch's fp32 weight gradient of the same bf16 values: borders (taps
    that fall outside the image), stride 2 with odd sizes, images of one pixel."""
    from conftest import record_achieved
    L = _lib.load()
    g = torch.Generator(device=DEV).manual_seed(H * 100 + W)
    cl = torch.channels_last
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    x = torch.randn((B, ci, H, W), generator=g, device=DEV).to(torch.bfloat16).contiguous(memory_format=cl)
    dy = torch.randn((B, co, Ho, Wo), generator=g, device=DEV).to(torch.bfloat16).contiguous(memory_format=cl)
    dw = torch.full((co, ci, 3, 3), float("nan"), device=DEV)     # overwritten, in the parameter's own layout
    ws = torch.empty((max(int(L.dmm_wgrad_workspace_bytes(B * Ho * Wo, co, 9 * ci)), 16),), dtype=torch.uint8, device=DEV)
    assert L.dmm_wgrad3x3_bf16(dy.data_ptr(), x.data_ptr(), B, H, W, ci, co, stride, dw.data_ptr(), ws.data_ptr(), ws.numel(),
                               torch.cuda.current_stream().cuda_stream) == 0
    ref = torch.nn.grad.conv2d_weight(x.float().contiguous(), (co, ci, 3, 3), dy.float().contiguous(), stride=stride, padding=1)
    got = dw
    err = float((got - ref).abs().max()) / max(float(ref.abs().max()), 1.0)
    assert err <= 1e-4, err
    record_achieved(f"wgrad3x3/{B}x{H}x{W}x{ci}x{co}s{stride}/rel", err)


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


def test_weight_preparation_of_a_segment_in_one_launch_is_the_cast_layout_and_flip():
    """dmm_wprep3x3_bf16 over a table of several convolutions == per convolution: the bf16 channels-last cast of the master and
    (square, stride 1) the flipped + transposed weight its data gradient runs on as a forward convolution -- bit for bit;
    and the data gradient computed that way equals autograd's."""
    g = torch.Generator(device=DEV).manual_seed(5)
    shapes = [(64, 64, True), (128, 64, False), (256, 256, True), (64, 192, False), (512, 512, True)]
    ws = [torch.randn((co, ci, 3, 3), generator=g, device=DEV) for co, ci, _ in shapes]
    rec, outs, tile = [], [], 0
    for w, (co, ci, sq) in zip(ws, shapes):
        d = torch.empty((co, ci, 3, 3), dtype=torch.bfloat16, device=DEV).contiguous(memory_format=torch.channels_last)
        dt = torch.empty((ci, co, 3, 3), dtype=torch.bfloat16, device=DEV).contiguous(memory_format=torch.channels_last) if sq else None
        rec += [w.data_ptr(), d.data_ptr(), 0 if dt is None else dt.data_ptr(), co | (ci << 32), tile]
        tile += (co // 32) * (ci // 32)
        outs.append((d, dt))
    table = torch.tensor(rec, dtype=torch.int64, device=DEV)
    L = _lib.load()
    _lib.check(L.dmm_wprep3x3_bf16(table.data_ptr(), len(ws), tile, torch.cuda.current_stream().cuda_stream), "wprep")
    torch.cuda.synchronize()
    for w, (d, dt) in zip(ws, outs):
        ref = w.to(torch.bfloat16)
        assert torch.equal(d, ref) and d.is_contiguous(memory_format=torch.channels_last)
        if dt is not None:
            assert torch.equal(dt, torch.flip(ref, (2, 3)).transpose(0, 1))
    # the flipped weight computes the data gradient
    w, (d, dt) = ws[2], outs[2]
    x = torch.randn((2, 256, 9, 11), generator=g, device=DEV).bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    dy = torch.randn((2, 256, 9, 11), generator=g, device=DEV).bfloat16().contiguous(memory_format=torch.channels_last)
    F.conv2d(x, d, None, 1, 1).backward(dy)
    got = F.conv2d(dy, dt, None, 1, 1)
    assert (got.float() - x.grad.float()).abs().max().item() <= 2e-2 * x.grad.float().abs().max().item()
    assert L.dmm_wprep3x3_bf16(None, 0, 0, None) == 0 and L.dmm_wprep3x3_bf16(None, 2, 5, None) != 0


def test_casting_many_weights_in_one_launch_is_the_bf16_cast():
    g = torch.Generator(device=DEV).manual_seed(11)
    srcs = [torch.randn(n, generator=g, device=DEV) for n in (64 * 64, 8192, 8200, 2048 * 512, 8, 1024 * 256 + 8)]
    dsts = [torch.zeros(s_.numel() + 8, dtype=torch.bfloat16, device=DEV) for s_ in srcs]          # (+ 8: nothing behind n is touched)
    rec, blk = [], 0
    for s_, d in zip(srcs, dsts):
        rec += [s_.data_ptr(), d.data_ptr(), s_.numel(), blk]
        blk += (s_.numel() + 8191) // 8192
    table = torch.tensor(rec, dtype=torch.int64, device=DEV)
    L = _lib.load()
    _lib.check(L.dmm_cast_many_bf16(table.data_ptr(), len(srcs), blk, torch.cuda.current_stream().cuda_stream), "cast_many")
    torch.cuda.synchronize()
    for s_, d in zip(srcs, dsts):
        assert torch.equal(d[:-8], s_.to(torch.bfloat16)) and not d[-8:].any()
    assert L.dmm_cast_many_bf16(None, 0, 0, None) == 0 and L.dmm_cast_many_bf16(None, 1, 1, None) != 0


@pytest.mark.parametrize("B,C,H,W", [(12, 256, 64, 112), (2, 512, 33, 57), (3, 64, 7, 9), (1, 8, 1, 1)])
def test_stride_2_subsample_and_its_gradient_are_the_strided_copies(B, C, H, W):
    from dmm_net_amd.train_encoder import _subsample, _upsample_zero
    g = torch.Generator(device=DEV).manual_seed(B + C)
    x = torch.randn((B, C, H, W), generator=g, device=DEV).bfloat16().contiguous(memory_format=torch.channels_last)
    y = _subsample(x, 2)
    assert torch.equal(y, x[:, :, ::2, ::2]) and y.is_contiguous(memory_format=torch.channels_last)
    dy = torch.randn(y.shape, generator=g, device=DEV).bfloat16().contiguous(memory_format=torch.channels_last)
    dx = _upsample_zero(dy, tuple(x.shape), 2)
    ref = torch.zeros_like(x)
    ref[:, :, ::2, ::2] = dy
    assert torch.equal(dx, ref) and dx.is_contiguous(memory_format=torch.channels_last)
    L = _lib.load()
    assert L.dmm_subsample2_bf16(x.data_ptr(), 1, 4, 4, 12, y.data_ptr(), None) != 0       # C % 8
    assert L.dmm_upsample2_zero_bf16(None, 1, 4, 4, 8, None, None) != 0


def test_bias_gradient_through_the_statistics_kernel():
    from dmm_net_amd.train_encoder import _channel_sums
    g = torch.Generator(device=DEV).manual_seed(9)
    for shape in ((12, 256, 16, 28), (2, 64, 5, 7), (3, 24, 4, 4)):          # (24 channels: the fallback)
        dy = torch.randn(shape, generator=g, device=DEV).bfloat16().contiguous(memory_format=torch.channels_last)
        ref = dy.double().sum((0, 2, 3))
        got = _channel_sums(dy).double()
        assert (got - ref).abs().max().item() <= 1e-4 * (1 + ref.abs().max().item()) * math.sqrt(dy.numel() / shape[1])


@pytest.mark.parametrize("model", ["resnet34", "resnet50"])
def test_graphed_fp32_mode_equals_the_plain_encoder_on_replays(model):
    """``TrainEncoder(dtype=float32)`` computes what ``FeatureEncoder`` computes (stock BatchNorm, fp32 GEMMs): outputs,
    every parameter gradient and the BatchNorm buffers must agree over THREE steps with different images -- the first call
    captures, the later ones replay (a mis-ordered memset node or a stale static buffer shows up only there)."""
    torch.manual_seed(3)
    ref = _tame(FeatureEncoder(model, hidden_size=32).to(DEV).train())
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
        assert _rel(gt, gr) <= 1e-2, (step, _rel(gt, gr))     # (fp32 summation-order noise of MIOpen's atomics through 34-50 layers: 1e-3 .. 6e-3)
        for (n, a), (_, b) in zip(enc.named_buffers(), ref.named_buffers()):
            assert float((a.float() - b.float()).abs().max()) <= 1e-3 * (float(b.float().abs().max()) + 1.0), (step, n)


@pytest.mark.parametrize("B,C,H,W,G", [(12, 256, 16, 28, 3), (6, 1024, 8, 14, 2), (8, 64, 9, 11, 4), (3, 512, 4, 5, 3)])
@pytest.mark.parametrize("relu,has_res", [(True, False), (True, True), (False, False)])
def test_grouped_bn_kernels_are_the_plain_kernels_called_once_per_group(B, C, H, W, G, relu, has_res):
    """``groups`` statistics groups in one launch == the layer called on the G sub-batches in order: outputs, saved statistics,
    the running statistics after G momentum updates, dx / dres per group, dweight / dbias as the sums over the groups."""
    g = torch.Generator(device=DEV).manual_seed(C + G)
    cl = torch.channels_last
    x = (torch.randn((B, C, H, W), generator=g, device=DEV) * 1.5 + 0.3).bfloat16().contiguous(memory_format=cl)
    res = torch.randn((B, C, H, W), generator=g, device=DEV).bfloat16().contiguous(memory_format=cl) if has_res else None
    dy = torch.randn((B, C, H, W), generator=g, device=DEV).bfloat16().contiguous(memory_format=cl)
    w = torch.rand(C, generator=g, device=DEV) + 0.5
    b = torch.randn(C, generator=g, device=DEV) * 0.1

    def run(groups):
        rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
        outs = []
        if groups == G:
            xs, rs, ds = [x], [res], [dy]
        else:
            xs, ds = list(x.chunk(G, 0)), list(dy.chunk(G, 0))
            rs = list(res.chunk(G, 0)) if has_res else [None] * G
        dws, dbs = 0, 0
        for xi, ri, di in zip(xs, rs, ds):
            xi = xi.contiguous(memory_format=cl).detach().requires_grad_(True)
            ri = None if ri is None else ri.contiguous(memory_format=cl).detach().requires_grad_(True)
            wi, bi = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
            y = _BNActFn.apply(xi, wi, bi, rm, rv, 0.1, 1e-5, relu, ri, groups if groups == G else 1)
            y.backward(di.contiguous(memory_format=cl))
            outs.append((y.detach(), xi.grad, None if ri is None else ri.grad))
            dws, dbs = dws + wi.grad, dbs + bi.grad
        cat = lambda k: None if outs[0][k] is None else torch.cat([o[k] for o in outs], 0)
        return cat(0), cat(1), cat(2), dws, dbs, rm, rv
    got, want = run(G), run(1)
    for k, (a, c) in enumerate(zip(got, want)):
        if a is None:
            assert c is None
            continue
        a, c = a.float(), c.float()
        tol = 2e-2 if k < 3 else 2e-3                                        # (bf16 planes: an ulp where the sums' order differs)
        assert float((a - c).abs().max()) <= tol * (float(c.abs().max()) + 1e-3), (k, float((a - c).abs().max()))
    L = _lib.load()
    assert L.dmm_bn_stats_grouped_bf16(x.data_ptr(), 10, 64, 3, x.data_ptr(), None) == 1      # rows % groups


@pytest.mark.parametrize("relu,has_res,G", [(True, True, 1), (True, False, 1), (False, False, 3), (True, True, 3)])
def test_forked_output_adds_its_two_gradients_inside_the_backward_kernels(relu, has_res, G):
    """``fork=True``: the layer's output as two tensors for two consumers; the gradients that arrive for them are added inside
    the backward kernels (fp32) -- the same dx / dres / dweight / dbias as one output with the summed gradient."""
    g = torch.Generator(device=DEV).manual_seed(17 + G)
    cl = torch.channels_last
    B, C, H, W = 6, 256, 9, 13
    x = (torch.randn((B, C, H, W), generator=g, device=DEV) * 1.3).bfloat16().contiguous(memory_format=cl)
    res = torch.randn((B, C, H, W), generator=g, device=DEV).bfloat16().contiguous(memory_format=cl) if has_res else None
    da = torch.randn((B, C, H, W), generator=g, device=DEV).bfloat16().contiguous(memory_format=cl)
    db_ = torch.randn((B, C, H, W), generator=g, device=DEV).bfloat16().contiguous(memory_format=cl)
    w = torch.rand(C, generator=g, device=DEV) + 0.5
    b = torch.randn(C, generator=g, device=DEV) * 0.1

    def run(fork):
        xi = x.detach().clone().requires_grad_(True)
        ri = None if res is None else res.detach().clone().requires_grad_(True)
        wi, bi = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
        out = _BNActFn.apply(xi, wi, bi, rm, rv, 0.1, 1e-5, relu, ri, G, fork)
        if fork:
            assert torch.equal(out[0], out[1])
            torch.autograd.backward(list(out), [da, db_])
        else:
            out.backward((da.float() + db_.float()).to(torch.bfloat16))
        return xi.grad, None if ri is None else ri.grad, wi.grad, bi.grad
    got, want = run(True), run(False)
    for k, (a, c) in enumerate(zip(got, want)):
        if a is None:
            assert c is None
            continue
        a, c = a.float(), c.float()
        tol = 3e-2 if k < 2 else 1e-2                   # (the reference rounds the summed gradient to bf16 first)
        assert float((a - c).abs().max()) <= tol * (float(c.abs().max()) + 1e-3), (k, float((a - c).abs().max()))
    # only one of the two outputs used
    xi = x.detach().clone().requires_grad_(True)
    y1, y2 = _BNActFn.apply(xi, w.clone().requires_grad_(True), b.clone().requires_grad_(True), None, None, 0.1, 1e-5, relu,
                            None, G, True)
    y2.backward(da)
    assert xi.grad is not None and bool(torch.isfinite(xi.grad.float()).all())


def test_a_clip_in_one_call_with_per_frame_statistics_is_the_trainers_loop_of_calls():
    """``TrainEncoder.forward(cat(frames), bn_groups=T)`` against the reference's loop (one encoder call per frame step, one
    backward; trainer.py:95-131) on the plain ``FeatureEncoder``: features per frame, the gradients of the summed loss, the
    running statistics after T updates and ``num_batches_tracked``.  fp32 mode (same arithmetic as the plain encoder), graph
    replays included; then the bf16 form: first-layer statistics and the count."""
    torch.manual_seed(7)
    ref = _tame(FeatureEncoder("resnet50", hidden_size=32).to(DEV).train())
    enc = copy.deepcopy(ref)
    te = TrainEncoder(enc, dtype=torch.float32)
    T = 3
    for step in range(3):
        frames = [torch.randn(2, 3, 96, 128, device=DEV) for _ in range(T)]
        for m in (ref, enc):
            m.zero_grad(set_to_none=True)
        fr = [ref(f) for f in frames]
        ft = te(torch.cat(frames, 0), bn_groups=T)
        for t_ in range(T):
            for a, b in zip(ft["backbone_feature"] + ft["refine_input_feat"],
                            fr[t_]["backbone_feature"] + fr[t_]["refine_input_feat"]):
                a = a[2 * t_:2 * t_ + 2]
                assert float((a.float() - b).abs().max()) <= 2e-3 * float(b.abs().max()), (step, t_)
        sum(_loss(f) for f in fr).backward()
        # the same loss on the stacked features, frame by frame (_loss: fixed directions per output element)
        sum(_loss({k: tuple(v[2 * t_:2 * t_ + 2] for v in ft[k]) for k in ft}) for t_ in range(T)).backward()
        gr, gt = _grads(ref), _grads(enc)
        assert all((gr[k] is None) == (gt[k] is None) for k in gr), step
        assert _rel(gt, gr) <= 1e-2, (step, _rel(gt, gr))
        for (n, a), (_, b) in zip(enc.named_buffers(), ref.named_buffers()):
            assert float((a.float() - b.float()).abs().max()) <= 1e-3 * (float(b.float().abs().max()) + 1.0), (step, n)
    # the bf16 form: the stem's BatchNorm sees the same bf16 convolution outputs either way
    ref2 = _tame(FeatureEncoder("resnet50", hidden_size=32).to(DEV).train())
    enc2 = copy.deepcopy(ref2)
    loop = TrainEncoder(ref2, graphs=False)
    clip = TrainEncoder(enc2)
    frames = [torch.randn(2, 3, 96, 128, device=DEV) for _ in range(T)]
    for f in frames:
        loop(f)
    clip(torch.cat(frames, 0), bn_groups=T)
    n0 = int(ref2.base.bn1.num_batches_tracked)
    # (the capturing call runs its warm-up passes too: the counts differ by a multiple of T, the statistics converge to the same)
    assert (int(enc2.base.bn1.num_batches_tracked) - n0) % T == 0
    out = clip(torch.cat(frames, 0), bn_groups=T)
    assert all(torch.isfinite(o.float()).all() for o in out["backbone_feature"])


def _tame(enc, gamma=0.2):
    """Residual branches start small (torchvision's ``zero_init_residual`` idea).  A random-init ResNet with unit gammas doubles
    a perturbation every few blocks: ANY two bf16 evaluations of it -- two eager runs of the same code: MIOpen's split-K
    kernels and the statistics' atomics sum in arrival order -- then disagree by O(1) after 16-33 blocks."""
    with torch.no_grad():
        for m in enc.base.modules():
            if hasattr(m, "bn3"):
                m.bn3.weight.fill_(gamma)
            elif hasattr(m, "conv2") and not hasattr(m, "conv3"):
                m.bn2.weight.fill_(gamma)
    return enc


def test_graphed_bf16_step_is_as_close_to_fp32_as_the_eager_bf16_step():
    """The shipped form (bf16, fused BatchNorm, GEMM 1x1s, own weight gradients, graph replays).  bf16 evaluations of a
    50-layer network are not reproducible to the bit (see ``_tame``), so the comparison is statistical, over replays:
    (a) no gradient is ever non-finite (what the memset nodes of MIOpen's weight-gradient solvers produced before
        ``SafeGraph`` rewrote them), and the parameters the step must not touch stay without a gradient;
    (b) graph replay vs the same functions run eagerly: no further apart than two EAGER runs are from each other (x 1.5);
    (c) against the fp32 encoder: the replayed step is as close as the eager step (x 1.15), the heads' gradients within 25 %,
        all gradients aligned (cosine >= 0.85)."""
    from conftest import record_achieved
    torch.manual_seed(5)
    ref = _tame(FeatureEncoder("resnet50").to(DEV).train())
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
        b.zero_grad(set_to_none=True)
        loss(eager(img)).backward()
        ge2 = _grads(b)
        for name, g in (("graph", gg), ("eager", ge)):
            bad = [k for k, v in g.items() if v is not None and not bool(torch.isfinite(v).all())]
            assert not bad, (step, name, bad[:4])
        assert all((gg[k] is None) == (ge[k] is None) for k in ge)
        assert all(gg[k] is None for k in gg if k.startswith(("sk", "bn")))       # skips_need_grad=False: like autograd
        e_ge, e_ee = _rel(gg, ge), _rel(ge2, ge)
        assert e_ge <= 1.5 * e_ee + 0.02, (step, e_ge, e_ee)
        e_g32, e_e32 = _rel(gg, gr), _rel(ge, gr)
        assert e_g32 <= 1.15 * e_e32 + 0.02, (step, e_g32, e_e32)
        heads = lambda g: {k: v for k, v in g.items() if k.startswith("prop") and v is not None}
        e_heads = _rel(heads(gg), heads(gr))
        ks = [k for k in gr if gr[k] is not None]
        dot = sum(float((gg[k] * gr[k]).sum()) for k in ks)
        cos = dot / math.sqrt(sum(float(gg[k].square().sum()) for k in ks) * sum(float(gr[k].square().sum()) for k in ks))
        assert e_heads <= 0.25 and cos >= 0.85, (step, e_heads, cos)
        assert abs(float(lg) - float(lr)) <= 0.05 * max(abs(float(lr)), 0.05), (float(lg), float(lr))
        for tag, v in (("graph_vs_eager", e_ge), ("eager_vs_eager", e_ee), ("graph_vs_fp32", e_g32), ("eager_vs_fp32", e_e32),
                       ("heads_vs_fp32", e_heads), ("cos_vs_fp32", cos)):
            record_achieved(f"train_encoder/step{step}/{tag}", v)
    plan = next(iter(graph._plans.values()))[0]
    record_achieved("train_encoder/memset_nodes_rewritten", sum(v[0] for d in plan.rewritten.values() for v in d.values()))


def test_weight_gradients_on_the_side_stream_change_no_gradient():
    """``overlap_wgrad``: a segment's weight gradients are captured into a second graph that replays on a side stream beside
    the next segment's backward chain, and the segment is handed over one segment late.  Against the same plan with the
    launches inline: the heads' weight gradients and all gradients together are no further apart than two runs of the inline
    form are from each other (the body's bf16 features are not reproducible run to run, see ``_tame``), no gradient is missing
    or non-finite -- over replays, and with three forwards in flight.  (The kernels themselves are pinned bit-reproducibly in
    ``test_wgrad_*``; what this test adds is that the deferred launches read the right, still-live operands.)"""
    torch.manual_seed(21)
    ref = _tame(FeatureEncoder("resnet50").to(DEV).train())
    a, b = copy.deepcopy(ref), copy.deepcopy(ref)
    side, inline = TrainEncoder(a, skips_need_grad=False), TrainEncoder(b, overlap_wgrad=False, skips_need_grad=False)
    assert side.overlap_wgrad and not inline.overlap_wgrad
    loss = lambda f: _loss(f, skips=False)
    heads_w = lambda g: {k: v for k, v in g.items() if k.startswith("prop") and v is not None and v.dim() == 4}
    for step in range(3):
        frames = [torch.randn(4, 3, 128, 224, device=DEV) for _ in range(3 if step == 2 else 1)]
        for m in (a, b):
            m.zero_grad(set_to_none=True)
        sum(loss(side(f)) for f in frames).backward()
        sum(loss(inline(f)) for f in frames).backward()
        gs, gi = _grads(a), _grads(b)
        b.zero_grad(set_to_none=True)
        sum(loss(inline(f)) for f in frames).backward()
        gi2 = _grads(b)
        assert all((gs[k] is None) == (gi[k] is None) for k in gi)
        assert not [k for k, v in gs.items() if v is not None and not bool(torch.isfinite(v).all())]
        assert _rel(heads_w(gs), heads_w(gi)) <= 1.5 * _rel(heads_w(gi2), heads_w(gi)) + 0.02
        assert _rel(gs, gi) <= 1.5 * _rel(gi2, gi) + 0.02, (step, _rel(gs, gi), _rel(gi2, gi))
    plan = next(iter(side._plans.values()))[0]
    # (the stem's 7x7 convolution stays on the stock path: its segment has no weight-gradient graph of its own)
    assert set(plan.wgrad) == set(side.segments) - {"stem"} and side.segments[0] == "stem" and side.segments[-1] == "heads"
    assert len(plan.fwd_body) == 4 and len(plan.fwd_head) == 4 and plan.side is not None
    assert not plan.busy


def test_a_plain_fused_optimiser_steps_on_the_handed_over_gradients():
    """Without a bucketer the optimiser reads the handed-over gradients as they are: every gradient must have its
    parameter's dtype and layout (torch's multi-tensor Adam refuses anything else) -- also for the convolutions that stay on
    the stock path (the 7x7 stem, the 32-wide head), whose weight gradients autograd returns channels-last strided."""
    torch.manual_seed(4)
    enc = _tame(FeatureEncoder("resnet50").to(DEV).train())
    te = TrainEncoder(enc, skips_need_grad=False)
    opt = torch.optim.Adam([p for p in enc.parameters()], lr=1e-4, fused=True)
    before = enc.base.conv1.weight.detach().clone()
    losses = []
    img = torch.randn(4, 3, 128, 224, device=DEV)
    for _ in range(3):
        opt.zero_grad(set_to_none=True)
        l = _loss(te(img), skips=False)
        l.backward()
        for n, p in enc.named_parameters():
            assert p.grad is None or (p.grad.stride() == p.stride() and p.grad.dtype == p.dtype), n
        opt.step()
        losses.append(float(l))
    assert not torch.equal(before, enc.base.conv1.weight) and losses[-1] < losses[0], losses


def test_gradient_hand_over_feeds_the_bucketer_and_accumulates():
    """Parameter gradients are handed over like AccumulateGrad does: ``p.grad`` set (or added to) and the post-accumulate
    hooks called, segment by segment, last segment first -- ``GradBucketer(overlap=True)`` launches its buckets in order and
    the reduced gradients (world 1: unchanged) equal a plain backward's; a second backward accumulates."""
    import os
    import torch.distributed as dist
    from dmm_net_amd.distributed import GradBucketer
    torch.manual_seed(9)
    # fp32 mode: the hand-over is the same code for every dtype, and fp32 evaluations are reproducible enough to compare sums
    # of gradients from different calls (two bf16 evaluations of a deep network are not, see _tame)
    enc = _tame(FeatureEncoder("resnet34", hidden_size=32).to(DEV).train())
    te = TrainEncoder(enc, dtype=torch.float32, skips_need_grad=False)
    img = torch.randn(4, 3, 96, 128, device=DEV)
    loss = lambda f: _loss(f, skips=False)
    img2 = torch.randn(4, 3, 96, 128, device=DEV)
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


def test_several_forwards_in_flight_like_the_trainers_clip():
    """The reference's trainer calls the encoder once per frame of a clip and backpropagates ONCE through all of them
    (trainer.py:95-131, train.py:296-307).  Every forward that is still waiting for its backward owns a plan of its own (the
    second / third call of the same shape captures another one; later steps reuse them), each parameter's hooks run once per
    backward pass with the SUM of the frames' gradients, and a forward whose outputs are dropped without a backward gives its
    plan back.  fp32 mode (reproducible arithmetic) against the plain encoder."""
    torch.manual_seed(11)
    ref = _tame(FeatureEncoder("resnet34", hidden_size=32).to(DEV).train())
    enc = copy.deepcopy(ref)
    te = TrainEncoder(enc, dtype=torch.float32, skips_need_grad=False)
    calls = []
    for p in enc.parameters():
        p.register_post_accumulate_grad_hook(lambda q: calls.append(id(q)))
    loss = lambda f: _loss(f, skips=False)
    for step in range(3):
        frames = [torch.randn(2, 3, 96, 128, device=DEV) for _ in range(3)]
        for m in (ref, enc):
            m.zero_grad(set_to_none=True)
        calls.clear()
        with torch.no_grad():
            te(frames[0])                                         # (no_grad: eager, touches no plan)
        dropped = te(frames[1])                                   # a forward nobody backpropagates: its plan comes back
        del dropped
        sum(loss(te(f)) for f in frames).backward()
        sum(loss(ref(f)) for f in frames).backward()
        gt, gr = _grads(enc), _grads(ref)
        assert all((gr[k] is None) == (gt[k] is None) for k in gr if not k.startswith(("sk", "bn")))
        assert _rel(gt, {k: v for k, v in gr.items() if not k.startswith(("sk", "bn"))}) <= 5e-3, step
        used = [id(p) for p in enc.parameters() if p.grad is not None]
        assert sorted(calls) == sorted(used)                      # once per parameter, whatever the number of frames
        plans = next(iter(te._plans.values()))
        assert len(plans) == 3 and not any(p.busy for p in plans)


def test_a_failed_capture_leaves_the_stream_and_the_next_capture_intact():
    """``SafeGraph``: a capture that fails (here: a synchronising call under capture) raises ``CaptureFailed``, the calling
    thread's stream is the one it was, later launches and captures work.  (``torch.cuda.graph`` leaves the capture stream
    current in that case.  Not covered, on purpose: a capture invalidated from ANOTHER thread while MIOpen is inside it -- a
    stress test with a pinning / uploading thread hit that once in ~10 full-suite runs, and the library's internal streams
    then stay in capture state for the rest of the process: LABLOG round 6, item 18.)"""
    from dmm_net_amd.graphs import CaptureFailed, SafeGraph
    cur = torch.cuda.current_stream()
    x = torch.ones(1024, device=DEV)
    g = SafeGraph()
    with pytest.raises(CaptureFailed):
        with g.capture():
            y = x * 2
            y.sum().item()                                # a synchronising call: not permitted under capture
    assert torch.cuda.current_stream() == cur and not torch.cuda.is_current_stream_capturing()
    assert float((x + 1).sum()) == 2048.0
    g2 = SafeGraph()
    with g2.capture():
        z = x * 3
    g2.replay()
    torch.cuda.synchronize()
    assert float(z.sum()) == 3072.0


def test_swapped_parameter_storage_is_seen_by_the_next_call():
    """The graphs hold parameter ADDRESSES: after ``p.data = other_tensor`` (what ``load_state_dict(assign=True)`` does) the
    next forward must compute with the new storage -- the plans are captured again -- not replay on the old one."""
    torch.manual_seed(2)
    enc = _tame(FeatureEncoder("resnet34", hidden_size=32).to(DEV).train())
    te = TrainEncoder(enc, dtype=torch.float32)
    img = torch.randn(2, 3, 96, 128, device=DEV)
    a = te(img)["backbone_feature"][0].detach().clone()
    b = te(img)["backbone_feature"][0].detach().clone()                       # (a replay: the same up to the running statistics)
    assert float((a - b).abs().max()) <= 1e-4 * float(a.abs().max())
    with torch.no_grad():
        enc.base.layer1[0].conv1.weight.data = torch.zeros_like(enc.base.layer1[0].conv1.weight)
    c = te(img)["backbone_feature"][0].detach().clone()
    ref = TrainEncoder(enc, dtype=torch.float32, graphs=False)(img)["backbone_feature"][0].detach()
    assert float((c - a).abs().max()) > 1e-3 * float(a.abs().max())          # the zeroed weight is in effect
    assert float((c - ref).abs().max()) <= 2e-3 * float(ref.abs().max())
    sum(p.float().mean() for p in te(img)["backbone_feature"]).backward()     # and a backward through the new plans runs
    assert enc.base.layer1[0].conv1.weight.grad is not None


def test_safe_graph_turns_memset_and_memcpy_nodes_into_kernel_nodes():
    """``graphs.SafeGraph``: a capture that contains hipMemsetAsync calls and a device-to-device hipMemcpyAsync (what MIOpen /
    torch issue inside a captured step) is rewritten before it is instantiated -- the memset nodes become kernel nodes -- and
    replays give what the eager sequence gives, every time (element sizes 1 / 2 / 4, odd byte counts, an unaligned start)."""
    import ctypes
    from dmm_net_amd.graphs import SafeGraph
    hip = None
    with open("/proc/self/maps") as f:
        for ln in f:
            if "libamdhip64.so" in ln:
                hip = ctypes.CDLL(ln.split()[-1])
                break
    assert hip is not None
    hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
    hip.hipMemsetD16Async.argtypes = [ctypes.c_void_p, ctypes.c_ushort, ctypes.c_size_t, ctypes.c_void_p]
    hip.hipMemsetD32Async.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
    n = 4099
    buf8 = torch.zeros(n + 3, dtype=torch.uint8, device=DEV)
    buf16 = torch.zeros(n, dtype=torch.int16, device=DEV)
    buf32 = torch.zeros(n, dtype=torch.int32, device=DEV)
    src = torch.arange(n, dtype=torch.float32, device=DEV)
    dst = torch.zeros(n, dtype=torch.float32, device=DEV)
    acc = torch.zeros(n, dtype=torch.float32, device=DEV)
    g = SafeGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with g.capture():
            s = torch.cuda.current_stream().cuda_stream
            assert hip.hipMemsetAsync(buf8.data_ptr() + 1, 0x5A, n, s) == 0          # unaligned start, odd count
            assert hip.hipMemsetD16Async(buf16.data_ptr(), 0x1234, n, s) == 0
            assert hip.hipMemsetD32Async(buf32.data_ptr(), 7, n, s) == 0
            dst.copy_(src)                                                            # same dtype, contiguous: a memcpy node
            acc += dst + buf32.float() + buf16.float() + buf8[1:n + 1].float()
    torch.cuda.current_stream().wait_stream(side)
    # (the copy_ is a 1-D memcpy node: this runtime's API cannot read its description back, it stays and is counted in `left`)
    assert g.rewritten[0] == 3 and g.rewritten[1] + g.left >= 1, (g.rewritten, g.left)
    for k in range(1, 4):
        buf8.fill_(1), buf16.fill_(1), buf32.fill_(1), dst.fill_(-1.0)                # what a mis-ordered node would leave behind
        src.add_(1.0)
        g.replay()
        torch.cuda.synchronize()
        assert int(buf8[0]) == 1 and int(buf8[n + 1]) == 1 and bool((buf8[1:n + 1] == 0x5A).all())
        assert bool((buf16 == 0x1234).all()) and bool((buf32 == 7).all()) and torch.equal(dst, src)
    want = sum(torch.arange(n, dtype=torch.float32, device=DEV) + j for j in range(1, 4)) + 3 * (7 + 0x1234 + 0x5A)
    assert torch.equal(acc, want)

