"""bf16 channels-last TRAINING form of the encoder (reference a10 as the trainer runs it; BASELINE configs[3]).

Counterpart of ``dmm/modules/model_encoder.py:86-162`` + ``base.py:18-69`` + ``vision.py:6-38`` under
``train.py:296-307`` (forward, ``loss.backward()``, optimiser step).  Same parameters as ``FeatureEncoder`` -- the fp32
``nn.Parameter`` objects of the wrapped encoder stay the masters the optimiser, the checkpoints and
``distributed.GradBucketer`` see -- evaluated the way that is fast on MI355X (12 frames of 255 x 448, ResNet-101: 31.25 ms
per training step with the stock fp32 modules, 15.4 ms here; ``profiles/r06_cfg4_train_encoder_steps.md`` has every step in
between):

* **forward and backward as HIP-graph replays.**  The stock step is ~2 100 launches of 5-50 us kernels and HOST bound (17.6 ms
  of kernels inside a 24.4 ms step under bf16 autocast).  The body is cut into a CHAIN of segments (stem | layer1 | layer2 |
  layer3 in parts | layer4) plus the heads; the forward between two taps and every segment's backward
  (``torch.autograd.grad`` over the segment) are captured once per input shape and replayed; segments hand activations and
  gradients to each other in static buffers.  Memset nodes of a capture are rewritten into kernel nodes (``graphs.SafeGraph``).
* **two streams.**  In the backward only the data gradients are on the critical chain.  A segment's weight gradients are
  recorded while its chain is captured and captured afterwards into a graph of their own that replays on a SIDE stream beside
  the next segment's chain; the heads of a pyramid level (forward and backward) are side-stream graphs too.  (Two graphs on
  two streams overlap on this runtime; forked branches inside one graph do not: ``tools/graph_branch_probe.py``.)
* **bf16 activations, channels-last, end to end; fp32 master weights.**  BatchNorm runs on bf16 activations with fp32
  parameters and statistics.  No autocast.
* **1x1 convolutions are matrix products** of the [B*H*W, Cin] activation matrix (2/3 of a bottleneck): forward and data
  gradient on hipBLASLt; 3x3 forward / data gradient on MIOpen; EVERY weight gradient from ``dmm_wgrad*_bf16``
  (``csrc/dmm_wgrad.hip``: MFMA, split over the rows, fp32 straight into the master's gradient, no atomics) -- hipBLASLt's pick
  for these products has no split-K (66 us each), MIOpen's bf16 solvers bring a zeroing and a cast launch each.
* **BatchNorm (+ residual) (+ ReLU) as two launches each way** (``dmm_bn_*`` in ``csrc/dmm_encoder_train.hip``) where the
  stock path issues 5-6 (three MIOpen BatchNorm kernels + add + clamp; backward likewise).

Gradient hand-over.  Parameter gradients are NOT returned through autograd (350 inputs to one node).  Each segment is an
autograd node chained to its neighbours by a token, and also takes the segment's HUB leaf: a leaf's AccumulateGrad runs once
per backward pass after its last user -- the moment every forward in flight (the reference's trainer calls the encoder once
per frame of a clip and backpropagates once, ``trainer.py:95-131``; each such forward owns a plan) has produced the segment's
gradients.  The hub's hook sums them, sets / accumulates ``p.grad`` and calls the parameters' post-accumulate-grad hooks, like
AccumulateGrad does -- ``GradBucketer(overlap=True)`` and plain optimisers work unchanged.  A segment is handed over one
segment late (the hand-over waits for its side-stream graph).  As with DDP's bucket views, a gradient may alias a static
buffer of the captured step: it is valid until the encoder's next forward (which moves a gradient that is still in place --
gradient accumulation -- into a tensor of its own first); do not hold on to the tensor object itself across steps.

Not supported: double backward, ``retain_graph`` replays of the same forward, forward hooks on the wrapped modules, changing
``requires_grad`` of a parameter after the first forward of a shape (the captured backward computes the gradients of the
parameters that required one at capture time).  In-place parameter updates (optimiser steps, ``load_state_dict``) are seen
by the replays; ``.to()`` / ``.cuda()`` drop the captured plans.  Fallbacks inside the same segments: the 7x7 stem and widths
that are not multiples of 64 take the stock convolution ops, ``dtype=float32`` the stock BatchNorm as well (the mode the tests
pin the plumbing with).  Off the GPU (or with ``graphs=False``) the same segment functions run eagerly under autograd.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from .encoder import _NO_CTX, Bottleneck, FeatureEncoder, ResNetBody
from .graphs import CaptureFailed, SafeGraph

_CL = torch.channels_last
_CAPTURE_LOCK = __import__("threading").Lock()     # plan captures set module-level scratch (_ARENA, _DEFER) and put the process
#                                                    into HIP's global capture mode: one at a time (nn.DataParallel threads)


# ---- scratch for the BatchNorm statistics ------------------------------------------------------------------------------
class _Arena:
    """fp32 scratch the statistics kernels accumulate into (they need zeroed [2, C] blocks): inside a captured segment ONE
    zeroing launch at the head of the graph clears the blocks of all its layers -- a ``torch.zeros`` per layer and direction
    was 238 launches of ~4 us per ResNet-101 step (profiles/r06_cfg4_train_encoder_steps.md).  A capture is traced
    once, so handing out consecutive slices while it runs fixes every layer's block for all replays."""

    def __init__(self, device, floats: int = 1 << 20):
        self.buf = torch.empty(int(floats), dtype=torch.float32, device=device)     # (zeroed by the scope, once per replay)
        self.off = 0

    def take(self, n: int) -> torch.Tensor:
        n = (n + 63) & ~63
        if self.off + n > self.buf.numel():
            raise RuntimeError("TrainEncoder: statistics arena exhausted")
        t = self.buf[self.off:self.off + n]
        self.off += n
        return t


_ARENA: Optional[_Arena] = None          # set while a segment is being captured (captures never run side by side)


def _zeroed(n: int, device) -> torch.Tensor:
    if _ARENA is not None and _ARENA.buf.device == device:
        return _ARENA.take(n)
    return torch.zeros(n, dtype=torch.float32, device=device)


class _arena_scope:
    def __init__(self, arena):
        self.arena = arena

    def __enter__(self):
        global _ARENA
        _ARENA = self.arena
        self.arena.buf.zero_()               # a fill KERNEL (captured: replayed at the head of the graph)
        return self.arena

    def __exit__(self, *exc):
        global _ARENA
        _ARENA = None


# ---- BatchNorm (+ residual) (+ ReLU), training mode, bf16 NHWC: two launches each way -------------------------------
class _BNActFn(torch.autograd.Function):
    """y = act(BN_train(x) (+ residual)) on a channels-last bf16 activation; statistics, scale and shift in fp32.
    forward = ``dmm_bn_stats_bf16`` (per-channel sum / sum of squares) + ``dmm_bn_apply_bf16`` (normalise, residual, ReLU,
    running statistics); backward = ``dmm_bn_bwd_reduce_bf16`` (sum g, sum g*xhat with g = dy * [y > 0]) +
    ``dmm_bn_bwd_dx_bf16`` (dx, and g as the residual branch's gradient)."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, momentum, eps, relu, residual, groups=1, fork=False):
        from . import _lib
        L = _lib.load()
        assert x.is_cuda and x.dtype == torch.bfloat16 and x.is_contiguous(memory_format=_CL)
        B, C, H, W = x.shape
        assert B % groups == 0, (B, groups)
        R = B * H * W
        stream = torch.cuda.current_stream(x.device).cuda_stream
        stats = _zeroed(groups * 2 * C, x.device)
        y = torch.empty_like(x, memory_format=_CL)
        saved = torch.empty((groups, 2, C), dtype=torch.float32, device=x.device)  # mean, invstd of every statistics group
        res = None
        if residual is not None:
            assert residual.shape == x.shape and residual.dtype == x.dtype
            res = residual.contiguous(memory_format=_CL)
        with _lib.device_guard(x.device):
            _lib.check(L.dmm_bn_stats_grouped_bf16(x.data_ptr(), R, C, groups, stats.data_ptr(), stream),
                       "dmm_bn_stats_grouped_bf16")
            _lib.check(L.dmm_bn_apply_grouped_bf16(x.data_ptr(), None if res is None else res.data_ptr(), R, C, groups,
                                                   stats.data_ptr(), weight.data_ptr(), bias.data_ptr(),
                                                   None if running_mean is None else running_mean.data_ptr(),
                                                   None if running_var is None else running_var.data_ptr(), float(momentum),
                                                   float(eps), int(relu), y.data_ptr(), saved.data_ptr(), stream),
                       "dmm_bn_apply_grouped_bf16")
        ctx.save_for_backward(x, y, weight, bias, saved)
        ctx.relu, ctx.has_res, ctx.groups = bool(relu), residual is not None, groups
        # fork: the output as TWO tensors (the second an alias) for its two consumers -- a residual block's first convolution and
        # its identity branch --, so that their gradients arrive separately and are added inside the backward kernels
        return (y, y.view_as(y)) if fork else y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy, dy2=None):
        from . import _lib
        L = _lib.load()
        x, y, weight, bias, saved = ctx.saved_tensors
        # no residual in front of the ReLU: the mask is recomputed from x (the forward's own fma), y is not read
        mode = 0 if not ctx.relu else (1 if ctx.has_res else 2)
        B, C, H, W = x.shape
        R = B * H * W
        if dy is None:                                    # (only the alias was used)
            dy, dy2 = dy2, None
        dy = dy.contiguous(memory_format=_CL)
        if dy2 is not None:
            dy2 = dy2.to(dy.dtype).contiguous(memory_format=_CL)
        stream = torch.cuda.current_stream(x.device).cuda_stream
        groups = ctx.groups
        sums = _zeroed(groups * 2 * C, x.device)
        dx = torch.empty_like(x, memory_format=_CL)
        dres = torch.empty_like(x, memory_format=_CL) if ctx.has_res else None
        dw = torch.empty((C,), dtype=torch.float32, device=x.device)
        db = torch.empty((C,), dtype=torch.float32, device=x.device)
        with _lib.device_guard(x.device):
            p2 = None if dy2 is None else dy2.data_ptr()
            _lib.check(L.dmm_bn_bwd_reduce_grouped_bf16(dy.data_ptr(), p2, x.data_ptr(), y.data_ptr(), R, C, groups,
                                                        saved.data_ptr(), weight.data_ptr(), bias.data_ptr(), mode,
                                                        sums.data_ptr(), stream), "dmm_bn_bwd_reduce_grouped_bf16")
            _lib.check(L.dmm_bn_bwd_dx_grouped_bf16(dy.data_ptr(), p2, x.data_ptr(), y.data_ptr(), R, C, groups, saved.data_ptr(),
                                                    weight.data_ptr(), bias.data_ptr(), sums.data_ptr(), mode, dx.data_ptr(),
                                                    None if dres is None else dres.data_ptr(), dw.data_ptr(), db.data_ptr(),
                                                    stream), "dmm_bn_bwd_dx_grouped_bf16")
        return dx, dw, db, None, None, None, None, None, dres, None, None


def _bn_fusable(bn: nn.BatchNorm2d) -> bool:
    c8 = bn.num_features // 8
    return (bn.affine and bn.track_running_stats and bn.momentum is not None and bn.num_features % 8 == 0
            and 0 < c8 <= 256 and 256 % c8 == 0)


def _bn_act(x, bn: nn.BatchNorm2d, relu: bool, residual=None, fused: bool = True, counted: bool = False, groups: int = 1,
            fork: bool = False):
    """BatchNorm2d module ``bn`` (its fp32 parameters and running statistics) on a bf16 channels-last activation,
    followed by the optional residual add and ReLU.  ``fused`` and training and on the device: the two-launch HIP form
    (``counted``: the caller has already advanced ``num_batches_tracked`` for its whole segment in one launch).
    ``groups`` > 1 (training): the batch is ``groups`` consecutive sub-batches, each normalised with its OWN statistics and
    counted as a call of its own by the running statistics -- what ``groups`` calls of the module on the sub-batches do."""
    if fused and bn.training and x.is_cuda and x.dtype == torch.bfloat16 and _bn_fusable(bn):
        if not counted:
            with torch.no_grad():
                bn.num_batches_tracked.add_(groups)
        return _BNActFn.apply(x.contiguous(memory_format=_CL), bn.weight, bn.bias, bn.running_mean, bn.running_var,
                              bn.momentum, bn.eps, relu, residual, groups, fork)
    if groups > 1 and bn.training:
        y = torch.cat([bn(c) for c in x.chunk(groups, 0)], 0)
    else:
        y = bn(x)
    if residual is not None:
        y = y + residual
    y = F.relu(y) if relu else y
    return (y, y) if fork else y                          # (``fork``: the output for two consumers; only the fused form makes it two tensors)


# ---- weight gradients: launched where the backward reaches them, or collected for a graph of their own -------------------
_DGRAD_AS_FORWARD = True                 # (module switch for the A/B in tools/cfg4_probe.py)
_DEFER: Optional[list] = None            # set while a segment's backward is captured with ``overlap_wgrad``: the launches are
#                                          recorded (operands kept alive) and captured afterwards into a SECOND graph that replays
#                                          on a side stream beside the next segment's backward chain


def _wgrad_launch(rec):
    from . import _lib
    L = _lib.load()
    kind, dy, x, dims, dw = rec
    dev = x.device
    stream = torch.cuda.current_stream(dev).cuda_stream
    if kind == "1x1":
        rows, co, ci = dims
        ws = torch.empty((max(int(L.dmm_wgrad_workspace_bytes(rows, co, ci)), 16),), dtype=torch.uint8, device=dev)
        with _lib.device_guard(dev):
            _lib.check(L.dmm_wgrad_bf16(dy.data_ptr(), x.data_ptr(), rows, co, ci, co, ci, dw.data_ptr(), ws.data_ptr(),
                                        ws.numel(), stream), "dmm_wgrad_bf16")
    else:
        B, H, W, ci, co, stride, Ho, Wo = dims
        ws = torch.empty((max(int(L.dmm_wgrad_workspace_bytes(B * Ho * Wo, co, 9 * ci)), 16),), dtype=torch.uint8, device=dev)
        with _lib.device_guard(dev):
            _lib.check(L.dmm_wgrad3x3_bf16(dy.data_ptr(), x.data_ptr(), B, H, W, ci, co, stride, dw.data_ptr(), ws.data_ptr(),
                                           ws.numel(), stream), "dmm_wgrad3x3_bf16")


def _wgrad(rec):
    if _DEFER is not None:
        _DEFER.append(rec)
    else:
        _wgrad_launch(rec)


# ---- convolutions ----------------------------------------------------------------------------------------------------
def _wgrad_ok(m: nn.Conv2d) -> bool:
    return (m.groups == 1 and m.dilation == (1, 1) and m.in_channels % 64 == 0 and m.out_channels % 64 == 0
            and m.stride in ((1, 1), (2, 2)))


def _subsample(x, stride: int):
    """x[:, :, ::stride, ::stride] of a channels-last activation as a tensor of its own: one 16-byte-per-thread launch for stride 2
    in bf16 (the stock strided copy moves 2 bytes per thread: 52 us for layer2's downsample input at config 4, 5 with this)."""
    B, C, H, W = x.shape
    if stride == 2 and x.is_cuda and x.dtype == torch.bfloat16 and C % 8 == 0 and x.is_contiguous(memory_format=_CL):
        from . import _lib
        y = torch.empty((B, C, (H + 1) // 2, (W + 1) // 2), dtype=x.dtype, device=x.device, memory_format=_CL)
        with _lib.device_guard(x.device):
            _lib.check(_lib.load().dmm_subsample2_bf16(x.data_ptr(), B, H, W, C, y.data_ptr(),
                                                       torch.cuda.current_stream(x.device).cuda_stream), "dmm_subsample2_bf16")
        return y
    return x[:, :, ::stride, ::stride].contiguous(memory_format=_CL)


def _upsample_zero(dy, full, stride: int):
    """The gradient of ``_subsample``: a ``full``-shaped tensor with dy at the sampled positions and zero elsewhere, written once."""
    B, C, H, W = full
    if stride == 2 and dy.is_cuda and dy.dtype == torch.bfloat16 and C % 8 == 0:
        from . import _lib
        dy = dy.contiguous(memory_format=_CL)
        dx = torch.empty(full, dtype=dy.dtype, device=dy.device, memory_format=_CL)
        with _lib.device_guard(dy.device):
            _lib.check(_lib.load().dmm_upsample2_zero_bf16(dy.data_ptr(), B, H, W, C, dx.data_ptr(),
                                                           torch.cuda.current_stream(dy.device).cuda_stream),
                       "dmm_upsample2_zero_bf16")
        return dx
    dx = torch.zeros(full, dtype=dy.dtype, device=dy.device).to(memory_format=_CL)
    dx[:, :, ::stride, ::stride] = dy
    return dx


class _Conv1x1Fn(torch.autograd.Function):
    """1x1 convolution of a channels-last bf16 activation with an fp32 MASTER weight: y = X W^T on the [B*H*W, Cin] activation
    matrix (hipBLASLt); backward: dX = dY W (hipBLASLt) and dW = dY^T X straight into an fp32 gradient for the master
    (``dmm_wgrad_bf16``: MFMA, split over the rows) -- no bf16 weight gradient, no cast launch.  stride 2 = a row subsample."""

    @staticmethod
    def forward(ctx, x, weight, stride, shadow=None):
        co, ci = weight.shape[0], weight.shape[1]
        ctx.full = None
        if stride != 1:
            ctx.full = tuple(x.shape)
            x = _subsample(x, stride)
        B, _, H, W = x.shape
        # shadow: the bf16 copy of the weight the caller refreshed for its whole segment in one multi-tensor launch
        w = shadow if shadow is not None else weight.detach().view(co, ci).to(torch.bfloat16)
        y = torch.mm(x.permute(0, 2, 3, 1).reshape(B * H * W, ci), w.t())
        ctx.save_for_backward(x, w)
        ctx.stride = stride
        return y.view(B, H, W, co).permute(0, 3, 1, 2)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        from . import _lib
        x, w = ctx.saved_tensors
        B, ci, H, W = x.shape
        co = w.shape[0]
        dy = dy.contiguous(memory_format=_CL)
        dy_rows = dy.permute(0, 2, 3, 1).reshape(B * H * W, co)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.mm(dy_rows, w).view(B, H, W, ci).permute(0, 3, 1, 2)
            if ctx.full is not None:
                dx = _upsample_zero(dx, ctx.full, ctx.stride)
        dw = torch.empty((co, ci, 1, 1), dtype=torch.float32, device=x.device)
        _wgrad(("1x1", dy_rows, x, (B * H * W, co, ci), dw))
        return dx, dw, None, None


class _Conv3x3Fn(torch.autograd.Function):
    """3x3 / padding 1 convolution (stride 1 or 2) of a channels-last bf16 activation with an fp32 MASTER weight: forward and
    data gradient on MIOpen, the weight gradient by ``dmm_wgrad3x3_bf16`` (implicit patch matrix, MFMA, fp32 out) -- MIOpen's
    bf16 weight-gradient solvers cost a zeroing and a cast launch each and clear their workspace with a memset node."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, shadow=None):
        # stride 1 and as many channels in as out (conv2 of a bottleneck): the data gradient is the SAME convolution problem with
        # the weight flipped and transposed, dX = conv(dY, W'[ci, co, kh, kw] = W[co, ci, 2 - kh, 2 - kw]) -- MIOpen's forward
        # kernel for it takes 28 us where its backward-data kernel takes 50 (profiles/r06_kernel_stats_config4_bf16.csv), on
        # the backward's critical chain; the flipped copy is made in the forward, off that chain.
        # shadow = (w, wt): both made for the caller's whole segment in one launch (``TrainEncoder._tick``)
        if shadow is not None:
            w, wt = shadow
        else:
            w = weight.detach().to(dtype=torch.bfloat16, memory_format=_CL)
            wt = None
            if _DGRAD_AS_FORWARD and stride == 1 and w.shape[0] == w.shape[1] and x.requires_grad:
                wt = torch.flip(w, (2, 3)).transpose(0, 1).contiguous(memory_format=_CL)
        if not (_DGRAD_AS_FORWARD and stride == 1 and w.shape[0] == w.shape[1]):
            wt = None
        b = None if bias is None else bias.detach().to(torch.bfloat16)
        y = F.conv2d(x, w, b, stride, 1)
        ctx.save_for_backward(x, w, wt if wt is not None else w)
        ctx.stride, ctx.has_bias, ctx.flipped = stride, bias is not None, wt is not None
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        x, w, wt = ctx.saved_tensors
        B, ci, H, W = x.shape
        co = w.shape[0]
        dy = dy.contiguous(memory_format=_CL)
        dx = None
        if ctx.needs_input_grad[0]:
            if ctx.flipped:
                dx = F.conv2d(dy, wt, None, 1, 1)
            else:
                dx = torch.ops.aten.convolution_backward(dy, x, w, None, [ctx.stride] * 2, [1, 1], [1, 1], False, [0, 0], 1,
                                                         [True, False, False])[0]
        Ho, Wo = dy.shape[2], dy.shape[3]
        dw = torch.empty((co, ci, 3, 3), dtype=torch.float32, device=x.device)      # the master's own layout
        _wgrad(("3x3", dy, x, (B, H, W, ci, co, ctx.stride, Ho, Wo), dw))
        db = _channel_sums(dy) if ctx.has_bias else None
        return dx, dw, db, None, None


def _channel_sums(dy):
    """sum over (batch, rows, columns) of a channels-last bf16 gradient in fp32 (a bias gradient): the statistics kernel's first
    row -- one launch where ``dy.float().sum((0, 2, 3))`` is a cast of the whole tensor and a reduction."""
    B, C, H, W = dy.shape
    c8 = C // 8
    if dy.is_cuda and dy.dtype == torch.bfloat16 and C % 8 == 0 and c8 <= 256 and 256 % c8 == 0:
        from . import _lib
        stats = _zeroed(2 * C, dy.device)
        with _lib.device_guard(dy.device):
            _lib.check(_lib.load().dmm_bn_stats_bf16(dy.data_ptr(), B * H * W, C, stats.data_ptr(),
                                                     torch.cuda.current_stream(dy.device).cuda_stream), "dmm_bn_stats_bf16")
        return stats[:C]
    return dy.float().sum((0, 2, 3))


def _conv(x, m: nn.Conv2d, dtype, linear_1x1: bool = True, own_wgrad: bool = True, shadow=None):
    """Convolution ``m`` (fp32 master weight) on a channels-last activation in the compute dtype.  On the device in bf16: a
    1x1 convolution is the product of the activation matrix with W^T (hipBLASLt forward and data gradient), a 3x3 one runs on
    MIOpen, and both take their weight gradient from ``dmm_wgrad*_bf16``; anything else (the 7x7 stem, widths that are not a
    multiple of 64, fp32 mode, the CPU) goes through the stock ops with a differentiable cast of the weight."""
    fast = own_wgrad and x.is_cuda and dtype == torch.bfloat16 and x.dtype == dtype and _wgrad_ok(m)
    one = m.kernel_size == (1, 1) and m.padding == (0, 0)
    if fast and linear_1x1 and one and m.bias is None:
        return _Conv1x1Fn.apply(x.contiguous(memory_format=_CL), m.weight, m.stride[0], shadow)
    if fast and m.kernel_size == (3, 3) and m.padding == (1, 1):
        return _Conv3x3Fn.apply(x.contiguous(memory_format=_CL), m.weight, m.bias, m.stride[0], shadow)
    b = None if m.bias is None else m.bias.to(dtype)
    if linear_1x1 and one and m.groups == 1 and m.dilation == (1, 1):
        if m.stride != (1, 1):
            x = x[:, :, ::m.stride[0], ::m.stride[1]].contiguous(memory_format=_CL)
        w = m.weight.view(m.out_channels, m.in_channels).to(dtype)
        return F.linear(x.permute(0, 2, 3, 1), w, b).permute(0, 3, 1, 2)
    w = m.weight.to(dtype=dtype, memory_format=_CL)
    return F.conv2d(x, w, b, m.stride, m.padding, m.dilation, m.groups)


class TrainEncoder(nn.Module):
    """``TrainEncoder(FeatureEncoder(...))``: same ``forward(img) -> feature dict`` (model_encoder.py:157-160), same
    parameters (``self.src`` is the wrapped encoder: build the optimiser / save checkpoints from it as before), training
    mode, bf16 channels-last, HIP-graph replays.  ``backbone_feature`` comes back as fp32 NCHW (what the ROI feature
    kernel's backward takes); ``refine_input_feat`` / ``body_feature`` as bf16 channels-last.

    ``skips_need_grad``: the decoder's inputs (``refine_input_feat``: ``sk_k`` + ``bn_k``) take part in the backward.  A
    trainer with the refine decoder leaves it on; a step that never sends a gradient there (bench.py's config 4: no decoder)
    switches it off so that the backward graph does not run those four convolutions on zeros.

    A/B switches (each measured in ``profiles/r06_cfg4_train_encoder_steps.md``; the defaults are the fast settings):
    ``graphs`` (False: the same segment functions eagerly), ``linear_1x1`` (False: 1x1 convolutions on MIOpen too),
    ``fused_bn`` (False: ``torch.nn.BatchNorm2d`` + add + relu), ``own_wgrad`` (False: the libraries' weight gradients),
    ``overlap_wgrad`` (False: one stream -- weight gradients inline in the chain, heads on the main stream), ``layer3_parts``
    (segments layer3 is cut into), ``miopen_find`` (let MIOpen search its solvers during the warm-up of a new shape; the
    shapes of BASELINE configs[3] ship in ``dmm_net_amd/miopen_db``)."""

    def __init__(self, encoder: FeatureEncoder, dtype=torch.bfloat16, graphs: bool = True, linear_1x1: bool = True,
                 fused_bn: bool = True, own_wgrad: bool = True, overlap_wgrad: bool = True, skips_need_grad: bool = True,
                 miopen_find: bool = False, warmup: int = 2, layer3_parts: int = 3):
        super().__init__()
        if not isinstance(encoder.base, ResNetBody):
            raise NotImplementedError("TrainEncoder is the bf16 channels-last training form of the ResNet bodies")
        self.src = encoder
        self.dtype, self.graphs, self.linear_1x1, self.fused_bn = dtype, bool(graphs), bool(linear_1x1), bool(fused_bn)
        self.own_wgrad = bool(own_wgrad)
        self.overlap_wgrad = bool(overlap_wgrad) and self.own_wgrad
        self.skips_need_grad, self.miopen_find, self.warmup = bool(skips_need_grad), bool(miopen_find), int(warmup)
        # the body as a CHAIN of segments: stem | layer1 | layer2 | layer3 in ``layer3_parts`` runs of blocks | layer4, then the
        # heads.  (name, the blocks it runs, which of the four taps x2..x5 its output is, or None.)  Finer segments = a finer
        # pipeline between the backward chain and the weight-gradient graphs on the side stream: the exposed tail is the FIRST
        # segment's weight gradients, and a segment's weight gradients should not outlast the next segment's chain (layer3 is
        # half of a ResNet-101: as ONE segment its 2.5 ms of weight gradients ran beside 1.7 ms of chain).
        body = encoder.base
        l3 = list(body.layer3)
        n3 = max(1, min(int(layer3_parts), len(l3)))
        cuts = [round(i * len(l3) / n3) for i in range(n3 + 1)]
        chain = [("stem", ["stem"], None), ("layer1", list(body.layer1), 0), ("layer2", list(body.layer2), 1)]
        chain += [(f"layer3_{i}" if n3 > 1 else "layer3", l3[cuts[i]:cuts[i + 1]], 2 if i == n3 - 1 else None) for i in range(n3)]
        chain.append(("layer4", list(body.layer4), 3))
        self.__dict__["_chain"] = chain
        self.segments = tuple(c[0] for c in chain) + ("heads",)
        self.__dict__["_plans"] = {}             # (shape, device) -> _Plan; not module state
        self.__dict__["_hubs"] = {}              # device index -> {segment: hub leaf}
        self.__dict__["_pending"] = {}           # segment -> plans whose backward of it ran in the current backward pass
        self.__dict__["_ticked"] = {}            # id(module) -> covered by the running segment's _tick (rebuilt per call)
        self.__dict__["_shadows"] = {}           # id(1x1 conv) -> its persistent bf16 weight copy

    def _apply(self, fn, *args, **kwargs):
        # .to() / .cuda() / .float() re-create the parameters' storage: the captured graphs (which hold the old addresses), the
        # bf16 weight copies and the hub leaves go; the next forward captures again
        self._plans.clear(), self._shadows.clear(), self._hubs.clear(), self._pending.clear()
        self.__dict__.get("_wprep", {}).clear()
        self.__dict__.get("_wcast", {}).clear()
        return super()._apply(fn, *args, **kwargs)

    # ---- the encoder in segments (plain functions of tensors; parameters come from self.src) ------------------------
    def _cbr(self, x, conv, bn, relu, residual=None, fork=False):
        t = self.__dict__["_ticked"]
        y = _conv(x, conv, self.dtype, self.linear_1x1, self.own_wgrad, t.get(id(conv)))
        return _bn_act(y, bn, relu, residual, self.fused_bn, counted=id(bn) in t, groups=self.__dict__.get("_bn_groups", 1),
                       fork=fork)

    def _tick(self, name: str, x: torch.Tensor, mods=None):
        """Per-segment housekeeping in ONE launch each instead of one per layer: ``num_batches_tracked += 1`` of every
        BatchNorm that takes the fused kernels (116 launches per ResNet-101 step before), and the bf16 copies of the 1x1
        weights (70).  ``_ticked`` tells ``_cbr`` which modules were covered."""
        t = self.__dict__["_ticked"]
        if not (x.is_cuda and self.dtype == torch.bfloat16 and self.training):
            return
        mods = [m for mod in (self._seg_modules()[name] if mods is None else mods) for m in mod.modules()]
        if self.fused_bn:
            bns = [m for m in mods if isinstance(m, nn.BatchNorm2d) and m.training and _bn_fusable(m)]
            if bns:
                with torch.no_grad():
                    torch._foreach_add_([m.num_batches_tracked for m in bns], self.__dict__.get("_bn_groups", 1))
                for m in bns:
                    t[id(m)] = True
        if self.own_wgrad and self.linear_1x1:
            convs = [m for m in mods if isinstance(m, nn.Conv2d) and m.kernel_size == (1, 1) and m.padding == (0, 0)
                     and m.bias is None and _wgrad_ok(m)]
            if convs:
                sh = self.__dict__["_shadows"]
                dst = []
                for m in convs:
                    w = sh.get(id(m))
                    if w is None or w.device != m.weight.device:
                        w = sh[id(m)] = torch.empty((m.out_channels, m.in_channels), dtype=torch.bfloat16, device=m.weight.device)
                    dst.append(w)
                    t[id(m)] = w
                self._cast_1x1(convs, dst)
        if self.own_wgrad:
            convs = [m for m in mods if isinstance(m, nn.Conv2d) and m.kernel_size == (3, 3) and m.padding == (1, 1)
                     and _wgrad_ok(m)]
            if convs:
                self._prep_3x3(convs, t)

    def _cast_1x1(self, convs, dst):
        """The bf16 copies of a segment's 1x1 weights by ONE launch over a device table (``torch._foreach_copy_`` with a dtype
        change: 20-38 us per segment); inside a capture without a table yet, the multi-tensor copy."""
        from . import _lib
        memo = self.__dict__.setdefault("_wcast", {})
        key = tuple(id(m) for m in convs)
        ptrs = tuple(m.weight.data_ptr() for m in convs) + tuple(d.data_ptr() for d in dst)
        got = memo.get(key)
        ok = all(m.weight.is_contiguous() and m.weight.numel() % 8 == 0 and m.weight.data_ptr() % 16 == 0 for m in convs) \
            and all(d.data_ptr() % 16 == 0 for d in dst)        # (16-byte pieces; e.g. parameters that are views of a flat buffer)
        if not ok or ((got is None or got[0] != ptrs) and torch.cuda.is_current_stream_capturing()):
            with torch.no_grad():
                torch._foreach_copy_(dst, [m.weight.detach().view(m.out_channels, m.in_channels) for m in convs])
            return
        dev = convs[0].weight.device
        if got is None or got[0] != ptrs:
            rec, blk = [], 0
            for m, d in zip(convs, dst):
                n = m.weight.numel()
                rec += [m.weight.data_ptr(), d.data_ptr(), n, blk]
                blk += (n + 8191) // 8192
            got = memo[key] = (ptrs, _lib.small_to_device(rec, torch.int64, dev), blk)
        with _lib.device_guard(dev):
            _lib.check(_lib.load().dmm_cast_many_bf16(got[1].data_ptr(), len(convs), got[2],
                                                      torch.cuda.current_stream(dev).cuda_stream), "dmm_cast_many_bf16")

    def _prep_3x3(self, convs, t):
        """bf16 channels-last copies of the 3x3 weights of one segment (and the flipped + transposed ones their data gradients
        run on as forward convolutions) by ONE launch over a device table -- a cast, a layout copy and a flip launch per
        convolution before: 99 launches of ~5.5 us in a ResNet-101 forward."""
        from . import _lib
        memo = self.__dict__.setdefault("_wprep", {})
        key = tuple(id(m) for m in convs)
        ptrs = tuple(m.weight.data_ptr() for m in convs)
        got = memo.get(key)
        if got is None or got[0] != ptrs:
            if torch.cuda.is_current_stream_capturing():
                return                                   # (no table yet and no upload inside a capture: per-convolution copies)
            dev = convs[0].weight.device
            rec, pairs, tile = [], [], 0
            for m in convs:
                co, ci = m.out_channels, m.in_channels
                w = torch.empty((co, ci, 3, 3), dtype=torch.bfloat16, device=dev, memory_format=_CL)
                wt = None
                if _DGRAD_AS_FORWARD and m.stride == (1, 1) and co == ci:
                    wt = torch.empty((ci, co, 3, 3), dtype=torch.bfloat16, device=dev, memory_format=_CL)
                rec += [m.weight.data_ptr(), w.data_ptr(), 0 if wt is None else wt.data_ptr(), co | (ci << 32), tile]
                tile += (co // 32) * (ci // 32)
                pairs.append((w, wt))
            table = _lib.small_to_device(rec, torch.int64, dev)
            got = memo[key] = (ptrs, table, pairs, tile)
        _, table, pairs, tiles = got
        dev = convs[0].weight.device
        with _lib.device_guard(dev):
            _lib.check(_lib.load().dmm_wprep3x3_bf16(table.data_ptr(), len(convs), tiles,
                                                     torch.cuda.current_stream(dev).cuda_stream), "dmm_wprep3x3_bf16")
        for m, pr in zip(convs, pairs):
            t[id(m)] = pr

    def _block(self, x, blk, fork=False):
        """One residual block.  ``x``: a tensor, or the PAIR (x, alias) a predecessor made with ``fork`` -- the first convolution
        takes one, the identity branch (or the downsample convolution) the other, so that the two gradients reach the
        predecessor's BatchNorm separately and are added inside its backward kernels (autograd's own add of them: one launch
        over the whole tensor per block).  ``fork``: return such a pair for the next block."""
        xa, xb = x if isinstance(x, tuple) else (x, x)
        idt = xb if blk.downsample is None else self._cbr(xb, blk.downsample[0], blk.downsample[1], False)
        if isinstance(blk, Bottleneck):
            out = self._cbr(xa, blk.conv1, blk.bn1, True)
            out = self._cbr(out, blk.conv2, blk.bn2, True)
            return self._cbr(out, blk.conv3, blk.bn3, True, idt, fork=fork)
        out = self._cbr(xa, blk.conv1, blk.bn1, True)
        return self._cbr(out, blk.conv2, blk.bn2, True, idt, fork=fork)

    def _layer(self, x, layer):
        for q, blk in enumerate(layer):
            x = self._block(x, blk, fork=q + 1 < len(layer))
        return x

    def _seg_body(self, name: str, x):
        """One body segment of the chain: (the stem when it is the first,) its blocks in order.  -> (output,)"""
        body = self.src.base
        self._tick(name, x)
        ops_ = next(c[1] for c in self._chain if c[0] == name)
        for q, op in enumerate(ops_):
            if isinstance(op, str):                                      # "stem": conv1 -> bn1 -> relu -> maxpool
                x = x.to(self.dtype).contiguous(memory_format=_CL)
                x = body.maxpool(self._cbr(x, body.conv1, body.bn1, True))
            else:
                # (the segment's last block hands ONE tensor on: its consumers are other graphs)
                x = self._block(x, op, fork=q + 1 < len(ops_) and not isinstance(ops_[q + 1], str))
        return (x,)

    def _head(self, x, head):
        out = self._cbr(x, head[0], head[1], True)                       # base.py:43-54: conv -> BN -> ReLU -> conv -> BN
        return self._cbr(out, head[3], head[4], False)

    def _head_level(self, k: int, x, tick: bool = True):
        """The heads of ONE pyramid level (k = 2..5) on that level's tap: ``prop_k`` (-> fp32 NCHW: what the ROI feature kernel,
        forward and backward, works on) and the decoder's skip projection ``bn_k(sk_k(x))``.  -> (prop, skip)"""
        s = self.src
        if tick:
            self._tick("heads", x, mods=self._head_modules(k))
        prop = self._head(x, getattr(s, f"prop{k}"))
        with (torch.enable_grad() if self.skips_need_grad else torch.no_grad()):
            skip = self._cbr(x, getattr(s, f"sk{k}"), getattr(s, f"bn{k}"), False)
        return prop.float().contiguous(), skip

    def _head_modules(self, k: int):
        s = self.src
        return [getattr(s, f"prop{k}"), getattr(s, f"sk{k}"), getattr(s, f"bn{k}")]

    def _head_params(self, k: int) -> List[nn.Parameter]:
        mods = self._head_modules(k)
        if not self.skips_need_grad:
            mods = mods[:1]
        return [p for m in mods for p in m.parameters() if p.requires_grad]

    def _seg_heads(self, x2, x3, x4, x5):
        self._tick("heads", x2)
        lv = [self._head_level(k, x, tick=False) for k, x in ((2, x2), (3, x3), (4, x4), (5, x5))]
        return tuple(p for p, _ in lv) + tuple(sk for _, sk in reversed(lv))       # props 2..5, skips 5..2

    def _seg_modules(self) -> Dict[str, List[nn.Module]]:
        s, body = self.src, self.src.base
        out = {}
        for name, ops, _ in self._chain:
            mods = []
            for op in ops:
                mods += [body.conv1, body.bn1] if isinstance(op, str) else [op]
            out[name] = mods
        out["heads"] = [s.prop2, s.prop3, s.prop4, s.prop5, s.sk2, s.sk3, s.sk4, s.sk5, s.bn2, s.bn3, s.bn4, s.bn5]
        return out

    def _seg_params(self) -> Dict[str, List[nn.Parameter]]:
        out = {}
        for name, mods in self._seg_modules().items():
            if name == "heads":
                out[name] = [p for k in (2, 3, 4, 5) for p in self._head_params(k)]
            else:
                out[name] = [p for m in mods for p in m.parameters() if p.requires_grad]
        return out

    def _arena_floats(self, name: str, backward: bool) -> int:
        """Upper bound of what a segment's graph takes from its arena: [2, C] per BatchNorm and per convolution bias (+ 64
        floats of rounding per request), the same forward and backward (the weight-gradient kernels overwrite their
        outputs: nothing to zero)."""
        n = 0
        for mod in self._seg_modules()[name]:
            for m in mod.modules():
                if isinstance(m, nn.BatchNorm2d):
                    n += self.__dict__.get("_bn_groups", 1) * 2 * m.num_features + 64
                elif isinstance(m, nn.Conv2d) and m.bias is not None:
                    n += 2 * m.out_channels + 64            # (its bias gradient: the statistics kernel's sums)
        return n + 1024

    @staticmethod
    def _pack(taps, heads):
        return {"backbone_feature": tuple(heads[:4]), "refine_input_feat": tuple(heads[4:]), "body_feature": tuple(taps)}

    def _eager(self, img):
        x, taps = img, [None] * 4
        for name, _, tap in self._chain:
            (x,) = self._seg_body(name, x)
            if tap is not None:
                taps[tap] = x
        return self._pack(taps, self._seg_heads(*taps))

    # ---- entry ---------------------------------------------------------------------------------------------------------
    def forward(self, img: torch.Tensor, bn_groups: int = 1) -> Dict[str, Tuple[torch.Tensor, ...]]:
        """``bn_groups`` = G > 1 (training): ``img`` is G consecutive sub-batches and every BatchNorm takes its batch statistics
        per sub-batch -- the features, gradients and running statistics of G calls on the sub-batches in order (the reference's
        trainer calls the encoder once per frame step of a clip, trainer.py:95-131; a clip's frames stacked along the batch
        with G = frames is the same computation in one pass: convolutions never mix images)."""
        assert img.dim() == 4 and img.shape[1] == 3, img.shape           # model_encoder.py:91-92
        bn_groups = int(bn_groups) if self.training else 1
        assert bn_groups >= 1 and img.shape[0] % bn_groups == 0, (img.shape[0], bn_groups)
        self.__dict__["_bn_groups"] = bn_groups
        self.__dict__["_ticked"].clear()
        self.__dict__.pop("_late", None)         # (a hand-over left behind by a backward pass that raised)
        if not (self.graphs and img.is_cuda and self.training and torch.is_grad_enabled()):
            return self._eager(img)
        # the captured graphs hold the ADDRESSES of parameters and buffers: storage that was swapped since (``p.data = ...``,
        # ``load_state_dict(assign=True)``; ``.to()`` goes through ``_apply``) makes every plan stale -- captured again
        fp = tuple(t.data_ptr() for t in self.src.parameters()) + tuple(t.data_ptr() for t in self.src.buffers())
        if fp != self.__dict__.get("_storage"):
            if self.__dict__.get("_storage") is not None:
                self._plans.clear(), self._shadows.clear(), self._pending.clear()
                self.__dict__.get("_wprep", {}).clear()
                self.__dict__.get("_wcast", {}).clear()
            self.__dict__["_storage"] = fp
        key = (tuple(img.shape), img.dtype, img.device.index, self.skips_need_grad, bn_groups)
        plans = self._plans.setdefault(key, [])
        # a plan's static buffers belong to ONE forward until its backward has run: a second forward of the same shape before
        # that (the trainer's clip: one encoder call per frame, one backward; trainer.py:95-131) takes / captures another plan
        plan = next((p for p in plans if not p.busy), None)
        if plan is None:
            try:
                with _CAPTURE_LOCK:
                    plan = _Plan(self, img)
            except CaptureFailed as e:
                # a capture was invalidated (an API call of another thread of the process the runtime does not tolerate beside a
                # capture, even in thread-local mode): this call runs the same functions eagerly, the next call of the shape
                # captures again.  The calling thread's stream is intact (graphs.SafeGraph).
                import warnings
                warnings.warn(f"TrainEncoder: graph capture failed ({str(e)[:200]}); this call runs eagerly", RuntimeWarning)
                torch.cuda.synchronize(img.device)
                self.__dict__["_ticked"].clear()
                return self._eager(img)
            plans.append(plan)
        return plan.run(img)

    def _hubs_for(self, device):
        hubs = self._hubs.get(device.index)
        if hubs is None:
            hubs = self._hubs[device.index] = {}
            for name in self.segments:
                h = hubs[name] = torch.zeros((), device=device, requires_grad=True)
                h.register_post_accumulate_grad_hook(lambda t, name=name: self._flush(name, t))
        return hubs

    def _flush(self, name: str, hub: torch.Tensor):
        """The hub leaf of ``name`` has received its gradient: every plan that took part in this backward pass has replayed the
        segment's backward graph(s).  Their parameter gradients are summed (into the first plan's static buffers) and handed
        over: ``p.grad`` set / accumulated, post-accumulate-grad hooks called -- once per parameter and backward pass, like
        autograd's own AccumulateGrad.  With the weight gradients on a side stream the hand-over of a segment happens ONE
        SEGMENT LATE (the current stream has to wait for that segment's side-stream graph, and must not do so before the next
        segment's chain -- which the side stream runs beside -- has been enqueued); the first segment of the network, whose
        backward runs last, is handed over together with its predecessor."""
        hub.grad = None
        plans = self._pending.pop(name, [])
        late = self.__dict__.pop("_late", None)
        if late is not None:
            self._hand_over(*late)
        if not plans:
            return
        if any(p.side is not None and (name in p.wgrad or name == "heads") for p in plans) and name != self.segments[0]:
            self.__dict__["_late"] = (name, plans)
        else:
            self._hand_over(name, plans)

    def _hand_over(self, name: str, plans):
        for p in plans:
            p.wait_wgrad(name)
        params, base = plans[0].params[name], plans[0].pgrads[name]
        for other in plans[1:]:
            pairs = [(b, g) for b, g in zip(base, other.pgrads[name]) if b is not None and g is not None]
            if pairs:
                torch._foreach_add_([b for b, _ in pairs], [g for _, g in pairs])
        plans[0].aliased = True
        _deliver(params, base)


class _Plan:
    """The captured graphs of one input shape: forward graphs along ``enc.segments`` (stem -> layer1 -> layer2 -> layer3 in parts
    -> layer4 -> heads; each reads its predecessor's static outputs in place), backward graphs in the reverse order (each reads its successors' static input
    gradients in place), all in one memory pool and captured in the order they replay."""

    def __init__(self, enc: TrainEncoder, img: torch.Tensor):
        self.enc = enc
        dev = img.device
        chain = enc._chain
        self.params = enc._seg_params()
        self.static_img = img.detach().clone()
        pool = torch.cuda.graph_pool_handle()
        # ---- warm-up (eager, off the capture): MIOpen / hipBLASLt pick their kernels, the allocator reaches its size;
        # BatchNorm's running statistics are put back afterwards (the warm-up and the captures are not training steps)
        bns = [m for m in enc.src.modules() if isinstance(m, nn.BatchNorm2d)]
        keep = [(m.running_mean.clone(), m.running_var.clone(), m.num_batches_tracked.clone()) for m in bns]
        try:
            self._build(enc, img, dev, chain, pool)
        finally:                                          # (also when a capture failed: the caller falls back to the eager step)
            for (rm, rv, nb), m in zip(keep, bns):
                with torch.no_grad():
                    m.running_mean.copy_(rm), m.running_var.copy_(rv), m.num_batches_tracked.copy_(nb)

    def _build(self, enc, img, dev, chain, pool):
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        old = torch.backends.cudnn.benchmark
        torch.backends.cudnn.benchmark = old or enc.miopen_find
        try:
            with torch.cuda.stream(side):
                for _ in range(max(enc.warmup, 1)):
                    out = enc._eager(self.static_img)
                    leaves = [t for t in out["backbone_feature"] + out["refine_input_feat"] if t.requires_grad]
                    torch.autograd.grad(leaves, [p for ps in self.params.values() for p in ps],
                                        [torch.zeros_like(t) for t in leaves], allow_unused=True)
                    del out, leaves
        finally:
            torch.backends.cudnn.benchmark = old
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        # ---- forward captures, in replay order; a segment's input is a DETACHED view of its predecessor's output that
        # requires grad, so that every segment owns a separate autograd graph.  The BODY is captured as one graph per tap
        # interval (stem + layer1 | layer2 | layer3 | layer4: what the forward needs between two taps; the finer segments only
        # matter to the backward), the heads as one graph per pyramid level: a level's heads only need that level's tap, so
        # their graph replays on the SIDE stream beside the deeper body graphs (two graphs on two streams overlap on this
        # runtime, tools/graph_branch_probe.py); likewise their backward runs beside the body's backward chain.  Everything
        # that replays on the side stream allocates from a pool of its own.
        self.bwd, self.ins, self.outs = {}, {}, {}
        self.arenas = []                                  # statistics scratch of every captured graph (kept alive with it)
        overlap = enc.overlap_wgrad
        self.side = torch.cuda.Stream(device=dev) if overlap else None
        pool_s = torch.cuda.graph_pool_handle()           # the side stream's graphs (head levels, weight gradients)

        def leaf(t):
            return t.detach().requires_grad_(True)

        def new_arena(floats):
            a = _Arena(dev, floats)
            self.arenas.append(a)
            return a
        head_floats = enc._arena_floats("heads", False)
        self.fwd_body, self.fwd_head = [], []             # per tap interval: the body graph, the level's heads graph
        taps, hin, lv, x, k_ = [None] * 4, [None] * 4, [None] * 4, self.static_img, 0
        while k_ < len(chain):
            g = SafeGraph()
            todo = []
            for q in range(k_, len(chain)):
                todo.append(chain[q])
                if chain[q][2] is not None:
                    break
            with g.capture(pool=pool), _arena_scope(new_arena(sum(enc._arena_floats(n, False) for n, _, _ in todo))):
                for (name, _, tap) in todo:
                    inp = x if name == chain[0][0] else leaf(x)
                    (x,) = enc._seg_body(name, inp)
                    self.ins[name], self.outs[name] = (inp,), (x,)
            tap = todo[-1][2]
            taps[tap] = x
            k_ += len(todo)
            self.fwd_body.append(g)
            gh_ = SafeGraph()
            hin[tap] = leaf(x)
            with gh_.capture(pool=pool_s), _arena_scope(new_arena(head_floats)):
                lv[tap] = enc._head_level(tap + 2, hin[tap])
            self.fwd_head.append(gh_)
        heads = tuple(p_ for p_, _ in lv) + tuple(sk for _, sk in reversed(lv))     # props 2..5, skips 5..2 (as _seg_heads)
        self.ins["heads"], self.outs["heads"] = tuple(hin), heads
        self.ev_tap = [torch.cuda.Event() for _ in range(4)]
        self.ev_heads_fwd = torch.cuda.Event()
        # ---- backward captures, reverse order.  Gradients that arrive from outside: one static buffer per head output that
        # requires grad.  Gradients between segments: the tensors autograd.grad returned in the successor's capture.
        self.gout = [torch.zeros_like(o) if o.requires_grad else None for o in heads]
        self.pgrads = {}

        # weight gradients on a side stream (``overlap_wgrad``): a segment's backward is captured as TWO graphs -- the chain
        # (BatchNorm backward, data gradients, everything the next layer down waits for) and, from the launches recorded while
        # that capture ran, the weight gradients.  On replay the second graph runs on a side stream beside the NEXT segment's
        # chain.  The recorded operands (dY, X) are kept alive for the plan's lifetime -- the chain of the next segment must
        # not reuse their memory while the side stream reads it.
        self.wgrad, self.keep = {}, {}
        self.ev_chain = {n: torch.cuda.Event() for n in enc.segments}
        self.ev_wgrad = {n: torch.cuda.Event() for n in enc.segments}

        def capture_bwd(name, outs, make_gouts, inputs, ps, chain_pool):
            """-> (gradients of ``inputs``, gradients of ``ps``, chain graph, recorded weight-gradient launches or None).
            ``make_gouts``: called INSIDE the capture (sums of gradient buffers of two consumers are part of the graph)."""
            global _DEFER
            wrt = [t for t in inputs if t.requires_grad] + ps
            g = SafeGraph()
            _DEFER = [] if overlap else None
            try:
                with g.capture(pool=chain_pool), _arena_scope(new_arena(enc._arena_floats(name, True))):
                    grads = list(torch.autograd.grad(outs, wrt, make_gouts(), allow_unused=True))
                    # a gradient is handed over in its PARAMETER's layout (the multi-tensor optimisers refuse anything else:
                    # "params, grads ... must have same dtype, device, and layout"): the stock convolutions' weight gradients
                    # come back channels-last strided (the 7x7 stem, widths the library's kernels do not take)
                    for q, (t, g_) in enumerate(zip(wrt, grads)):
                        if g_ is not None and isinstance(t, nn.Parameter) and (g_.stride() != t.stride() or g_.dtype != t.dtype):
                            grads[q] = torch.empty_like(t).copy_(g_)
                records = _DEFER
            finally:
                _DEFER = None
            n_in = len(wrt) - len(ps)
            return grads[:n_in], list(grads[n_in:]), g, records

        def capture_wgrad(name, records):
            if not records:
                return None
            gw = SafeGraph()
            with gw.capture(pool=pool_s):
                for rec in records:
                    _wgrad_launch(rec)
            self.keep.setdefault(name, []).extend(records)
            return gw
        # the heads, level by level, deepest first (the order the body's backward needs their tap gradients in)
        self.bwd_head, self.ev_bh = [None] * 4, [torch.cuda.Event() for _ in range(4)]
        gh, hgrads, hrec = [None] * 4, {}, []
        where = {id(o): n for n, o in enumerate(heads)}
        for tap in (3, 2, 1, 0):
            prop, skip = lv[tap]
            outs_ = [o for o in (prop, skip) if o.requires_grad]
            gbuf = [self.gout[where[id(o)]] for o in outs_]
            ps = enc._head_params(tap + 2)
            (gin,), pg, g, records = capture_bwd("heads", outs_, lambda gbuf=gbuf: list(gbuf), (hin[tap],), ps, pool_s)
            gh[tap], self.bwd_head[tap] = gin, g
            hgrads.update({id(p_): g_ for p_, g_ in zip(ps, pg)})
            hrec.append(records)
        self.pgrads["heads"] = [hgrads.get(id(p_)) for p_ in self.params["heads"]]
        # (captured in the order they replay on the side stream -- the four chains, then the four weight-gradient graphs: graphs
        # that share a pool replay in their capture order)
        hw = [g for g in (capture_wgrad("heads", r) for r in hrec) if g is not None]
        if hw:
            self.wgrad["heads"] = hw
        # a tap feeds the next body segment AND the heads: its gradient is the sum of the two static buffers (one add, captured
        # at the head of the segment's backward)
        g_next = None
        for k_ in range(len(chain) - 1, -1, -1):
            name, _, tap = chain[k_]
            terms = [t for t in (g_next, gh[tap] if tap is not None else None) if t is not None]
            make = (lambda terms=terms: [terms[0] + terms[1]]) if len(terms) == 2 else (lambda terms=terms: [terms[0]])
            gin, pg, g, records = capture_bwd(name, list(self.outs[name]), make, self.ins[name] if k_ > 0 else (),
                                              self.params[name], pool)
            self.bwd[name], self.pgrads[name] = g, pg
            gw = capture_wgrad(name, records)
            if gw is not None:
                self.wgrad[name] = [gw]
            g_next = gin[0] if k_ > 0 else None
        self.tap_of = {name: tap for name, _, tap in chain}
        self.result = TrainEncoder._pack(taps, heads)
        self.heads = heads
        self.token = torch.zeros((), device=dev)           # what the segment nodes hand each other (autograd ordering only)
        self.zero = torch.zeros((), device=dev)            # their gradient: the data moves in the static buffers
        self.busy = False
        self.aliased = False                               # some p.grad may still BE one of this plan's static buffers
        # what the rewrite found: {segment: (memset nodes, memcpy nodes) turned into kernel nodes} per direction
        tot = lambda gs: tuple(sum(g.rewritten[q] for g in gs) for q in (0, 1))
        self.rewritten = {"fwd": {"body": tot(self.fwd_body), "heads": tot(self.fwd_head)},
                          "bwd": dict({k: g.rewritten for k, g in self.bwd.items()}, heads=tot(self.bwd_head)),
                          "wgrad": {k: tot(gs) for k, gs in self.wgrad.items()}}

    def run(self, img):
        enc = self.enc
        if self.aliased:
            # the last backward handed this plan's static gradient buffers out as ``p.grad``.  The pool reuses that memory for
            # activations of the FORWARD graphs (a replayed step needs the two at different times), so a gradient that is still
            # in place now -- gradient accumulation: no zero_grad / optimiser step in between -- moves into its own tensor first
            for name in enc.segments:
                for p, g in zip(self.params[name], self.pgrads[name]):
                    if g is not None and p.grad is not None and p.grad.data_ptr() == g.data_ptr():
                        p.grad = p.grad.clone()
            self.aliased = False
        self.static_img.copy_(img)
        main = torch.cuda.current_stream(img.device)
        for t in range(4):
            self.fwd_body[t].replay()
            if self.side is None:
                self.fwd_head[t].replay()
            else:                                          # this level's heads beside the deeper body graphs
                self.ev_tap[t].record(main)
                self.side.wait_event(self.ev_tap[t])
                with torch.cuda.stream(self.side):
                    self.fwd_head[t].replay()
        if self.side is not None:
            self.ev_heads_fwd.record(self.side)
            main.wait_event(self.ev_heads_fwd)
        self.busy = True
        hubs = enc._hubs_for(img.device)
        token = None
        for name in enc.segments[:-1]:
            token = _SegFn.apply(self, name, token, hubs[name])
        outs = _SegFn.apply(self, "heads", token, hubs["heads"])
        # (the body's own levels are handed out detached: inside the graphs they are inputs of the heads, not autograd leaves
        # of the caller)
        return {"backbone_feature": tuple(outs[:4]), "refine_input_feat": tuple(outs[4:]),
                "body_feature": tuple(t.detach() for t in self.result["body_feature"])}

    def backward_segment(self, name: str, grads: Sequence[Optional[torch.Tensor]] = ()):
        if name == "heads":
            k = 0
            for g in self.gout:
                if g is None:
                    continue
                gi = grads[k] if k < len(grads) else None
                if gi is None:
                    g.zero_()
                elif gi.data_ptr() != g.data_ptr():
                    g.copy_(gi)
                k += 1
        # gradient accumulation (a second backward before zero_grad): a gradient that still IS this graph's static buffer
        # would be overwritten by the replay -- it moves into its own tensor first (the usual flow, gradients set to None or
        # re-pointed at bucket views between steps, never takes this path)
        for p, g in zip(self.params[name], self.pgrads[name]):
            if g is not None and p.grad is not None and p.grad.data_ptr() == g.data_ptr():
                p.grad = p.grad.clone()
        main = torch.cuda.current_stream(self.static_img.device)
        side = self.side
        if name == "heads":
            # the heads' backward, deepest level first, on the side stream: the body's chain only waits for the level whose tap
            # gradient it needs next
            if side is not None:
                self.ev_chain[name].record(main)           # (the incoming gradients are in their static buffers)
                side.wait_event(self.ev_chain[name])
            with (torch.cuda.stream(side) if side is not None else _NO_CTX):
                for tap in (3, 2, 1, 0):
                    self.bwd_head[tap].replay()
                    if side is not None:
                        self.ev_bh[tap].record(side)
                for gw in self.wgrad.get(name, ()):
                    gw.replay()
                if side is not None:
                    self.ev_wgrad[name].record(side)
        else:
            tap = self.tap_of[name]
            if side is not None and tap is not None:
                main.wait_event(self.ev_bh[tap])
            self.bwd[name].replay()
            if name in self.wgrad:
                if side is None:
                    for gw in self.wgrad[name]:
                        gw.replay()
                else:
                    self.ev_chain[name].record(main)
                    side.wait_event(self.ev_chain[name])
                    with torch.cuda.stream(side):
                        for gw in self.wgrad[name]:
                            gw.replay()
                        self.ev_wgrad[name].record(side)
        self.enc.__dict__["_pending"].setdefault(name, []).append(self)

    def wait_wgrad(self, name: str):
        """The current stream waits for this plan's weight-gradient graph of ``name`` (a no-op without one)."""
        if self.side is not None and (name in self.wgrad or name == "heads"):
            torch.cuda.current_stream(self.static_img.device).wait_event(self.ev_wgrad[name])


def _deliver(params, grads):
    """What autograd's AccumulateGrad does for a leaf, then its post-accumulate-grad hooks (GradBucketer's among them)."""
    acc_dst, acc_src = [], []
    for p, g in zip(params, grads):
        if g is None:
            continue
        if p.grad is None:
            p.grad = g.detach()
        else:
            acc_dst.append(p.grad), acc_src.append(g)
    if acc_dst:
        torch._foreach_add_(acc_dst, acc_src)
    for p, g in zip(params, grads):
        hooks = getattr(p, "_post_accumulate_grad_hooks", None)
        if g is not None and hooks:
            for h in list(hooks.values()):
                h(p)


class _Lease:
    """A plan is in flight from its forward until the backward of its first segment has run -- or until the autograd graph
    that holds this object is freed without a backward."""

    def __init__(self, plan):
        self.plan = plan

    def release(self):
        if self.plan is not None:
            self.plan.busy = False
            self.plan = None

    def __del__(self):
        self.release()


class _SegFn(torch.autograd.Function):
    """One segment of a plan as an autograd node.  The nodes of a plan are chained by a token (stem -> layer1 -> layer2 -> layer3_* -> layer4 ->
    heads; the heads' node returns the real outputs), so autograd runs their backwards last segment first; the data itself
    moves between the captured graphs in their static buffers.  Every node also takes the encoder's HUB leaf of its segment:
    autograd runs a leaf's AccumulateGrad once per backward pass, after the LAST node that uses it -- with several forwards
    in flight (the trainer calls the encoder once per frame of a clip and backpropagates once, trainer.py:95-131) that is
    the moment all of them have produced this segment's parameter gradients, and the hub's hook hands their sum over."""

    @staticmethod
    def forward(ctx, plan, name, token, hub):
        ctx.plan, ctx.name, ctx.has_token = plan, name, token is not None
        ctx.lease = _Lease(plan) if name == plan.enc.segments[0] else None
        ctx.set_materialize_grads(False)
        if name != "heads":
            return plan.token.detach()
        outs = tuple(o.detach() for o in plan.heads)
        ctx.mark_non_differentiable(*[o for o, g in zip(outs, plan.gout) if g is None])
        return outs

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *grads):
        plan, name = ctx.plan, ctx.name
        plan.backward_segment(name, [g for g, s in zip(grads, plan.gout) if s is not None] if name == "heads" else ())
        if ctx.lease is not None:
            ctx.lease.release()
        return None, None, (plan.zero if ctx.has_token else None), plan.zero
