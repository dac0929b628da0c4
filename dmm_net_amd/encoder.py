"""Encoder of the matching path: ResNet body + skip / proposal heads (reference a10).

Counterpart of ``dmm/modules/vision.py:6-38`` (torchvision ``ResNet`` subclasses returning x5..x1),
``dmm/modules/base.py:18-69`` (``FeatureExtractorBase``: ``sk2..5`` + ``bn2..5`` skip projections for the decoder,
``prop2..5`` two-conv heads that feed the ROI feature extractor) and the feature part of
``dmm/modules/model_encoder.py:86-162`` (``forward_base`` + the head calls :136-146).

The convolutions are plain ``torch.nn`` modules: on ROCm they run on MIOpen (the only MFMA work on the whole path
-- the matching kernels are bandwidth / latency bound and never touch the matrix cores).  For inference
``fold_batchnorm`` + ``GraphedEncoder`` (bf16 autocast, NCHW) is the fast setting on MI355X with the MIOpen of this
image (ResNet-50, 8 frames of 255x255: 3.6 ms eager channels_last bf16 -> 2.5 ms; round 1, LABLOG).  torchvision is not a dependency: the body is
written out here with torchvision's parameter names (``conv1, bn1, layer1..4.N.convK/bnK/downsample.0/1, fc``)
so reference checkpoints (``encoder`` keys, ``dmm/utils/utils.py:57-111``) load with ``load_state_dict``.

Not here ("next"): offline-proposal lookup, mask paste, NMS + top-k
(``model_encoder.py:115-134``) -- proposals are an input of this module's callers.

Parity: un-pinned.  torchvision / pretrained weights are not available offline and the reference's encoder cannot
be imported (needs maskrcnn_benchmark), so tests check structure (parameter counts of the published
architectures, state-dict key names, output strides / channels) and gradient flow only.
"""
from __future__ import annotations

from typing import Dict, Tuple

import contextlib
import os

import torch
import torch.nn as nn

_NO_CTX = contextlib.nullcontext()


def get_skip_dims(model_name: str):
    """dmm/utils/utils.py:249-256."""
    if model_name in ("resnet50", "resnet101", "coco"):
        return [2048, 1024, 512, 256, 64]
    if model_name == "resnet34":
        return [512, 256, 128, 64, 64]
    if model_name == "vgg16":
        return [512, 512, 256, 128, 64]
    raise Exception("The base model you chose is not supported ! {}".format(model_name))


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return self.relu(out + idt)


class Bottleneck(nn.Module):
    """torchvision's v1.5 bottleneck: the stride sits on the 3x3 convolution."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        return self.relu(out + idt)


class ResNetBody(nn.Module):
    """ResNet returning (x5, x4, x3, x2, x1) like the reference's subclasses (vision.py:11-21)."""

    def __init__(self, block, layers, num_classes=1000):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], 2)
        self.layer3 = self._make_layer(block, 256, layers[2], 2)
        self.layer4 = self._make_layer(block, 512, layers[3], 2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))      # unused by DMM-Net; kept for checkpoint key parity
        self.fc = nn.Linear(512 * block.expansion, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes * block.expansion, 1, stride, bias=False),
                                       nn.BatchNorm2d(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        layers += [block(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    def forward(self, x):
        x1 = self.relu(self.bn1(self.conv1(x)))
        x = self.maxpool(x1)
        x2 = self.layer1(x)
        x3 = self.layer2(x2)
        x4 = self.layer3(x3)
        x5 = self.layer4(x4)
        return x5, x4, x3, x2, x1


def ResNet34():
    return ResNetBody(BasicBlock, [3, 4, 6, 3])


def ResNet50():
    return ResNetBody(Bottleneck, [3, 4, 6, 3])


def ResNet101():
    return ResNetBody(Bottleneck, [3, 4, 23, 3])


class VGG16(nn.Module):
    """The reference's fourth backbone option (``base_model == 'vgg16'``, model_encoder.py:48-49; vision.py:57-115): the
    13 convolutions of VGG-16 (3x3, ReLU, no BatchNorm) in five stages, each closed by a 2x2 max pool; the five taps
    are the POOLED stage outputs -- (x5, x4, x3, x2, x1) at strides 32 / 16 / 8 / 4 / 2 with 512 / 512 / 256 / 128 / 64
    channels (``get_skip_dims('vgg16')``).  ``features`` keeps torchvision's flat conv / ReLU / pool numbering (0..30), so a
    torchvision VGG-16 state dict loads into it; the classifier is not built (DMM-Net never runs it)."""

    STAGES = ((64, 64), (128, 128), (256, 256, 256), (512, 512, 512), (512, 512, 512))

    def __init__(self):
        super().__init__()
        layers, cin, self.taps = [], 3, []
        for widths in self.STAGES:
            for cout in widths:
                layers += [nn.Conv2d(cin, cout, 3, padding=1), nn.ReLU(inplace=True)]
                cin = cout
            layers.append(nn.MaxPool2d(2, 2))
            self.taps.append(len(layers) - 1)                     # index of the stage's pool: 4, 9, 16, 23, 30
        self.features = nn.Sequential(*layers)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
                nn.init.constant_(m.bias, 0)
        # the reference's VGG16 (vision.py:57-76) and torchvision state dicts carry classifier.{0,3,6}.* entries; DMM-Net never
        # runs the classifier and it is not built here: the keys are dropped on load, so a strict load of such a dict works
        self._register_load_state_dict_pre_hook(self._drop_classifier_keys)

    @staticmethod
    def _drop_classifier_keys(state_dict, prefix, *args):
        for k in [k for k in state_dict if k.startswith(prefix + "classifier.")]:
            del state_dict[k]

    def forward(self, x):
        outs = []
        for i, layer in enumerate(self.features):
            x = layer(x)
            if i in self.taps:
                outs.append(x)
        x1, x2, x3, x4, x5 = outs
        return x5, x4, x3, x2, x1


def _prop_head(cin, cmid, cout, k, pad):
    # base.py:43-54: conv -> BN -> ReLU -> conv -> BN
    return nn.Sequential(nn.Conv2d(cin, cmid, k, padding=pad), nn.BatchNorm2d(cmid), nn.ReLU(),
                         nn.Conv2d(cmid, cout, k, padding=pad), nn.BatchNorm2d(cout))


class FeatureEncoder(nn.Module):
    """Body + heads.  ``forward(img [B,3,H,W])`` returns the reference's feature dict (model_encoder.py:157-160):
    ``backbone_feature`` = (p2, p3, p4, p5) (``hidden_size`` channels at strides 4/8/16/32),
    ``refine_input_feat`` = (x5_skip, x4_skip, x3_skip, x2_skip), ``body_feature`` = (x2, x3, x4, x5)."""

    def __init__(self, base_model: str = "resnet50", hidden_size: int = 128, kernel_size: int = 3):
        super().__init__()
        from . import use_shipped_miopen_db
        use_shipped_miopen_db()              # here, not at import: only a process that builds an encoder gets the find-db
        dims = get_skip_dims(base_model)
        hid, ker = int(hidden_size), int(kernel_size)
        pad = 0 if ker == 1 else 1
        backbones = {"resnet34": ResNet34, "resnet50": ResNet50, "resnet101": ResNet101, "vgg16": VGG16}
        if base_model not in backbones:                          # model_encoder.py:50-51
            raise Exception("The base model you chose is not supported ! {} (have: {})".format(base_model, sorted(backbones)))
        self.base = backbones[base_model]()
        self.sk5 = nn.Conv2d(dims[0], hid, ker, padding=pad)
        self.sk4 = nn.Conv2d(dims[1], hid, ker, padding=pad)
        self.sk3 = nn.Conv2d(dims[2], hid // 2, ker, padding=pad)
        self.sk2 = nn.Conv2d(dims[3], hid // 4, ker, padding=pad)
        self.bn5, self.bn4 = nn.BatchNorm2d(hid), nn.BatchNorm2d(hid)
        self.bn3, self.bn2 = nn.BatchNorm2d(hid // 2), nn.BatchNorm2d(hid // 4)
        self.prop5 = _prop_head(dims[0], hid, hid, ker, pad)
        self.prop4 = _prop_head(dims[1], hid, hid, ker, pad)
        self.prop3 = _prop_head(dims[2], hid // 2, hid, ker, pad)
        self.prop2 = _prop_head(dims[3], hid // 4, hid, ker, pad)

    def get_skip_params(self):
        """base.py:62-69."""
        plist = []
        for p in (self.sk2, self.sk3, self.sk4, self.sk5, self.bn2, self.bn3, self.bn4, self.bn5,
                  self.prop5, self.prop4, self.prop3, self.prop2):
            plist.extend(list(p.parameters()))
        return plist

    def get_backbone_para(self):
        for _, p in self.base.named_parameters():
            if p.requires_grad:
                yield p

    def forward(self, img: torch.Tensor) -> Dict[str, Tuple[torch.Tensor, ...]]:
        assert img.dim() == 4 and img.shape[1] == 3, img.shape           # model_encoder.py:91-92
        x5, x4, x3, x2, _ = self.base(img)
        x5_skip = self.bn5(self.sk5(x5))
        x4_skip = self.bn4(self.sk4(x4))
        x3_skip = self.bn3(self.sk3(x3))
        x2_skip = self.bn2(self.sk2(x2))
        p5, p4, p3, p2 = self.prop5(x5), self.prop4(x4), self.prop3(x3), self.prop2(x2)
        return {"backbone_feature": (p2, p3, p4, p5),
                "refine_input_feat": (x5_skip, x4_skip, x3_skip, x2_skip),
                "body_feature": (x2, x3, x4, x5)}


# ---- inference-time BatchNorm folding ---------------------------------------------------------------------------
def _fold_pair(conv: nn.Conv2d, bn: nn.BatchNorm2d) -> nn.Conv2d:
    """conv -> BN(eval) == one conv with W' = W * g / sqrt(var + eps), b' = (b - mean) * g / sqrt(var + eps) + beta."""
    s = bn.weight.detach() / torch.sqrt(bn.running_var.detach() + bn.eps)
    out = nn.Conv2d(conv.in_channels, conv.out_channels, conv.kernel_size, conv.stride, conv.padding, conv.dilation,
                    conv.groups, bias=True).to(device=conv.weight.device, dtype=conv.weight.dtype)
    b = conv.bias.detach() if conv.bias is not None else torch.zeros_like(bn.running_mean)
    with torch.no_grad():
        out.weight.copy_(conv.weight.detach() * s.view(-1, 1, 1, 1))
        out.bias.copy_((b - bn.running_mean.detach()) * s + bn.bias.detach())
    return out


def fold_batchnorm(encoder: "FeatureEncoder") -> "FeatureEncoder":
    """Deep copy of an ``eval()`` encoder with every BatchNorm folded into the convolution before it (the body's
    53 / 104 conv-BN pairs, the four ``sk``/``bn`` skips, both convolutions of the four ``prop`` heads).  Same
    outputs up to fp32 rounding, about half the launches: BatchNorm inference + bias kernels were ~15 % of the
    encoder's GPU time on MI355X.  Inference only (the copy has no BatchNorm state left to train)."""
    import copy
    assert not encoder.training, "fold_batchnorm needs eval() mode (running statistics)"
    enc = copy.deepcopy(encoder)
    body = enc.base
    if not isinstance(body, VGG16):                          # (the VGG body has no BatchNorm: only the heads fold)
        body.conv1, body.bn1 = _fold_pair(body.conv1, body.bn1), nn.Identity()
        for layer in (body.layer1, body.layer2, body.layer3, body.layer4):
            for blk in layer:
                for i in (1, 2, 3):
                    if hasattr(blk, f"conv{i}"):
                        setattr(blk, f"conv{i}", _fold_pair(getattr(blk, f"conv{i}"), getattr(blk, f"bn{i}")))
                        setattr(blk, f"bn{i}", nn.Identity())
                if blk.downsample is not None:
                    blk.downsample = nn.Sequential(_fold_pair(blk.downsample[0], blk.downsample[1]), nn.Identity())
    for k in (5, 4, 3, 2):
        setattr(enc, f"sk{k}", _fold_pair(getattr(enc, f"sk{k}"), getattr(enc, f"bn{k}")))
        setattr(enc, f"bn{k}", nn.Identity())
        head = getattr(enc, f"prop{k}")
        setattr(enc, f"prop{k}", nn.Sequential(_fold_pair(head[0], head[1]), nn.Identity(), head[2],
                                                _fold_pair(head[3], head[4]), nn.Identity()))
    return enc.eval()


class GraphedEncoder:
    """Inference-time replay of ``FeatureEncoder.forward`` from one captured HIP graph per input shape.

    At the product's batch sizes the ResNet forward is launch bound on MI355X (ResNet-50, 8 frames of 255x255: ~160
    MIOpen / elementwise launches, 3.9 ms eager against well under 1 ms of arithmetic); replaying a captured graph
    removes the per-launch host cost.  The encoder must be in ``eval()`` mode (BatchNorm running statistics; nothing
    in the graph may depend on host state).  Outputs are views of the graph's static buffers: consume (or clone) them
    before the next call.  ``torch.backends.cudnn.benchmark = True`` before the first call lets MIOpen search its
    solvers during the warm-up (ResNet-50, 8 frames: 36 s once, 2.21 -> 2.08 ms per replay; round 2, LABLOG).  ``autocast_dtype=torch.bfloat16`` captures the bf16 path of BASELINE config 3.
    """

    static_outputs = True      # outputs alias the graph's buffers (video.FrameLoop clones what it keeps across frames)

    def __init__(self, encoder: "FeatureEncoder", autocast_dtype=None, warmup: int = 3, weights_dtype=None,
                 miopen_find: bool = False):
        """``weights_dtype=torch.bfloat16`` converts the (BatchNorm-folded) encoder's parameters ONCE and runs the whole
        forward in that dtype; ``autocast_dtype`` keeps fp32 parameters and lets autocast re-cast all of them on every
        replay (131 cast kernels per ResNet-50 forward, ~15 % of the graph).  Use one or the other."""
        assert not encoder.training, "capture needs eval() mode"
        assert autocast_dtype is None or weights_dtype is None
        if weights_dtype is not None:
            assert not isinstance(encoder, FastEncoder), "FastEncoder carries its own bf16 weights"
            assert not any(isinstance(m, nn.BatchNorm2d) for m in encoder.modules()), \
                "fold_batchnorm() first: BatchNorm statistics should not be rounded to a 16-bit type"
            encoder = encoder.to(weights_dtype)
        self.encoder, self.dtype, self.warmup, self.wdtype = encoder, autocast_dtype, int(warmup), weights_dtype
        # miopen_find: run the warm-up of every new input shape with torch.backends.cudnn.benchmark = True, i.e. let
        # MIOpen time all applicable solvers once per convolution shape instead of taking its heuristic pick (ResNet-50
        # at 8 x 255 x 255, channels-last bf16: device time per forward 1.61 -> 1.25 ms, the search replaces the
        # split-K implicit-GEMM kernels and their cast / zero helpers by CK kernels).  The results land in MIOpen's
        # user find-db: dmm_net_amd/miopen_db ships the entries of BASELINE configs 3 and 4, so on those shapes the
        # "search" is a lookup; a new shape costs a one-time search of some tens of seconds.
        self.miopen_find = bool(miopen_find)
        self._graphs = {}

    def _forward(self, x):
        if self.wdtype is not None:
            with torch.no_grad():
                return self.encoder(x.to(self.wdtype))
        with torch.no_grad(), torch.autocast("cuda", dtype=self.dtype or torch.bfloat16, enabled=self.dtype is not None):
            return self.encoder(x)

    def __call__(self, img: torch.Tensor):
        if not img.is_cuda:
            raise RuntimeError("GraphedEncoder needs the images on the MI355X")
        key = (tuple(img.shape), img.dtype, img.is_contiguous(memory_format=torch.channels_last), img.device.index)
        entry = self._graphs.get(key)
        if entry is None:
            static_in = img.clone(memory_format=torch.preserve_format)
            side = torch.cuda.Stream(device=img.device)
            side.wait_stream(torch.cuda.current_stream(img.device))
            with torch.cuda.stream(side):                         # warm-up off the capture: MIOpen picks its kernels
                old = torch.backends.cudnn.benchmark
                torch.backends.cudnn.benchmark = old or self.miopen_find
                try:
                    for _ in range(self.warmup):
                        self._forward(static_in)
                finally:
                    torch.backends.cudnn.benchmark = old
            torch.cuda.current_stream(img.device).wait_stream(side)
            if isinstance(self.encoder, FastEncoder) and self.encoder.heads_overlap:
                entry = self._graphs[key] = self._capture_levels(static_in)
            else:
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                    static_out = self._forward(static_in)
                entry = self._graphs[key] = (graph, static_in, static_out)
        graph, static_in, static_out = entry
        static_in.copy_(img)
        if isinstance(graph, tuple):
            main = torch.cuda.current_stream(img.device)
            side = self.encoder._heads_stream(img.device, main)
            for body, heads in zip(*graph):
                body.replay()
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    heads.replay()
            for tail in graph[0][len(graph[1]):]:
                tail.replay()
            main.wait_stream(side)
        else:
            graph.replay()
        return static_out

    def _capture_levels(self, static_in):
        """``FastEncoder`` as 4 body graphs (stem + layer1, layer2, layer3, layer4; replayed on the caller's stream) and 4
        head graphs (``prop_k`` + ``sk_k`` of one level; replayed on the encoder's side stream as soon as the level is
        done; ``sk_5`` is a fifth graph on the caller's stream beside ``prop_5``).  At the product's batch sizes every convolution is a 5-30 us launch that fills a fraction of the 256 CUs and
        the 12 head convolutions were a serial ~0.28 ms tail behind layer4; a single graph does not run its branches side
        by side, two streams do.  Each chain has its own memory pool (graphs that share a pool must never overlap)."""
        fast = self.encoder
        pools = torch.cuda.graph_pool_handle(), torch.cuda.graph_pool_handle()
        body, heads, feats, props, skips = [], [], [], [], []
        x = static_in
        for i, k in enumerate(fast.LEVELS):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=pools[0], capture_error_mode="thread_local"), torch.no_grad():
                x = fast._level(i, fast._stem(static_in) if i == 0 else x)
            h = torch.cuda.CUDAGraph()
            with torch.cuda.graph(h, pool=pools[1], capture_error_mode="thread_local"), torch.no_grad():
                p = fast._prop(k, x, on_side=True)
                if k != fast.LEVELS[-1]:
                    sk = fast._skip(k, x, on_side=True)
            body.append(g), heads.append(h), feats.append(x), props.append(p)
            if k == fast.LEVELS[-1]:                     # nothing left to hide under: the last pair is split over both
                g = torch.cuda.CUDAGraph()               # streams (this graph follows the fork on the caller's stream)
                with torch.cuda.graph(g, pool=pools[0], capture_error_mode="thread_local"), torch.no_grad():
                    sk = fast._skip(k, x)
                body.append(g)
            skips.append(sk)
        return (tuple(body), tuple(heads)), static_in, fast._pack(feats, props, skips)


def _streams_overlap(main, cand, cycles: int = 400_000) -> bool:
    """True when a kernel on ``cand`` runs BESIDE a kernel on ``main`` (the two streams sit on different hardware queues):
    a spin kernel on each, forked and joined by events, against one spin kernel alone."""
    def timed(both):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(main)
        if both:
            cand.wait_event(e0)
            with torch.cuda.stream(cand):
                torch.cuda._sleep(cycles)
        with torch.cuda.stream(main):
            torch.cuda._sleep(cycles)
        if both:
            main.wait_stream(cand)
        e1.record(main)
        e1.synchronize()
        return e0.elapsed_time(e1)
    timed(True)
    alone = min(timed(False), timed(False))
    return min(timed(True), timed(True)) < 1.5 * alone


def pick_parallel_stream(dev, others, priority: int = 0, tries: int = 8):
    """A stream whose kernels run BESIDE those of every stream in ``others``.  HIP maps streams onto a few hardware
    queues (4 per priority level by default) as they are created, and two streams that share a queue run one after the
    other: whether a "side stream" overlaps anything depends on what else the process has created.  torch hands out
    streams from a pool per priority in turn, so successive candidates sit on successive queues; each is probed
    (``_streams_overlap``).  (High-priority streams are NOT the way out: with the heads or the loop's encoder on a
    priority -1 stream a stand-alone process ran the config-3 forward in 4.1 ms instead of 0.95 -- measured on three
    boxes -- although the same code inside a process with more streams ran at full speed.)"""
    cand = None
    for _ in range(tries):
        cand = torch.cuda.Stream(device=dev, priority=priority)
        if all(_streams_overlap(o, cand) for o in others):
            return cand
    return cand


# ---- channels-last inference encoder: 1x1 convolutions as hipBLASLt GEMMs, fused epilogues ----------------------
def _bias_act_(x: torch.Tensor, bias, residual=None, relu: bool = True) -> torch.Tensor:
    """In place on a channels-last bf16 activation [B,C,H,W] (memory [B,H,W,C]): x = act(x + bias[c] (+ residual)) --
    ONE HIP launch (``dmm_bias_act_bf16``) instead of eager's bias add + residual add + clamp."""
    from . import _lib
    assert x.is_cuda and x.dtype == torch.bfloat16 and x.is_contiguous(memory_format=torch.channels_last)
    B, C, H, W = x.shape
    if residual is not None:
        assert residual.shape == x.shape and residual.dtype == x.dtype
        assert residual.is_contiguous(memory_format=torch.channels_last)
    with _lib.device_guard(x.device):
        rc = _lib.load().dmm_bias_act_bf16(x.data_ptr(), None if bias is None else bias.data_ptr(),
                                           None if residual is None else residual.data_ptr(), B * H * W, C, int(relu),
                                           torch.cuda.current_stream(x.device).cuda_stream)
    _lib.check(rc, "dmm_bias_act_bf16")
    return x


def _as_rows(x: torch.Tensor) -> torch.Tensor:
    """channels-last [B,C,H,W] -> its [B*H*W, C] matrix (a view: the memory already is that matrix)."""
    B, C, H, W = x.shape
    return x.permute(0, 2, 3, 1).reshape(B * H * W, C)


def _from_rows(y: torch.Tensor, B: int, H: int, W: int) -> torch.Tensor:
    return y.view(B, H, W, y.shape[1]).permute(0, 3, 1, 2)          # channels-last [B,C,H,W] view


class FastEncoder(nn.Module):
    """Inference form of ``FeatureEncoder`` for MI355X (BASELINE config 3: bf16).  Same function as
    ``fold_batchnorm(encoder)`` -- same parameters, BatchNorm folded -- evaluated differently:

    * activations stay channels-last bf16 end to end.  In NCHW, MIOpen wraps every implicit-GEMM convolution in a
      pair of layout transposes + cast / zero helper kernels (112 + 56 of the 349 launches of one ResNet-50 forward at
      8 x 255 x 255, 27 % of its device time; profiles/r02_encoder_kernel_table_nchw_eager.md);
    * the 1x1 convolutions (2/3 of a bottleneck) ARE matrix products of the [B*H*W, Cin] activation matrix: they go to
      hipBLASLt through ``torch.addmm`` / ``torch._addmm_activation`` with the bias (+ ReLU) in the GEMM epilogue;
    * what is left after the 3x3 / 7x7 MIOpen convolutions -- bias, residual, ReLU -- is one in-place HIP launch
      (``dmm_bias_act_bf16``) instead of two or three eager ones.

    All outputs stay channels-last views (``backbone_feature`` is read in place by the NHWC form of the fused ROIAlign
    kernel).  Wrap in ``GraphedEncoder(FastEncoder(enc))`` to replay from one HIP
    graph.  Reference: vision.py:6-38 (body), base.py:35-54 + model_encoder.py:136-146 (heads)."""

    fused_gemm = True          # 1x1 convolutions through dmm_conv1x1_bf16 (False: torch.mm / addmm + the epilogue kernel)
    # The `sk` / `prop` heads of a level only need that level's body output: they are issued on a SIDE stream as soon as
    # the level is done and run under the deeper levels of the body (at the product's batch sizes every convolution is a
    # 5-30 us launch that fills a fraction of the 256 CUs; the 12 head convolutions are ~30 % of the forward's kernel time
    # and were a serial tail after layer4).  Captured in a HIP graph the fork / join become graph edges.
    stem_fused = True          # conv1 + bias + ReLU + max pool tail as one epilogue kernel (dmm_bias_relu_maxpool_bf16)
    heads_overlap = True       # class attributes, set in code by the A/B tools (no environment switches in the product)

    def __init__(self, encoder: "FeatureEncoder", dtype=torch.bfloat16):
        super().__init__()
        from . import use_shipped_miopen_db
        use_shipped_miopen_db()
        assert not encoder.training, "FastEncoder is an inference form: eval() first"
        if not isinstance(encoder.base, ResNetBody):
            raise NotImplementedError("FastEncoder is the bf16 channels-last form of the ResNet bodies (resnet34 / 50 / "
                                      "101); a vgg16 encoder runs as FeatureEncoder / GraphedEncoder(FeatureEncoder)")
        enc = encoder if not any(isinstance(m, nn.BatchNorm2d) for m in encoder.modules()) else fold_batchnorm(encoder)
        assert dtype == torch.bfloat16, "the fused epilogue kernel is bf16"
        self.dtype = dtype
        self.src = enc                       # folded fp32 parameters stay the source of truth (state_dict)
        self._p = {}                         # id(conv) -> prepared (weight, fp32 bias, bf16 bias)
        self._ws = {}                        # scratch of the library GEMMs, one per stream that runs them
        self._side = {}                      # device index -> the heads' side stream
        self._on_side = False                # the call being issued belongs to the heads' stream
        self.avoid_streams = []              # streams of the caller the heads should not share a hardware queue with
        self._prepare()
        # the prepared tensors are derived state: rebuilt whenever the module moves (.to / .cuda) or loads weights
        self.register_load_state_dict_post_hook(lambda module, incompatible: module._prepare())
        self.eval()

    def _prepare(self):
        """bf16 weights in the layouts the kernels take + fp32 biases, from ``self.src`` (the folded fp32 parameters)."""
        dtype = self.dtype
        self._p = {}
        for m in self.src.modules():
            if not isinstance(m, nn.Conv2d):
                continue
            w = m.weight.detach()
            b = (m.bias.detach() if m.bias is not None else w.new_zeros((w.shape[0],))).float().contiguous()
            if m.kernel_size == (1, 1) and m.groups == 1:
                assert m.padding == (0, 0), "a padded 1x1 convolution is not a plain matrix product"
                wt = w.reshape(w.shape[0], w.shape[1]).t().contiguous().to(dtype)          # [Cin, Cout]
                self._p[id(m)] = (wt, b, b.to(dtype))
            else:
                col = None
                if self._patch_ok(m):                                  # [(kh, kw, cin), Cout]: rows of the patch matrix
                    col = w.permute(2, 3, 1, 0).reshape(-1, w.shape[0]).contiguous().to(dtype)
                self._p[id(m)] = (w.to(dtype).contiguous(memory_format=torch.channels_last), b, col)
        self._ws = {}
        self._patch_choice = {}              # (conv id, input shape) -> True: patch matrix + GEMM, False: MIOpen

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        if hasattr(self, "src"):
            self._prepare()
        return out

    # -- building blocks --------------------------------------------------------------------------------------
    def _gemm_scratch(self, device, stream):
        """32 MB of split-K scratch for the library GEMMs, one buffer per (device, body / heads role, stream that issues
        the call): the heads' GEMMs run beside the body's, and two callers on two streams must not share one either
        (ADVICE r3).  Under graph capture the stream is the capture stream; the role keeps the body's and the heads'
        graphs apart, which replay side by side."""
        key = (device.index, self._on_side, int(stream.cuda_stream))
        ws = self._ws.get(key)
        if ws is None:
            ws = self._ws[key] = torch.empty((32 << 20,), dtype=torch.uint8, device=device)
        return ws

    def _conv1x1(self, x, conv, relu, residual=None):
        """y = act(x @ W^T + b (+ residual)) on the activation matrix, ONE library GEMM with the whole tail in its
        epilogue (``dmm_conv1x1_bf16``: hipBLASLt, residual as the C operand).  stride 2 = a row subsample first."""
        from . import _lib
        wt, b32, bl = self._p[id(conv)]
        if conv.stride == (2, 2) and x.is_cuda and x.dtype == torch.bfloat16 and x.is_contiguous(memory_format=torch.channels_last):
            from .train_encoder import _subsample                       # 16 bytes per thread (the strided copy below: 2)
            x = _subsample(x, 2)
        elif conv.stride != (1, 1):
            x = x[:, :, ::conv.stride[0], ::conv.stride[1]]
        x = x.contiguous(memory_format=torch.channels_last)
        B, _, H, W = x.shape
        rows = _as_rows(x)
        if self.fused_gemm and rows.is_contiguous():
            stream = torch.cuda.current_stream(x.device)
            ws = self._gemm_scratch(x.device, stream)
            res = None
            if residual is not None:
                res = _as_rows(residual.contiguous(memory_format=torch.channels_last))
            y = torch.empty((rows.shape[0], wt.shape[1]), dtype=self.dtype, device=x.device)
            with _lib.device_guard(x.device):
                rc = _lib.load().dmm_conv1x1_bf16(rows.data_ptr(), wt.data_ptr(), b32.data_ptr(),
                                                  None if res is None else res.data_ptr(), rows.shape[0], wt.shape[0],
                                                  wt.shape[1], int(relu), y.data_ptr(), ws.data_ptr(), ws.numel(),
                                                  stream.cuda_stream)
            if rc == 0:
                return _from_rows(y, B, H, W)
            if rc != 2:                                                # anything but "no kernel for this shape"
                _lib.check(rc, "dmm_conv1x1_bf16")
        if residual is not None:
            # (torch.addmm(residual, rows, wt) would first COPY the residual into the output -- a DtoD memcpy per
            # block, 16 per ResNet-50 forward -- so the residual rides in the epilogue launch instead)
            return _bias_act_(_from_rows(torch.mm(rows, wt), B, H, W), b32, residual, relu)
        if relu:
            y = torch._addmm_activation(bl, rows, wt, use_gelu=False)  # bias + ReLU in the GEMM epilogue
        else:
            y = torch.addmm(bl, rows, wt)
        return _from_rows(y, B, H, W)

    # Opt-in (FastEncoder.patch_mode = "auto", set in code): 3x3 convolutions on SMALL feature maps as patch matrix + library GEMM.  layer3 / layer4
    # and the heads work on 16x16 ... 8x8 maps: the patch matrix is a few MB (one 3-4 us copy kernel) and the product runs
    # with bias (+ residual) + ReLU in its epilogue, against MIOpen's convolution + the separate bias / ReLU pass; both
    # are timed ONCE per (convolution, input shape) on the first call outside a capture and the faster one is kept.
    # Measured on the config-3 encoder with MIOpen's find-db picks: 0.931 -> 0.922 ms per forward -- MIOpen's CK kernels are
    # hard to beat with an explicit patch matrix, so the default stays "miopen" (= MIOpen for every k x k convolution).
    patch_mode = "miopen"
    PATCH_MAX_BYTES = 16 << 20

    @staticmethod
    def _patch_ok(conv):
        return (conv.kernel_size == (3, 3) and conv.padding == (1, 1) and conv.dilation == (1, 1) and conv.groups == 1
                and conv.stride in ((1, 1), (2, 2)) and conv.in_channels % 8 == 0)

    def _conv3x3_patches(self, x, conv, relu, residual=None):
        from . import _lib
        _, b32, wcol = self._p[id(conv)]
        B, C, H, W = x.shape
        s = conv.stride[0]
        Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
        x = x.contiguous(memory_format=torch.channels_last)
        stream = torch.cuda.current_stream(x.device)
        cols = torch.empty((B * Ho * Wo, 9 * C), dtype=self.dtype, device=x.device)
        y = torch.empty((B * Ho * Wo, wcol.shape[1]), dtype=self.dtype, device=x.device)
        ws = self._gemm_scratch(x.device, stream)
        res = None if residual is None else _as_rows(residual.contiguous(memory_format=torch.channels_last))
        with _lib.device_guard(x.device):
            L = _lib.load()
            _lib.check(L.dmm_im2col3x3_bf16(x.data_ptr(), B, H, W, C, s, cols.data_ptr(), stream.cuda_stream),
                       "dmm_im2col3x3_bf16")
            rc = L.dmm_conv1x1_bf16(cols.data_ptr(), wcol.data_ptr(), b32.data_ptr(), None if res is None else res.data_ptr(),
                                    cols.shape[0], cols.shape[1], wcol.shape[1], int(relu), y.data_ptr(), ws.data_ptr(),
                                    ws.numel(), stream.cuda_stream)
        if rc == 2:
            return None                                                # no library kernel for this shape
        _lib.check(rc, "dmm_conv1x1_bf16")
        return _from_rows(y, B, Ho, Wo)

    def _use_patches(self, x, conv, relu, residual):
        if self._p[id(conv)][2] is None or self.patch_mode == "miopen" or not x.is_cuda:
            return False
        B, C, H, W = x.shape
        s = conv.stride[0]
        if B * ((H - 1) // s + 1) * ((W - 1) // s + 1) * 9 * C * 2 > self.PATCH_MAX_BYTES:
            return False
        key = (id(conv), tuple(x.shape))
        pick = self._patch_choice.get(key)
        if pick is None:
            if torch.cuda.is_current_stream_capturing():
                return False                                           # not decided yet and no way to time here
            def timed(fn):
                for _ in range(3):
                    if fn() is None:
                        return float("inf")
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(8):
                    fn()
                e1.record()
                e1.synchronize()
                return e0.elapsed_time(e1)
            t_gemm = timed(lambda: self._conv3x3_patches(x, conv, relu, residual))
            t_lib = timed(lambda: self._conv_miopen(x, conv, relu, residual))
            pick = self._patch_choice[key] = bool(t_gemm < t_lib)
        return pick

    def _conv_miopen(self, x, conv, relu, residual=None):
        w, b32, _ = self._p[id(conv)]
        y = torch.nn.functional.conv2d(x, w, None, conv.stride, conv.padding, conv.dilation, conv.groups)
        return _bias_act_(y, b32, residual, relu)

    def _convkxk(self, x, conv, relu, residual=None):
        if self._use_patches(x, conv, relu, residual):
            y = self._conv3x3_patches(x, conv, relu, residual)
            if y is not None:
                return y
        return self._conv_miopen(x, conv, relu, residual)

    def _conv(self, x, conv, relu, residual=None):
        if conv.kernel_size == (1, 1) and conv.groups == 1:
            return self._conv1x1(x, conv, relu, residual)
        return self._convkxk(x, conv, relu, residual)

    def _block(self, x, blk):
        idt = x if blk.downsample is None else self._conv(x, blk.downsample[0], relu=False)
        if isinstance(blk, Bottleneck):
            out = self._conv(x, blk.conv1, relu=True)
            out = self._conv(out, blk.conv2, relu=True)
            return self._conv(out, blk.conv3, relu=True, residual=idt)
        out = self._conv(x, blk.conv1, relu=True)
        return self._conv(out, blk.conv2, relu=True, residual=idt)

    def _head(self, x, head):
        out = self._conv(x, head[0], relu=True)                        # conv -> (folded BN) -> ReLU
        return self._conv(out, head[3], relu=False)                    # conv -> (folded BN)

    def _heads_stream(self, dev, main=None):
        """The side stream of the heads for work whose body runs on ``main``.  HIP maps streams onto a few hardware queues
        in creation order and two streams that share a queue run one after the other -- then the fork only costs (measured
        inside a process that had created other streams before: config-3 encoder 1.13 ms against 0.95).  So a candidate is
        PROBED once per ``main``: a spin kernel on both streams must take the time of one, not of two."""
        key = (dev.index, main.cuda_stream if main is not None else 0)
        side = self._side.get(key)
        if side is None:
            if main is None or torch.cuda.is_current_stream_capturing():
                side = self._side.get((dev.index, 0)) or torch.cuda.Stream(device=dev)
            else:
                side = pick_parallel_stream(dev, [main] + [o for o in self.avoid_streams if o.device == main.device])
            self._side[key] = side
        return side

    # -- the forward in pieces (GraphedEncoder captures them as separate graphs) -----------------------------------
    LEVELS = (2, 3, 4, 5)

    def _stem(self, img):
        """conv1 -> (folded bn1) -> relu -> maxpool: the bias, the relu and the 3x3 / stride 2 pool are ONE pass over the
        convolution's output (``dmm_bias_relu_maxpool_bf16``, bit identical to the two separate passes)."""
        from . import _lib
        body = self.src.base
        x = img.to(self.dtype).contiguous(memory_format=torch.channels_last)
        mp = body.maxpool
        as_int = lambda v: v if isinstance(v, int) else v[0]
        fusable = (self.stem_fused and x.is_cuda and isinstance(mp, nn.MaxPool2d) and as_int(mp.kernel_size) == 3 and as_int(mp.stride) == 2
                   and as_int(mp.padding) == 1 and as_int(mp.dilation) == 1 and not mp.ceil_mode
                   and body.conv1.out_channels % 8 == 0)
        if not fusable:
            return mp(self._conv(x, body.conv1, relu=True))
        w, b32, _ = self._p[id(body.conv1)]
        c1 = body.conv1
        y = torch.nn.functional.conv2d(x, w, None, c1.stride, c1.padding, c1.dilation, c1.groups)
        B, C, H, W = y.shape
        out = torch.empty((B, C, (H - 1) // 2 + 1, (W - 1) // 2 + 1), dtype=self.dtype, device=y.device,
                          memory_format=torch.channels_last)
        with _lib.device_guard(y.device):
            _lib.check(_lib.load().dmm_bias_relu_maxpool_bf16(y.data_ptr(), b32.data_ptr(), B, H, W, C, out.data_ptr(),
                                                              torch.cuda.current_stream(y.device).cuda_stream),
                       "dmm_bias_relu_maxpool_bf16")
        return out

    def _level(self, i, x):
        for blk in getattr(self.src.base, f"layer{i + 1}"):
            x = self._block(x, blk)
        return x

    def _prop(self, k, x, on_side=False):
        self._on_side = bool(on_side)
        try:
            return self._head(x, getattr(self.src, f"prop{k}"))
        finally:
            self._on_side = False

    def _skip(self, k, x, on_side=False):
        self._on_side = bool(on_side)
        try:
            return self._conv(x, getattr(self.src, f"sk{k}"), relu=False)
        finally:
            self._on_side = False

    @staticmethod
    def _pack(feats, props, skips):
        # channels-last as they are: the NHWC form of the fused ROIAlign kernel reads them in place (16-byte lane loads of
        # contiguous channels; four NCHW copies per forward before)
        return {"backbone_feature": tuple(props), "refine_input_feat": tuple(reversed(skips)), "body_feature": tuple(feats)}

    def forward(self, img: torch.Tensor) -> Dict[str, Tuple[torch.Tensor, ...]]:
        assert img.dim() == 4 and img.shape[1] == 3, img.shape
        with torch.no_grad():
            # fork / join only outside a capture: one HIP graph replays its branches no faster than in sequence (measured:
            # config-3 encoder 1.05 ms with the fork captured, 1.02 without); GraphedEncoder overlaps the heads with
            # separate graphs on two streams instead
            fork = self.heads_overlap and img.is_cuda and not torch.cuda.is_current_stream_capturing()
            main = torch.cuda.current_stream(img.device) if fork else None
            side = self._heads_stream(img.device, main) if fork else None
            x = self._stem(img)
            feats, skips, props = [], [], []
            for i, k in enumerate(self.LEVELS):
                x = self._level(i, x)
                feats.append(x)
                last = k == self.LEVELS[-1]                             # nothing left to hide under: split the pair
                if fork:
                    side.wait_stream(main)                              # this level's heads under the deeper levels
                with (torch.cuda.stream(side) if fork else _NO_CTX):
                    props.append(self._prop(k, x, on_side=fork))
                    if not last:
                        skips.append(self._skip(k, x, on_side=fork))
                if last:
                    skips.append(self._skip(k, x))
            if fork:
                main.wait_stream(side)
                for t in props + skips:                                 # side-stream allocations consumed on `main`
                    t.record_stream(main)
        return self._pack(feats, props, skips)
