"""ROI feature extractor of the matching path (reference a9).

Counterpart of ``dmm/modules/feature_extractor.py:6-62`` (``FeatureExtractor`` /
``make_roi_mask_feature_extractor``): every proposal box is ROIAligned (14x14 bins, sampling ratio 2, legacy
maskrcnn_benchmark semantics) on ALL four levels of ``backbone_feature`` (strides 4, 8, 16, 32) and the pooled maps are
spatially averaged, giving one ``[4*C]`` vector per box.  Here that is ONE fused HIP kernel per direction
(``dmm_roialign4_mean_fwd`` / ``_bwd``); the reference's ``[R, 4, C, 14, 14]`` intermediate is never formed.

``proposals`` are duck-typed BoxLists: ``len(p)`` and ``p.bbox`` ([P,4] xyxy in image coordinates).
Parity: no upstream fixture exists for this third-party op; forward and gradients are pinned against G12, an independent
differentiable formulation of the published per-bin definition (tests/golden/gen_golden.py), and the oracle.
"""
from __future__ import annotations

import ctypes
from typing import Sequence

import torch
import torch.nn as nn

from . import _lib
from .ops import _DT

SCALES = (0.25, 0.125, 0.0625, 0.03125)      # feature_extractor.py:13


def convert_to_roi_format(boxes: Sequence) -> torch.Tensor:
    """feature_extractor.py:32-37: [R,5] = (batch index, x1, y1, x2, y2).  Three device ops whatever the batch size
    (the per-image full + cat of the reference costs ~17 launches at 8 images, more than the ROI kernel itself)."""
    bbs = [(b.bbox if hasattr(b, "bbox") else b) for b in boxes]
    allb = (torch.cat(bbs, dim=0) if len(bbs) > 1 else bbs[0]).float()
    ids = _lib.small_to_device([float(i) for i, bb in enumerate(bbs) for _ in range(bb.shape[0])], torch.float32, allb.device)
    return torch.cat([ids.unsqueeze(1), allb], dim=1)


def _layout(feats):
    """'nhwc' when every level is a channels_last tensor (memory [B,H,W,C]) the NHWC kernel accepts, else 'nchw'."""
    C = int(feats[0].shape[1])
    vec = 4 if feats[0].dtype == torch.float32 else 8
    lpc = C // vec
    ok = C % vec == 0 and ((lpc & (lpc - 1)) == 0 if lpc <= 64 else lpc % 64 == 0)
    if ok and all(f.is_contiguous(memory_format=torch.channels_last) and not f.is_contiguous() and f.data_ptr() % 16 == 0
                  for f in feats):
        return "nhwc"
    return "nchw"


def _arrays(feats):
    Hs = (ctypes.c_int * 4)(*[int(f.shape[2]) for f in feats])
    Ws = (ctypes.c_int * 4)(*[int(f.shape[3]) for f in feats])
    sc = (ctypes.c_float * 4)(*SCALES)
    return Hs, Ws, sc


def roialign4_mean_into(rois: torch.Tensor, feats, out: torch.Tensor) -> torch.Tensor:
    """Inference form with a caller-owned result: rois [R,5] fp32 (image index < 0 = dead row -> zeros), feats = the four
    NCHW-contiguous levels, out [R, 4*C] fp32.  Nothing is allocated: this is what a captured frame step replays."""
    f0 = feats[0]
    if not f0.is_cuda:
        raise _lib.DmmError("roi features need tensors on an MI355X device (no CPU fallback)")
    B, C, R = int(f0.shape[0]), int(f0.shape[1]), int(rois.shape[0])
    nhwc = _layout(feats) == "nhwc"
    assert all((nhwc or f.is_contiguous()) and f.dtype == f0.dtype and f.shape[:2] == f0.shape[:2] for f in feats)
    assert rois.is_contiguous() and rois.dtype == torch.float32 and out.is_contiguous() and out.shape == (R, 4 * C)
    Hs, Ws, sc = _arrays(feats)
    ptrs = (ctypes.c_void_p * 4)(*[f.data_ptr() for f in feats])
    L = _lib.load()
    with _lib.device_guard(rois.device):
        rc = (L.dmm_roialign4_mean_nhwc_fwd if nhwc else L.dmm_roialign4_mean_fwd)(
            ptrs, _DT[f0.dtype], B, C, Hs, Ws, sc, rois.data_ptr(), R, out.data_ptr(),
            torch.cuda.current_stream(rois.device).cuda_stream)
    _lib.check(rc, "dmm_roialign4_mean_nhwc_fwd" if nhwc else "dmm_roialign4_mean_fwd")
    return out


class _RoiAlign4Mean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rois, f2, f3, f4, f5):
        feats = [f2, f3, f4, f5]
        for f in feats:
            if not f.is_cuda:
                raise _lib.DmmError("roi features need tensors on an MI355X device (no CPU fallback)")
            assert f.dim() == 4 and f.dtype == feats[0].dtype and f.shape[:2] == feats[0].shape[:2]
        # channels-last levels of an inference encoder are read in place by the NHWC kernel; anything else as NCHW
        if not (_layout(feats) == "nhwc" and not any(f.requires_grad for f in feats)):
            feats = [f.contiguous() for f in feats]
        rois = rois.contiguous().float()
        C, R = feats[0].shape[1], rois.shape[0]
        out = roialign4_mean_into(rois, feats, torch.empty((R, 4 * C), dtype=torch.float32, device=rois.device))
        ctx.save_for_backward(rois)
        ctx.shapes = [tuple(f.shape) for f in feats]
        ctx.dtype = feats[0].dtype
        return out

    @staticmethod
    def backward(ctx, dout):
        (rois,) = ctx.saved_tensors
        dout = dout.contiguous().float()
        dfs = [torch.zeros(s, dtype=torch.float32, device=dout.device) for s in ctx.shapes]
        B, C = ctx.shapes[0][0], ctx.shapes[0][1]
        Hs = (ctypes.c_int * 4)(*[s[2] for s in ctx.shapes])
        Ws = (ctypes.c_int * 4)(*[s[3] for s in ctx.shapes])
        sc = (ctypes.c_float * 4)(*SCALES)
        ptrs = (ctypes.c_void_p * 4)(*[d.data_ptr() for d in dfs])
        with _lib.device_guard(dout.device):
            rc = _lib.load().dmm_roialign4_mean_bwd(dout.data_ptr(), B, C, Hs, Ws, sc, rois.data_ptr(), rois.shape[0],
                                                    ptrs, torch.cuda.current_stream(dout.device).cuda_stream)
        _lib.check(rc, "dmm_roialign4_mean_bwd")
        return (None,) + tuple(d.to(ctx.dtype) for d in dfs)


class FeatureExtractor(nn.Module):
    """``forward(backbone_feature, proposals) -> [sum P, 4*C]`` (feature_extractor.py:20-30)."""

    def __init__(self):
        super().__init__()
        self.num_levels = 4
        self.output_size = (14, 14)

    def forward(self, backbone_feature, proposals):
        assert len(backbone_feature) == 4, "four pyramid levels expected (strides 4, 8, 16, 32)"
        rois = convert_to_roi_format(proposals)
        return _RoiAlign4Mean.apply(rois, *backbone_feature)


def make_roi_mask_feature_extractor():
    return FeatureExtractor()
