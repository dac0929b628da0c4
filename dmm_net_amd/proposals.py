"""Proposal preprocessing on the device: the step right before the matching path.

Counterparts of the reference's host-side Python loops:

* ``paste_masks``     -- ``Masker.forward_single_image`` / ``paste_mask_in_image`` / ``binmask_to_box``
                         (dmm/utils/masker.py:110-215): 28x28 mask probabilities + boxes -> the soft ``[P,1,H,W]``
                         masks the matching layer consumes + tight boxes (one HIP workgroup per proposal).
* ``nms`` / ``filter_results`` -- ``filter_results`` (dmm/utils/boxlist_ops.py:15-29): NMS(thresh) + top-k per image.

BoxLists are duck-typed (``bbox``, ``size``, ``fields()``, ``get_field``); ``SimpleBoxList`` is a minimal stand-in.
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import torch

from . import _lib


class SimpleBoxList:
    """The subset of maskrcnn_benchmark's BoxList this package touches."""

    def __init__(self, bbox: torch.Tensor, size, mode: str = "xyxy"):
        self.bbox, self.size, self.mode = bbox, size, mode
        self.extra_fields: Dict[str, torch.Tensor] = {}

    def __len__(self):
        return self.bbox.shape[0]

    def add_field(self, k, v):
        self.extra_fields[k] = v

    def get_field(self, k):
        return self.extra_fields[k]

    def fields(self):
        return list(self.extra_fields.keys())

    def __getitem__(self, idx):
        out = SimpleBoxList(self.bbox[idx], self.size, self.mode)
        for k, v in self.extra_fields.items():
            out.add_field(k, v[idx])
        return out

    def to(self, device):
        out = SimpleBoxList(self.bbox.to(device), self.size, self.mode)
        for k, v in self.extra_fields.items():
            out.add_field(k, v.to(device) if hasattr(v, "to") else v)
        return out

    def resize(self, size):
        """BoxList.resize of maskrcnn_benchmark for xyxy boxes: ``size`` = (width, height); tensor fields are kept."""
        rw, rh = float(size[0]) / float(self.size[0]), float(size[1]) / float(self.size[1])
        if rw == rh:
            bbox = self.bbox * rw
        else:
            bbox = self.bbox * self.bbox.new_tensor([rw, rh, rw, rh])
        out = SimpleBoxList(bbox, tuple(size), self.mode)
        for k, v in self.extra_fields.items():
            out.add_field(k, v if isinstance(v, torch.Tensor) else v.resize(size))
        return out


def paste_masks(mask_prob: torch.Tensor, boxes: torch.Tensor, im_h: int, im_w: int, thresh: float = 0.4,
                padding: int = 1, want_packed: bool = False):
    """mask_prob [P,1,M,M] (or [P,M,M]), boxes [P,4] xyxy -> (masks [P,1,im_h,im_w], tight boxes [P,4]).
    With ``want_packed`` also returns the 1-bit (mask > 0.5) planes [P, words] (int64, ``ops.iou_counts_packed``)."""
    if not mask_prob.is_cuda:
        raise _lib.DmmError("paste_masks needs tensors on an MI355X device (no CPU fallback)")
    prob = mask_prob.reshape(mask_prob.shape[0], mask_prob.shape[-2], mask_prob.shape[-1]).contiguous().float()
    boxes = boxes.contiguous().float()
    P, M = prob.shape[0], prob.shape[-1]
    assert prob.shape[-2] == M and boxes.shape == (P, 4)
    planes = torch.empty((P, 1, im_h, im_w), dtype=torch.float32, device=prob.device)
    nb = torch.empty((P, 4), dtype=torch.float32, device=prob.device)
    packed = None
    if want_packed:
        packed = torch.empty((P, 4 * ((im_h * im_w + 255) // 256)), dtype=torch.int64, device=prob.device)
    with _lib.device_guard(prob.device):
        rc = _lib.load().dmm_paste_masks_f32(prob.data_ptr(), P, M, boxes.data_ptr(), int(im_h), int(im_w), float(thresh),
                                             int(padding), planes.data_ptr(), im_h * im_w, nb.data_ptr(),
                                             None if packed is None else packed.data_ptr(),
                                             torch.cuda.current_stream(prob.device).cuda_stream)
    _lib.check(rc, "dmm_paste_masks_f32")
    return (planes, nb, packed) if want_packed else (planes, nb)


def nms_batched(boxes: Sequence[torch.Tensor], scores: Sequence[torch.Tensor], thresh: float, max_keep: int = 0):
    """Per-image NMS in one launch -> list of kept index tensors (descending score order)."""
    dev = boxes[0].device
    counts = [int(b.shape[0]) for b in boxes]
    offs = _lib.small_to_device([sum(counts[:i]) for i in range(len(counts) + 1)], torch.int32, dev)
    allb = torch.cat([b.float() for b in boxes], 0).contiguous()
    alls = torch.cat([s.float() for s in scores], 0).contiguous()
    keep = torch.empty((max(int(allb.shape[0]), 1),), dtype=torch.int32, device=dev)
    cnt = torch.empty((len(boxes),), dtype=torch.int32, device=dev)
    with _lib.device_guard(dev):
        rc = _lib.load().dmm_nms_f32(allb.data_ptr(), alls.data_ptr(), offs.data_ptr(), len(boxes), max(counts + [0]),
                                     float(thresh), int(max_keep), keep.data_ptr(), cnt.data_ptr(),
                                     torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(rc, "dmm_nms_f32")
    cnt_h = cnt.tolist()
    out, o = [], 0
    for i, n in enumerate(counts):
        out.append(keep[o:o + cnt_h[i]].long())
        o += n
    return out


def filter_results(boxlists: List, nms_thresh: float = 0.8, max_proposals: int = 0, score_field: str = "scores"):
    """boxlist_ops.py:15-29 -- NMS on the (tight) boxes, keep the top ``max_proposals``; mutates and returns the list."""
    keeps = nms_batched([b.bbox for b in boxlists], [b.get_field(score_field) for b in boxlists], nms_thresh,
                        max_proposals if max_proposals > 0 else 0)
    for i, k in enumerate(keeps):
        boxlists[i] = boxlists[i][k]
    return boxlists


def forward_mask_prop(mask_prob: Sequence[torch.Tensor], boxlists: Sequence, thresh: float = 0.4, padding: int = 1,
                      want_packed: bool = False):
    """MaskPostProcessor.forward_mask_prop (masker.py:27-50): paste every image's masks, re-box tightly, carry the fields.
    Images of one size (the usual case: one clip) go through ONE paste launch; the per-image planes are views of it.
    ``want_packed``: the paste kernel also emits the 1-bit (mask > 0.5) planes as field 'mask_packed' ([P, words]
    int64) -- all the cost pass ever looks at; ``DMM_Model.inference`` then counts on 1/32 of the proposal bytes."""
    sizes = {tuple(bl.size) for bl in boxlists}
    counts = [len(bl) for bl in boxlists]
    if len(sizes) == 1 and len(boxlists) > 1 and sum(counts) > 0:
        im_w, im_h = boxlists[0].size
        res = paste_masks(torch.cat(list(mask_prob), 0), torch.cat([bl.bbox for bl in boxlists], 0), im_h, im_w,
                          thresh, padding, want_packed=want_packed)
        per_image = list(zip(*[t.split(counts, 0) for t in res]))
    else:
        per_image = [paste_masks(prob, bl.bbox, bl.size[1], bl.size[0], thresh, padding, want_packed=want_packed)
                     for prob, bl in zip(mask_prob, boxlists)]
    out = []
    for res, bl in zip(per_image, boxlists):
        planes, tight = res[0], res[1]
        nb = SimpleBoxList(tight, bl.size, "xyxy")
        for f in bl.fields():
            nb.add_field(f, bl.get_field(f))
        nb.add_field("mask", planes)
        if want_packed:
            nb.add_field("mask_packed", res[2])
        out.append(nb)
    return out
