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


# ------------------------------------------------------------------------------------------------------------------
# two-phase preparation on fixed slots: no plane of a dropped proposal is written, nothing is gathered, no host sync
# ------------------------------------------------------------------------------------------------------------------
class ClipProposals:
    """The RAW proposals of a clip, resident on the device (what ``model_encoder.py:53-58`` loads per frame):
    ``prob`` [T,B,R,M,M] mask probabilities, ``boxes`` [T,B,R,4] xyxy in image coordinates, ``scores`` [T,B,R],
    ``counts`` [T,B] int32 (proposals of video b in frame t; rows past it are padding).  The per-frame kernels take the
    frame index from a device scalar, so a captured graph can walk the clip without host input."""

    def __init__(self, prob, boxes, scores, counts):
        T, B, R = scores.shape
        assert prob.shape[:3] == (T, B, R) and boxes.shape == (T, B, R, 4) and counts.shape == (T, B)
        assert prob.dtype == boxes.dtype == scores.dtype == torch.float32 and counts.dtype == torch.int32
        self.prob, self.boxes, self.scores, self.counts = prob, boxes, scores, counts
        self.T, self.B, self.R, self.M = T, B, R, int(prob.shape[-1])

    @classmethod
    def from_boxlists(cls, proposals: Sequence[Sequence], T: int, im_h: int, im_w: int, device, R: int = 0,
                      out: "ClipProposals" = None):
        """proposals[b][t] BoxLists with the raw 'mask' ([P,1,M,M] or [P,M,M]) and 'scores' | 'objectness'; the last
        entry of a video is reused for missing frames (evaluator.py:101-106); boxes are brought to the image size like
        ``BoxList.resize``.  With ``out`` the clip is written into the first T frames of that (larger) buffer."""
        B = len(proposals)
        items = []
        for t in range(T):
            for b in range(B):
                p = proposals[b][t] if len(proposals[b]) > t else proposals[b][-1]
                if tuple(p.size) != (im_w, im_h):
                    p = p.resize((im_w, im_h))
                items.append(p)
        field = "scores" if "scores" in items[0].fields() else "objectness"
        cnt = [len(p) for p in items]
        Rm = max(cnt + [1, int(R)])
        if out is not None:
            assert out.R >= Rm and out.T >= T and out.B == B
            Rm = out.R
        M = int(items[0].get_field("mask").shape[-1])
        src_dev = items[0].bbox.device

        def pack(get, tail, dtype=torch.float32):
            ts = [get(p).reshape((len(p),) + tail).to(dtype) for p in items]
            if all(c == Rm for c in cnt):
                return torch.stack(ts, 0).view((T, B, Rm) + tail)
            buf = torch.zeros((T * B, Rm) + tail, dtype=dtype, device=src_dev)
            for i, x in enumerate(ts):
                buf[i, :cnt[i]] = x
            return buf.view((T, B, Rm) + tail)
        prob = pack(lambda p: p.get_field("mask"), (M, M))
        boxes = pack(lambda p: p.bbox, (4,))
        scores = pack(lambda p: p.get_field(field), ())
        counts = torch.tensor(cnt, dtype=torch.int32).view(T, B)
        dev = torch.device(device)
        # host-resident inputs (a loaded proposal file) go through pinned staging: an asynchronous copy out of pageable
        # memory may still be reading it when these temporaries are freed
        stage = lambda x: x.pin_memory() if (not x.is_cuda and dev.type == "cuda") else x
        prob, boxes, scores, counts = stage(prob), stage(boxes), stage(scores), stage(counts)
        if out is not None:
            out.prob[:T].copy_(prob, non_blocking=True)
            out.boxes[:T].copy_(boxes, non_blocking=True)
            out.scores[:T].copy_(scores, non_blocking=True)
            out.counts[:T].copy_(counts, non_blocking=True)
            return out
        return cls(prob.to(dev, non_blocking=True), boxes.to(dev, non_blocking=True), scores.to(dev, non_blocking=True),
                   counts.to(dev, non_blocking=True))

    @classmethod
    def empty(cls, T, B, R, M, device):
        f32 = dict(dtype=torch.float32, device=device)
        return cls(torch.zeros((T, B, R, M, M), **f32), torch.zeros((T, B, R, 4), **f32), torch.zeros((T, B, R), **f32),
                   torch.zeros((T, B), dtype=torch.int32, device=device))


class ProposalSlots:
    """K fixed slots per image holding the proposals that survive NMS + top-k, in descending score order: soft planes
    ``planes`` [B,K,H,W], their 1-bit form ``packed`` [B,K,words], tight ``boxes`` [B,K,4], ``scores`` [B,K], the live
    count ``count`` [B] int32 (ON THE DEVICE: it is the ``n_valid`` of the matching kernels) and the ROIAlign rows
    ``rois`` [B*K,5].  Slots past the count are dead (score 0, roi image index -1, stale plane never read)."""

    def __init__(self, B: int, K: int, H: int, W: int, R: int, device, soft_planes: bool = True):
        from .ops import pack_words
        f32 = dict(dtype=torch.float32, device=device)
        i32 = dict(dtype=torch.int32, device=device)
        self.B, self.K, self.H, self.W, self.R = B, K, H, W, R
        # soft_planes=False: only the 1-bit planes are produced (the frame step's epilogue pastes the selected proposals
        # on the fly, dmm_step_finish_f32) -- the soft planes are 10x the bytes of everything else in a frame step
        self.planes = torch.zeros((B, K, H, W), **f32) if soft_planes else None
        self.packed = torch.zeros((B, K, pack_words(H * W)), dtype=torch.int64, device=device)
        self.boxes, self.scores = torch.zeros((B, K, 4), **f32), torch.zeros((B, K), **f32)
        self.rois = torch.zeros((B * K, 5), **f32)
        self.count = torch.zeros((B,), **i32)
        self.tight = torch.zeros((B, R, 4), **f32)               # phase-1 result: tight boxes of ALL raw proposals
        self.keep = torch.zeros((B, K), **i32)


def prepare_slots(clip: ClipProposals, slots: ProposalSlots, nms_thresh: float, mask_thresh: float = 0.4,
                  padding: int = 1, step: torch.Tensor = None, img_base: torch.Tensor = None) -> ProposalSlots:
    """Masker paste + ``filter_results`` + BoxList indexing of one frame of ``clip`` (model_encoder.py:115-134) as three
    launches on fixed slots: tight boxes of every raw proposal (no plane written) -> NMS + top-K -> paste of the kept
    proposals into their slots.  ``step`` (int32 [1] on the device, None = frame 0) selects the frame; ``img_base``
    (int32 [T]) the image index of video 0 in the feature batch the roi rows refer to.  Nothing returns to the host."""
    if not clip.prob.is_cuda:
        raise _lib.DmmError("prepare_slots needs tensors on an MI355X device (no CPU fallback)")
    assert clip.B == slots.B and clip.R == slots.R
    L = _lib.load()
    s = torch.cuda.current_stream(clip.prob.device).cuda_stream
    sp = None if step is None else step.data_ptr()
    with _lib.device_guard(clip.prob.device):
        rc = L.dmm_proposal_boxes_f32(clip.prob.data_ptr(), clip.boxes.data_ptr(), clip.counts.data_ptr(), clip.B, clip.R,
                                      clip.M, slots.H, slots.W, float(mask_thresh), int(padding), sp,
                                      slots.tight.data_ptr(), s)
        _lib.check(rc, "dmm_proposal_boxes_f32")
        rc = L.dmm_nms_slots_f32(slots.tight.data_ptr(), clip.scores.data_ptr(), clip.counts.data_ptr(), clip.B, clip.R,
                                 float(nms_thresh), slots.K, sp, slots.keep.data_ptr(), slots.count.data_ptr(), s)
        _lib.check(rc, "dmm_nms_slots_f32")
        rc = L.dmm_paste_kept_f32(clip.prob.data_ptr(), clip.boxes.data_ptr(), clip.scores.data_ptr(),
                                  slots.tight.data_ptr(), slots.keep.data_ptr(), slots.count.data_ptr(), clip.B, clip.R,
                                  clip.M, slots.K, slots.H, slots.W, int(padding), sp,
                                  None if img_base is None else img_base.data_ptr(),
                                  None if slots.planes is None else slots.planes.data_ptr(),
                                  slots.H * slots.W, slots.packed.data_ptr(), slots.boxes.data_ptr(),
                                  slots.scores.data_ptr(), slots.rois.data_ptr(), s)
        _lib.check(rc, "dmm_paste_kept_f32")
    return slots
