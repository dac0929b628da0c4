"""Per-video frame loop around the matching layer + the data formats on either side of it.

Counterparts in the reference:

* ``mask_boxes`` / ``ohw_mask2boxlist``  -- ``dmm/utils/utils.py:179-210`` (+ ``binmask_to_bbox_xyxy_pt`` :114-143):
  first-frame object masks -> template boxes + ``template_valid``.  One HIP workgroup per plane instead of a
  ``nonzero()`` and four host syncs per object.
* ``merge_labels``  -- the label map of ``dmm/modules/evaluator.py:134-139`` (background = 1 - max, arg-max over
  [bg, objects]); one pass over the planes on the device, one byte per pixel out.
* ``davis_palette`` / ``save_label_png``  -- ``plot_scores_map`` (``dmm/utils/eval_helper.py:22-38``): indexed PNG
  with the DAVIS / PASCAL-VOC palette (the reference reads it from ``dmm/utils/bear/00000.png``; it is the standard
  bit-interleaved colour map, generated here).
* ``load_offline_proposals``  -- the offline proposal files (``predictions.pth`` / ``pred_DICT.pth`` /
  ``videos/<vid>.pth``, written by ``tools/reduce_pth_size_by_videos.py:62-126``, read by
  ``model_encoder.py:53-58``): pickled maskrcnn_benchmark ``BoxList`` objects, mapped onto ``SimpleBoxList``
  without importing maskrcnn_benchmark.
* ``FrameLoop``  -- the frame loop of ``Evaler.forward`` / ``inference_timestep`` (``evaluator.py:63-213``) with every
  video of the batch in one ragged launch per frame and ``mask_hist`` resident on the device.  The decoder
  (ConvLSTM refinement, out of scope for this package) is injected as ``refine``.

Nothing here has a CPU implementation of the device work: CPU tensors raise ``DmmError``.
"""
from __future__ import annotations

import io
import os
import pickle
from typing import Callable, List, Optional, Sequence

import torch

from . import _lib
from .proposals import ClipProposals, ProposalSlots, SimpleBoxList, filter_results, forward_mask_prop, prepare_slots


# ------------------------------------------------------------------------------------------------------------------
# device reductions
# ------------------------------------------------------------------------------------------------------------------
def mask_boxes(masks: torch.Tensor, thresh: float = 0.0):
    """masks [R,H,W] fp32 -> (boxes [R,4] fp32 xyxy of (mask > thresh), whole frame when empty; valid [R] int32)."""
    if not masks.is_cuda:
        raise _lib.DmmError("dmm_net_amd.video needs tensors on an MI355X device (no CPU fallback)")
    assert masks.dim() == 3 and masks.dtype == torch.float32, (masks.shape, masks.dtype)
    R, H, W = masks.shape
    if R and not (masks.stride(2) == 1 and masks.stride(1) == W and masks.stride(0) >= H * W):
        masks = masks.contiguous()
    boxes = torch.empty((R, 4), dtype=torch.float32, device=masks.device)
    valid = torch.empty((R,), dtype=torch.int32, device=masks.device)
    with _lib.device_guard(masks.device):
        rc = _lib.load().dmm_mask_boxes_f32(masks.data_ptr(), R, H, W, masks.stride(0) if R else H * W, float(thresh),
                                            boxes.data_ptr(), valid.data_ptr(),
                                            torch.cuda.current_stream(masks.device).cuda_stream)
    _lib.check(rc, "dmm_mask_boxes_f32")
    return boxes, valid


def ohw_mask2boxlist(ohw_mask: torch.Tensor):
    """utils.py:179-210: object masks [O,H,W] of one image -> (BoxList with 'mask' / 'scores', template_valid [O] long)."""
    O, H, W = ohw_mask.shape
    boxes, valid = mask_boxes(ohw_mask.float(), 0.0)
    bl = SimpleBoxList(boxes, (W, H), "xyxy")
    bl.add_field("mask", ohw_mask)
    bl.add_field("scores", ohw_mask.new_zeros((O,)) + 1)
    return bl, valid.long()


def merge_labels(outs: torch.Tensor, tplt_valid_batch: Optional[torch.Tensor] = None) -> torch.Tensor:
    """evaluator.py:134-139 for a batch: outs [B,O,HW] or [B,O,H,W] fp32, ``tplt_valid_batch`` [B,O] 0/1 (valid prefix)
    or [B] counts -> uint8 labels [B,HW] / [B,H,W] (0 = background, o+1 = object o)."""
    if not outs.is_cuda:
        raise _lib.DmmError("dmm_net_amd.video needs tensors on an MI355X device (no CPU fallback)")
    assert outs.dtype == torch.float32 and outs.dim() in (3, 4), (outs.shape, outs.dtype)
    shape = outs.shape
    hw = 1
    for d in shape[2:]:
        hw *= int(d)
    m = outs.reshape(shape[0], shape[1], hw)
    B, O, HW = m.shape
    if B * O * HW and m.stride(2) != 1:
        m = m.contiguous()
    ov = None
    if tplt_valid_batch is not None:
        ov = tplt_valid_batch if tplt_valid_batch.dim() == 1 else tplt_valid_batch.sum(1)
        ov = ov.to(device=m.device, dtype=torch.int32).contiguous()
    labels = torch.empty((B, HW), dtype=torch.uint8, device=m.device)
    with _lib.device_guard(m.device):
        rc = _lib.load().dmm_merge_labels_f32(m.data_ptr(), B, O, HW, m.stride(0), m.stride(1),
                                              None if ov is None else ov.data_ptr(), labels.data_ptr(),
                                              torch.cuda.current_stream(m.device).cuda_stream)
    _lib.check(rc, "dmm_merge_labels_f32")
    return labels.view(B, *shape[2:])


# ------------------------------------------------------------------------------------------------------------------
# output format: indexed PNG with the DAVIS palette
# ------------------------------------------------------------------------------------------------------------------
def davis_palette() -> List[int]:
    """768 ints: colour of label i has bit k of i spread to bit 7 - k//3 of channel k % 3 (PASCAL-VOC colour map)."""
    pal = []
    for i in range(256):
        r = g = b = 0
        c = i
        for j in range(8):
            r |= ((c >> 0) & 1) << (7 - j)
            g |= ((c >> 1) & 1) << (7 - j)
            b |= ((c >> 2) & 1) << (7 - j)
            c >>= 3
        pal += [r, g, b]
    return pal


def save_label_png(labels, fname: str) -> None:
    """plot_scores_map (eval_helper.py:22-38): label map [H,W] (or [1,H,W]) -> palette PNG; creates the directory."""
    from PIL import Image                                    # stdlib-free PNG writing is not worth owning
    import numpy as np
    d = os.path.dirname(fname)
    if d and not os.path.exists(d):
        os.makedirs(d)
    if isinstance(labels, torch.Tensor):
        labels = labels.detach().cpu().numpy()
    if labels.ndim == 3 and labels.shape[0] == 1:
        labels = labels[0]
    assert labels.ndim == 2, labels.shape
    img = Image.fromarray(labels.astype(np.uint8), "P")
    img.putpalette(davis_palette())
    img.save(fname)


# ------------------------------------------------------------------------------------------------------------------
# input format: offline proposal files
# ------------------------------------------------------------------------------------------------------------------
class _BoxListUnpickler(pickle.Unpickler):
    """Maps maskrcnn_benchmark's BoxList (attributes bbox / size / mode / extra_fields) onto SimpleBoxList."""

    def find_class(self, module, name):
        if name == "BoxList" and module.startswith("maskrcnn_benchmark"):
            return SimpleBoxList
        return super().find_class(module, name)


class _BoxListPickle:
    """``pickle_module`` for ``torch.load``."""
    __name__ = "dmm_net_amd.video._BoxListPickle"
    Unpickler = _BoxListUnpickler

    @staticmethod
    def load(f, **kw):
        return _BoxListUnpickler(f, **kw).load()

    @staticmethod
    def loads(b, **kw):
        return _BoxListUnpickler(io.BytesIO(b), **kw).load()


def load_offline_proposals(path: str, map_location="cpu"):
    """Read ``predictions.pth`` (list of BoxList), ``pred_DICT.pth`` ({vid: {frame: BoxList}}) or ``videos/<vid>.pth``
    ({frame: BoxList}) without maskrcnn_benchmark: every BoxList becomes a ``SimpleBoxList`` with its fields
    ('mask' [P,1,M,M] probabilities, 'scores' | 'objectness', ...)."""
    return torch.load(path, map_location=map_location, pickle_module=_BoxListPickle, weights_only=False)


# ------------------------------------------------------------------------------------------------------------------
# frame loop
# ------------------------------------------------------------------------------------------------------------------
import contextlib
_NULL = contextlib.nullcontext()


def _map_tensors(out, fn):
    """fn over every tensor in an encoder's output (dict / tuple / list nesting kept)."""
    if isinstance(out, torch.Tensor):
        return fn(out)
    if isinstance(out, dict):
        return {k: _map_tensors(v, fn) for k, v in out.items()}
    if isinstance(out, (tuple, list)):
        return type(out)(_map_tensors(v, fn) for v in out)
    return out


def _slice_batch(out, lo: int, hi: int):
    """Rows lo..hi of every tensor in an encoder's output."""
    return _map_tensors(out, lambda v: v[lo:hi])



class StepPlan:
    """Everything ONE frame step of the evaluator needs (evaluator.py:151-213 around dmm_model.py:48-86), at fixed
    device addresses, so that the step can be captured into a HIP graph once and replayed for every frame of every
    clip of the same shape with NO host input:

        step_select   row *step of the per-clip tables -> m_valid [B] (live templates, 0 = video skipped) / commit [B]
        proposal_boxes -> nms_slots -> paste_kept     raw proposals of frame *step -> K slots (proposals.prepare_slots);
                                                      ONLY the 1-bit planes of the kept proposals are written
        roialign4_mean                                slots' roi rows on the feature batch -> feat_p [B,K,D]
        match_solve_packed                            cosine, counts on 1-bit planes (proposals AND templates), solver -> Rb
        (x row_scale on Rb)                           only when the live templates are not a prefix (dmm_model.py:151-156)
        step_finish                                   mix with the selected proposals pasted on the fly -> full [B,O,H,W];
                                                      hist[b] = full[b] unless skipped (out_mask_last, :78-80); the
                                                      history's 1-bit planes; label map (evaluator.py:134-139)
        step_advance                                  *step += 1
    (more than 8 template slots or raw masks above 30 x 30: the soft planes are pasted and match_forward_packed +
    commit_masks + merge_labels run instead.)

    The clip's raw proposals live in ``clip`` ([T_cap,B,R,..]); the encoder's backbone features of two chunks of G frames
    in ``feats[l]`` [2*G*B, C, H_l, W_l] (chunk k in half k % 2), addressed through the roi rows' image index
    (``img_base[t]``).  ``hist`` is mask_last_occurence / mask_hist, ``tplt_feat`` the template features (fixed from
    frame 0, dmm_model.py:44)."""

    def __init__(self, B, O, H, W, R, Mm, K, G, T_cap, feat_like, device, cfg, nms_thresh, mask_thresh, padding,
                 tail: bool, fuse_epilogue: bool = True):
        from . import ops
        self.B, self.O, self.H, self.W, self.R, self.K, self.G, self.T_cap = B, O, H, W, R, K, G, T_cap
        self.cfg, self.tail = cfg, bool(tail)
        self.nms_thresh, self.mask_thresh, self.padding = float(nms_thresh), float(mask_thresh), int(padding)
        dev = torch.device(device)
        f32, i32 = dict(dtype=torch.float32, device=dev), dict(dtype=torch.int32, device=dev)
        self.clip = ClipProposals.empty(T_cap, B, R, Mm, dev)
        # fused epilogue (dmm_step_finish_f32): the selected proposals are pasted on the fly in the mix, the soft planes
        # are never written, commit / label merge / the history's 1-bit planes come out of the same pass
        self.fused = bool(fuse_epilogue) and O <= 8 and Mm + 2 * padding <= 32
        self.slots = ProposalSlots(B, K, H, W, R, dev, soft_planes=not self.fused)
        self.Pp = ops.padded_width(K, O)
        self.Rb = torch.zeros((B, O, self.Pp), **f32)
        self.packed_hist = torch.zeros((B, O, ops.pack_words(H * W)), dtype=torch.int64, device=dev)
        cl = [f.dim() == 4 and f.is_contiguous(memory_format=torch.channels_last) and not f.is_contiguous() for f in feat_like]
        self.feats = [torch.empty((2 * G * B,) + tuple(f.shape[1:]), dtype=f.dtype, device=dev,
                                  memory_format=torch.channels_last if c else torch.contiguous_format).zero_()
                      for f, c in zip(feat_like, cl)]
        C = int(feat_like[0].shape[1])
        self.D = 4 * C
        self.feat_p = torch.zeros((B * K, self.D), **f32)
        self.tplt_feat = torch.zeros((B, O, self.D), **f32)
        self.hist = torch.zeros((B, O, H, W), **f32)
        self.full = torch.zeros((B, O, H, W), **f32)
        self.out = (self.full, torch.zeros((B, O), **f32), torch.zeros((B, O), **f32), torch.zeros((B,), **i32))
        self.workspace = torch.empty((int(_lib.load().dmm_workspace_bytes_packed(B, K, O, self.D, H * W)),),
                                     dtype=torch.uint8, device=dev)
        self.step = torch.zeros((1,), **i32)
        self.tables = torch.zeros((T_cap, 2, B), **i32)          # [t, 0] = m_valid, [t, 1] = commit
        self.cur = torch.zeros((2, B), **i32)
        self.img_base = torch.zeros((T_cap,), **i32)
        self.n_tplt = torch.zeros((B,), **i32)
        self.row_scale_buf = torch.ones((B, O), **f32)           # fixed address: the captured step multiplies by it
        self.row_scale = None                                    # = row_scale_buf when the live templates are not a prefix
        self.labels = torch.zeros((B, H * W), dtype=torch.uint8, device=dev)
        self.graphs = {}                                         # row_scale? -> captured graph
        self.device = dev

    def key_fits(self, B, O, H, W, R, Mm, K, G, T, feat_like, tail, fuse_epilogue=True, cfg=None, nms_thresh=None,
                 mask_thresh=None, padding=None):
        """Can this plan (buffers + captured graphs) serve the clip?  Shapes AND every value the captured launches carry
        as an immediate: the solver configuration, the NMS / paste thresholds and the padding are baked into the graph,
        so a change of any of them (``loop.nms_thresh = ...``, another match layer) must rebuild and recapture --
        the BoxList path reads them live on every frame."""
        if cfg is not None and tuple(cfg) != tuple(self.cfg):
            return False
        if (nms_thresh is not None and float(nms_thresh) != self.nms_thresh) or \
                (mask_thresh is not None and float(mask_thresh) != self.mask_thresh) or \
                (padding is not None and int(padding) != self.padding):
            return False
        return ((self.B, self.O, self.H, self.W, self.R, self.clip.M, self.K, self.G, self.tail) ==
                (B, O, H, W, R, Mm, K, G, bool(tail)) and T <= self.T_cap and
                self.fused == (bool(fuse_epilogue) and O <= 8 and Mm + 2 * self.padding <= 32) and
                all(tuple(a.shape[1:]) == tuple(b.shape[1:]) and a.dtype == b.dtype and
                    a.is_contiguous() == b.is_contiguous() for a, b in zip(self.feats, feat_like)))

    def body(self):
        """The frame step as launches on the current stream (captured, or run directly)."""
        from . import ops
        from .roi_features import roialign4_mean_into
        L = _lib.load()
        s = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(L.dmm_step_select_i32(self.tables.data_ptr(), self.step.data_ptr(), 2 * self.B, self.cur.data_ptr(), s),
                   "dmm_step_select_i32")
        prepare_slots(self.clip, self.slots, self.nms_thresh, self.mask_thresh, self.padding, step=self.step,
                      img_base=self.img_base)
        roialign4_mean_into(self.slots.rois, self.feats, self.feat_p)
        score_weight, max_iter, proj_iter, lr, is_test = self.cfg
        if self.fused:
            ops.match_solve_packed(self.slots.packed, self.packed_hist, self.feat_p.view(self.B, self.K, self.D),
                                   self.tplt_feat, self.slots.scores, self.slots.count, self.cur[0], self.H * self.W,
                                   score_weight=score_weight, max_iter=max_iter, proj_iter=proj_iter, lr=lr,
                                   is_test=is_test, out=(self.Rb, self.out[1], self.out[2], self.out[3]),
                                   workspace=self.workspace)
            if self.row_scale is not None:
                self.Rb.mul_(self.row_scale[:, :, None])         # rows of slots i < O with valid[i] == 0: zero weights
            c = self.clip
            _lib.check(L.dmm_step_finish_f32(
                self.Rb.data_ptr(), self.Pp, c.prob.data_ptr(), c.boxes.data_ptr(), self.slots.keep.data_ptr(),
                self.slots.count.data_ptr(), self.B, c.R, c.M, self.K, self.O, self.H, self.W, self.padding,
                self.step.data_ptr(), self.cur[0].data_ptr(), self.cur[1].data_ptr() if self.tail else None,
                self.n_tplt.data_ptr(), self.full.data_ptr(), self.hist.data_ptr(), self.packed_hist.data_ptr(),
                self.labels.data_ptr() if self.tail else None, s), "dmm_step_finish_f32")
            _lib.check(L.dmm_step_advance(self.step.data_ptr(), s), "dmm_step_advance")
            return
        ops.match_forward_packed(self.slots.planes, self.slots.packed, self.hist, self.feat_p.view(self.B, self.K, self.D),
                                 self.tplt_feat, self.slots.scores, self.slots.count, self.cur[0],
                                 score_weight=score_weight, max_iter=max_iter, proj_iter=proj_iter, lr=lr, is_test=is_test,
                                 out=self.out, workspace=self.workspace)
        if self.row_scale is not None:
            self.full.mul_(self.row_scale[:, :, None, None])
        if self.tail:
            _lib.check(L.dmm_commit_masks_f32(self.full.data_ptr(), self.hist.data_ptr(), self.cur[1].data_ptr(), self.B,
                                              self.O * self.H * self.W, s), "dmm_commit_masks_f32")
            _lib.check(L.dmm_merge_labels_f32(self.full.data_ptr(), self.B, self.O, self.H * self.W,
                                              self.O * self.H * self.W, self.H * self.W, self.n_tplt.data_ptr(),
                                              self.labels.data_ptr(), s), "dmm_merge_labels_f32")
        _lib.check(L.dmm_step_advance(self.step.data_ptr(), s), "dmm_step_advance")

    def run_step(self, graph: bool):
        if not graph:
            return self.body()
        key = self.row_scale is not None
        g = self.graphs.get(key)
        if g is None:
            # warm-up on the real buffers would advance the clip: save / restore the two pieces of state it touches
            from .ops import _CAPTURE_LOCK
            keep_step, keep_hist, keep_packed = self.step.clone(), self.hist.clone(), self.packed_hist.clone()
            self.body()
            self.step.copy_(keep_step)
            self.hist.copy_(keep_hist)
            self.packed_hist.copy_(keep_packed)
            with _CAPTURE_LOCK, torch.cuda.device(self.device):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    self.body()
            self.graphs[key] = g
        g.replay()

class FrameLoop:
    """Frame loop of the evaluator for a batch of B videos (evaluator.py:63-213).

    ``encoder(img [B,3,H,W]) -> features`` (dict with 'backbone_feature', 'refine_input_feat'),
    ``dmm``: ``DMM_Model`` (is_test=1) with its ROI feature extractor,
    ``refine(features, prev_mask [B,O,HW], y_mask [B,O,HW], init_pred [B,O,H,W], mask_hist_new [B,O,H,W], valid [B,O],
    state) -> (outs [B,O,HW], mask_hist_new, state)``: the decoder step (:174-212); ``None`` = the matching layer's
    masks are the prediction.  Proposals come per video and frame as BoxLists with the raw 'mask' probabilities
    ([P,1,M,M], pasted + NMS-filtered here like model_encoder.py:115-134) or, with ``pasted=True``, already as
    image-size planes.

    Lifetime of encoder outputs: with ``encoder.GraphedEncoder`` the feature maps are views of the graph's static
    buffers and are overwritten by the next frame's replay.  Everything this loop keeps ACROSS frames is copied out
    (template vectors are fresh tensors from the ROI kernel; ``refine_input_feat`` of frame 0 is cloned); a ``refine``
    callable that keeps feature maps in its ``state`` must clone them likewise.

    Two reorderings against the reference's strictly sequential loop, neither changes a result for a feed-forward
    encoder (``lookahead = False``, ``encode_ahead = 1``, ``encode_overlap = False`` restore the reference's order):
    ``lookahead``: the proposals of frame t + 1 (paste, NMS, top-k -- they depend on nothing the loop computes) are
    prepared on a side stream after frame t's work has been enqueued, so the step's one host sync does not drain the main
    stream; ``encode_ahead``: that many frames of the clip go through the encoder as one time-major batch (the encoder
    has no temporal state; the templates, the mask history and the decoder carry it), and with ``encode_overlap`` the
    NEXT batch is encoded on its own stream while this one's steps run (outputs of a static-buffer encoder are cloned).
    An encoder whose output for an image depends on the rest of the batch (BatchNorm in train mode) needs
    ``encode_ahead = 1``.

    Two execution forms, bit-identical results (tests/test_gpu_video.py):
    ``slots = True`` (default; raw proposal masks, ``algo: 'relax'``): the FIXED-SLOT frame step of ``StepPlan`` -- the
    clip's raw proposals are uploaded once (``proposals.ClipProposals``; ``run`` also accepts one directly), the proposals
    that survive NMS + top-k live in ``max_proposals`` slots per video with the live count on the device, the whole step is
    nine launches and, with ``graph = True``, ONE HIP-graph replay per frame: no host sync, upload or allocation per
    frame.  ``fuse_epilogue`` (<= 8 template slots, raw masks <= 30 x 30): no soft proposal plane is ever written, the mix
    pastes its selected proposals on the fly and emits ``out_mask_last``, the label map and the template history's 1-bit
    planes in the same pass.  ``on_labels(b, t, labels)`` receives a view of a per-clip buffer, valid in stream order.
    ``slots = False``: the BoxList path -- paste every raw proposal, NMS, index the kept ones (one host sync per frame),
    ``DMM_Model.inference`` -- the reference's per-frame steps one to one (also taken for ``pasted = True`` / ``algo: 'hun'``).
    """

    def __init__(self, encoder: Callable, dmm, refine: Optional[Callable] = None, nms_thresh: float = 0.4,
                 max_proposals: int = 50, mask_thresh: float = 0.4, padding: int = 1, pasted: bool = False):
        self.encoder, self.dmm, self.refine = encoder, dmm, refine
        self.nms_thresh, self.max_proposals = float(nms_thresh), int(max_proposals)
        self.mask_thresh, self.padding, self.pasted = float(mask_thresh), int(padding), bool(pasted)
        self.lookahead = True                                    # proposals of frame t + 1 on a side stream (see run)
        self.encode_ahead = 0                                    # frames per encoder batch (see run); 1 = the reference's order;
                                                                 # 0 = by clip length: ceil(T / 3) within [4, 9]
        self.steps_priority = 0                                  # HIP stream priority of the steps' stream (-1: 0.412 -> 0.407 ms per step, within noise)
        self.encode_overlap = True                               # next chunk's encoder on its own stream (see run)
        # fixed-slot frame step (StepPlan): raw proposals of the whole clip on the device, two-phase paste, kept counts
        # stay on the device, and -- ``graph`` -- the whole step replayed from one HIP graph.  Off = the BoxList path
        # (paste every raw proposal, NMS, gather the kept ones, one host sync per frame).
        self.slots = True
        self.graph = True
        self.fuse_epilogue = True                                # dmm_step_finish_f32: mix with on-the-fly paste + commit +
                                                                 # label map in one pass, no soft proposal planes (StepPlan)
        self.encode_first = 0                                    # frames in the FIRST encoder batch (0 = encode_ahead): a short
                                                                 # first chunk shortens the pipeline fill before frame 0's step
        self.encoder_priority = 0                                # HIP stream priority of the encoder's side stream (-1 = high; measured harmful, see encoder.pick_parallel_stream)
        self._side = {}
        self._plan = None
        self._prefetched = None              # (key, encoder output) of the next clip's first chunk (run(next_frames=...))
        self.record_iters = False            # fixed-slot path: keep every step's solver iteration counts in
        self.last_iters = None               # ``last_iters`` [T,B] int32 (one small device copy per step; tests)

    def _frames_per_chunk(self, T: int, pipelined: bool = False) -> int:
        """``encode_ahead``, or by clip length when it is 0.  The ResNet gets more efficient with the batch (4 videos of
        255x448: 0.64 / 0.60 ms per frame step at 4 frames per chunk for clips of 12 / 24 frames, 0.50 at 8 of 24, 0.46 at
        9 of 36) while the first chunk is pipeline fill nothing overlaps -- about a third of the clip per chunk, between
        4 and 9 frames, was the best or within 2 % of it for clips of 12 to 48 frames.  ``pipelined`` (clips back to back,
        ``run(next_frames=...)``): the first chunk is issued under the previous clip, no fill to pay, so only the encoder's
        efficiency counts: up to 12 frames per chunk (12-frame clips: 0.50 / 0.48 / 0.42 ms per step at 4 / 6 / 12 frames
        per chunk; 36-frame clips: 0.39 / 0.38 / 0.39 / 0.43 at 9 / 12 / 18 / 36)."""
        if int(self.encode_ahead) > 0:
            return int(self.encode_ahead)
        if pipelined:
            return max(4, min(12, int(T)))
        return max(4, min(9, -(-int(T) // 3)))

    def _side_stream(self, dev, role="proposals", beside=()):
        prio = int(self.encoder_priority) if role == "encoder" else (int(self.steps_priority) if role == "steps" else 0)
        key = (role, dev.index if dev.index is not None else torch.cuda.current_device(), prio)
        if key not in self._side:
            if beside and not torch.cuda.is_current_stream_capturing():
                from .encoder import pick_parallel_stream
                self._side[key] = pick_parallel_stream(dev, list(beside), priority=prio)
            else:
                self._side[key] = torch.cuda.Stream(device=dev, priority=prio)
        return self._side[key]


    # ---- fixed-slot path ---------------------------------------------------------------------------------------------
    def _slots_ok(self, frames, proposals, O) -> bool:
        """The fixed-slot step covers the product's configuration: raw mask probabilities, the relaxed solver, shapes
        inside the kernels' envelopes.  Anything else takes the BoxList path."""
        if not (self.slots and frames.is_cuda and not self.pasted and self.dmm.match_algo == "relax"):
            return False
        if isinstance(proposals, ClipProposals):
            R, Mm = proposals.R, proposals.M
        else:
            ps = [p for v in proposals for p in v]
            if not ps or any("mask" not in p.fields() for p in ps):
                return False
            R, Mm = max(len(p) for p in ps), int(ps[0].get_field("mask").shape[-1])
        K = self.max_proposals if self.max_proposals > 0 else R
        return 0 < R <= 1024 and Mm + 2 * self.padding <= 64 and frames.shape[0] * K <= 65535 and O <= 32 and K <= 256

    def _run_slots(self, frames, first_masks, proposals, n_frames, targets, on_labels, next_frames=None):
        """``run`` on the fixed-slot step: zero host syncs per frame, and with ``graph`` one graph replay per frame.

        The steps run on the loop's OWN stream and the encoder's stream is PROBED against it: HIP maps streams onto a
        handful of hardware queues as they are created, and two streams that share a queue serialise -- it depended on
        what else the process had created whether the encoder really overlapped the steps (the same loop: 0.61 ms per
        step alone, 0.74 ms behind other workloads in one process = encoder + steps in series).
        The caller's stream is joined on both sides; ``on_labels`` callbacks run under the step stream."""
        dev = frames.device
        caller = torch.cuda.current_stream(dev)
        work = self._side_stream(dev, "steps")
        if self.encode_overlap:
            self._side_stream(dev, "encoder", beside=[work])     # probed: a queue of its own
            fast = getattr(self.encoder, "encoder", self.encoder)
            if hasattr(fast, "avoid_streams") and work not in fast.avoid_streams:
                fast.avoid_streams.append(work)                  # ... and the encoder's own side stream a third one
        work.wait_stream(caller)
        with torch.cuda.stream(work):
            history = self._run_slots_on_stream(frames, first_masks, proposals, n_frames, targets, on_labels, next_frames)
        caller.wait_stream(work)
        for h in history[:1]:
            h.record_stream(caller)                              # (the entries are views of one buffer)
        return history

    def _first_chunk(self, T: int, pipelined: bool = False) -> int:
        G = max(1, min(self._frames_per_chunk(T, pipelined), T))
        return max(1, min(int(self.encode_first) or G, G, T))

    def _run_slots_on_stream(self, frames, first_masks, proposals, n_frames, targets, on_labels, next_frames=None):
        B, T, C, H, W = frames.shape
        O = first_masks.shape[1]
        dev = frames.device
        main = torch.cuda.current_stream(dev)
        enc_side = self._side_stream(dev, "encoder") if self.encode_overlap else None
        # a prefetch is identified by the frames TENSOR it was issued for (held in the record, so the allocator cannot hand
        # its address to another clip) and that tensor's version counter; anything else is dropped
        pre, self._prefetched = self._prefetched, None
        if pre is not None and not (pre[0][0] is frames and pre[0][1] == frames._version):
            pre = None                                           # a prefetch for other frames: ignored
        pipelined = enc_side is not None and (pre is not None or next_frames is not None)
        G = max(1, min(self._frames_per_chunk(T, pipelined), T))
        need_features = self.refine is not None                  # the decoder reads refine_input_feat of every frame
        static = getattr(self.encoder, "static_outputs", False)
        plan = None

        # chunks of the clip: (first frame, frames); the first one may be shorter (encode_first)
        g0 = self._first_chunk(T, pipelined)
        chunks = [(0, g0)] + [(t0, min(G, T - t0)) for t0 in range(g0, T, G)]
        chunk_of = [k for k, (t0, g) in enumerate(chunks) for _ in range(g)]

        def encode(k, clip=None):
            """chunk k = frames [t0, t0 + g): encoder batch, time-major; its backbone features go to half k % 2 of the
            plan's feature batch ON THE STREAM THAT PRODUCED THEM (a static-output encoder overwrites them on its next
            call), after the steps that still read that half (chunk k - 2) have been passed on the main stream.
            ``clip``: the NEXT clip's frames -- its first chunk, prefetched."""
            t0, g = chunks[k] if clip is None else (0, self._first_chunk(clip.shape[1], True))
            src = frames if clip is None else clip
            ctx = torch.cuda.stream(enc_side) if enc_side is not None else _NULL
            fence = main.record_event() if enc_side is not None else None
            with ctx:
                xs = src[:, t0] if g == 1 else src[:, t0:t0 + g].transpose(0, 1).reshape(g * src.shape[0], C, H, W)
                out = self.encoder(xs)
                if need_features and static:
                    out = _map_tensors(out, lambda v: v.clone())
                if enc_side is not None:
                    _map_tensors(out, lambda v: (v.record_stream(main), v)[1])
                return out, fence

        def land(k, out, fence):
            """backbone features of chunk k -> the plan's feature batch (needs the plan, i.e. the first chunk's shapes)."""
            ctx = torch.cuda.stream(enc_side) if enc_side is not None else _NULL
            with ctx:
                if fence is not None:
                    enc_side.wait_event(fence)
                lo = (k % 2) * G * B
                for dst, src in zip(plan.feats, out["backbone_feature"]):
                    dst[lo:lo + src.shape[0]].copy_(src)
                return enc_side.record_event() if enc_side is not None else None

        if enc_side is not None:
            enc_side.wait_stream(main)
        if pre is not None and pre[0][2:] == (g0, enc_side):
            out0 = pre[1]                                        # issued under the previous clip's last steps
        else:
            out0, _ = encode(0)
        # ---- the plan (buffers + captured graph) for this shape --------------------------------------------------------
        if isinstance(proposals, ClipProposals):
            R, Mm = proposals.R, proposals.M
        else:
            R = max(len(p) for v in proposals for p in v)
            Mm = int(proposals[0][0].get_field("mask").shape[-1])
        K = self.max_proposals if self.max_proposals > 0 else R
        cfg = self.dmm.match_layer
        cfg = (float(cfg.cfgs["score_weight"]), int(cfg.max_iter), int(cfg.proj_iter), float(cfg.relax_lr),
               int(bool(cfg.is_test)))
        tail = self.refine is None
        plan = self._plan
        if plan is None or plan.device != dev or not plan.key_fits(
                B, O, H, W, R, Mm, K, G, T, out0["backbone_feature"], tail, self.fuse_epilogue, cfg=cfg,
                nms_thresh=self.nms_thresh, mask_thresh=self.mask_thresh, padding=self.padding):
            plan = self._plan = StepPlan(B, O, H, W, R, Mm, K, G, max(T, 32), out0["backbone_feature"], dev, cfg,
                                         self.nms_thresh, self.mask_thresh, self.padding, tail, self.fuse_epilogue)
        # (the plan's buffers may have just been created -- zero-filled -- on THIS stream: chunk 0 lands behind that, not
        # behind the fence taken before the plan existed; a high-priority encoder stream overtook the fill otherwise)
        ready = land(0, out0, main.record_event() if enc_side is not None else None)
        chunk = out0
        # ---- the clip's raw proposals and per-frame tables: one upload, before the loop ----------------------------
        if isinstance(proposals, ClipProposals):
            for dst, src in ((plan.clip.prob, proposals.prob), (plan.clip.boxes, proposals.boxes),
                             (plan.clip.scores, proposals.scores), (plan.clip.counts, proposals.counts)):
                dst[:T].copy_(src[:T], non_blocking=True)
        else:
            ClipProposals.from_boxlists(proposals, T, H, W, dev, out=plan.clip)
        plan.img_base[:T].copy_(_lib.small_to_device([(chunk_of[t] % 2) * G * B + (t - chunks[chunk_of[t]][0]) * B
                                                      for t in range(T)], torch.int32, dev))
        plan.step.zero_()
        y0 = first_masks.float().view(B, O, H * W)
        plan.hist.copy_(y0.view(B, O, H, W))
        if plan.fused:
            from . import ops
            plan.packed_hist.copy_(ops.pack_masks(plan.hist))
        history, state, prev_mask, tplt_valid = [], None, y0, None
        hist_all = torch.empty((T, B, O, H * W), dtype=torch.float32, device=dev)
        lab_all = torch.empty((T, B, H, W), dtype=torch.uint8, device=dev) if on_labels is not None else None
        it_all = torch.full((T, B), -1, dtype=torch.int32, device=dev) if self.record_iters else None
        next_chunk = None
        for t in range(T):
            extra = [n <= t for n in n_frames]
            if t == 0:
                y_mask = y0
            elif targets is not None:
                y_mask = targets[:, t].float().view(B, O, H * W)
            else:
                y_mask = None                                            # zeros (only the decoder reads it)
            kc = chunk_of[t]
            j = t - chunks[kc][0]
            if j == 0:
                if t > 0:
                    chunk, ready = next_chunk
                if ready is not None:
                    main.wait_event(ready)
                if kc + 1 < len(chunks):
                    o, f = encode(kc + 1)
                    next_chunk = (o, land(kc + 1, o, f))
                elif (next_frames is not None and enc_side is not None and next_frames.is_cuda
                      and tuple(next_frames.shape[2:]) == (C, H, W)):
                    # the encoder's stream has nothing left to do for this clip: the next clip's first chunk
                    next_frames.record_stream(enc_side)          # read there after this run has returned
                    self._prefetched = ((next_frames, next_frames._version,
                                         self._first_chunk(next_frames.shape[1], True), enc_side),
                                        encode(0, clip=next_frames)[0])
            if t == 0:                                                   # forward_timestep_init, :215-225
                boxes, valid = mask_boxes(y0.view(B * O, H, W), 0.0)
                tplt_valid = valid.view(B, O).long()
                n_tplt, row_scale = self.dmm._valid_layout(tplt_valid)  # the clip's ONE host sync
                ids = torch.arange(B, device=dev, dtype=torch.float32).repeat_interleave(O)
                rois = torch.cat([ids[:, None], boxes], 1)
                from .roi_features import roialign4_mean_into
                roialign4_mean_into(rois, plan.feats, plan.tplt_feat.view(B * O, -1))
                if row_scale is not None:
                    plan.tplt_feat.mul_(row_scale[:, :, None])
                if row_scale is not None:
                    plan.row_scale_buf.copy_(row_scale)
                plan.row_scale = plan.row_scale_buf if row_scale is not None else None
                plan.n_tplt.copy_(_lib.small_to_device(n_tplt, torch.int32, dev))
                rows = [[[0 if (n <= tt or n_tplt[b] == 0) else n_tplt[b] for b, n in enumerate(n_frames)],
                         [0 if (tt == 0 or n <= tt or n_tplt[b] == 0) else 1 for b, n in enumerate(n_frames)]]
                        for tt in range(T)]
                plan.tables[:T].copy_(_lib.small_to_device(rows, torch.int32, dev))
            plan.run_step(self.graph)
            if it_all is not None:
                it_all[t].copy_(plan.out[3])
            if self.refine is not None:
                features = chunk if chunks[kc][1] == 1 else _slice_batch(chunk, j * B, (j + 1) * B)
                live = plan.cur[0] > 0
                out_last = torch.where(live[:, None, None, None], plan.full, plan.hist)
                zeros = first_masks.new_zeros((B, O, H * W), dtype=torch.float32) if y_mask is None else y_mask
                outs, hist_new, state = self.refine(features, prev_mask, zeros, plan.full, out_last, tplt_valid, state)
                if t > 0:
                    plan.hist.copy_(hist_new.view(B, O, H, W))
                    if plan.fused:
                        from . import ops
                        plan.packed_hist.copy_(ops.pack_masks(plan.hist))
                outs = outs.reshape(B, O, H * W)
            else:
                outs = plan.full.view(B, O, H * W)
            if t == 0:                                                   # frame 0 only warms the decoder state, :119-128
                outs = y0
            hist_all[t].copy_(outs)
            prev_mask = hist_all[t]
            if on_labels is not None:
                if t == 0 or self.refine is not None:
                    lab_all[t].copy_(merge_labels(hist_all[t].view(B, O, H, W), tplt_valid))
                else:
                    lab_all[t].copy_(plan.labels.view(B, H, W))
                for b in range(B):
                    if not extra[b]:
                        on_labels(b, t, lab_all[t, b])
            history.append(hist_all[t])
        self.last_iters = it_all
        return history

    # model_encoder.py:115-134
    def prepare_proposals(self, raw: Sequence, im_h: int, im_w: int, device):
        props = []
        for p in raw:
            q = p.resize((im_w, im_h)) if tuple(p.size) != (im_w, im_h) else p
            props.append(q.to(device))
        if not self.pasted:
            # the paste kernel emits the 1-bit planes along with the soft ones: DMM_Model.inference counts on those
            props = forward_mask_prop([p.get_field("mask") for p in props], props, self.mask_thresh, self.padding,
                                      want_packed=True)
        score_field = "scores" if "scores" in props[0].fields() else "objectness"
        return filter_results(list(props), self.nms_thresh, self.max_proposals, score_field)

    @torch.no_grad()
    def run(self, frames: torch.Tensor, first_masks: torch.Tensor, proposals: Sequence[Sequence],
            n_frames: Optional[Sequence[int]] = None, targets: Optional[torch.Tensor] = None,
            on_labels: Optional[Callable] = None, next_frames: Optional[torch.Tensor] = None):
        """frames [B,T,3,H,W]; first_masks [B,O,H*W] (frame-0 annotation); proposals[b][t] (the last entry is reused
        for missing frames, evaluator.py:101-106); n_frames[b] = real length of video b (later frames are 'extra',
        :86); targets [B,T,O,HW] optional per-frame annotation (zeros otherwise).  Calls ``on_labels(b, t, uint8 [H,W])``
        for every real frame and returns the list over t of ``outs`` [B,O,HW].

        ``next_frames``: the frames of the clip the NEXT ``run`` call will be given (the evaluator walks a list of clips,
        evaluator.py:63-70).  Its first encoder chunk is then issued on the encoder's stream under this clip's last
        steps, when that stream has nothing left to do -- the next clip starts without the pipeline fill (one chunk of the
        encoder that nothing overlaps: a quarter of a 12-frame clip's time).  Results are unchanged; a later ``run`` on
        other frames simply ignores the prefetch."""
        B, T, C, H, W = frames.shape
        O = first_masks.shape[1]
        dev = frames.device
        n_frames = list(n_frames) if n_frames is not None else [T] * B
        if self._slots_ok(frames, proposals, O):
            return self._run_slots(frames, first_masks, proposals, n_frames, targets, on_labels, next_frames)
        history, state, mask_hist = [], None, None
        tplt_dict = tplt_valid = prev_mask = n_tplt = row_scale = None
        # Proposal look-ahead.  A frame's proposals (paste, tight boxes, NMS, top-k) depend on nothing the loop computes,
        # and their NMS ends in the step's only host sync (the kept counts).  On the main stream that sync drained
        # the whole step and the GPU idled while the host enqueued the next one; here the proposals of frame t + 1 are
        # prepared on a side stream AFTER frame t's work has been enqueued, so the sync waits for a few short kernels
        # only and the main stream never runs dry.
        main = torch.cuda.current_stream(dev) if dev.type == "cuda" else None
        side = self._side_stream(dev) if main is not None and self.lookahead else None

        def prepare(tt):
            raw = [proposals[b][tt] if len(proposals[b]) > tt else proposals[b][-1] for b in range(B)]
            if side is None:
                return self.prepare_proposals(raw, H, W, dev), None
            with torch.cuda.stream(side):
                out = self.prepare_proposals(raw, H, W, dev)
                for bl in out:                                   # allocated on `side`, consumed on `main`
                    for v in [bl.bbox] + [bl.get_field(f) for f in bl.fields()]:
                        if isinstance(v, torch.Tensor) and v.is_cuda:
                            v.record_stream(main)
                return out, side.record_event()

        G = max(1, min(self._frames_per_chunk(T), T))
        enc_side = self._side_stream(dev, "encoder") if main is not None and self.encode_overlap else None

        def encode(t0):
            g = min(G, T - t0)

            def go():
                xs = frames[:, t0] if g == 1 else frames[:, t0:t0 + g].transpose(0, 1).reshape(g * B, C, H, W)
                return self.encoder(xs)
            if enc_side is None:
                out = go()
                if getattr(self.encoder, "static_outputs", False):
                    # views of a graph's static buffers: the NEXT chunk is encoded (same graph, same buffers) as soon as
                    # this one has been taken, i.e. before its steps are enqueued -- they must not alias the buffers
                    out = _map_tensors(out, lambda v: v.clone())
                return out, None
            with torch.cuda.stream(enc_side):
                out = go()
                if getattr(self.encoder, "static_outputs", False):
                    # views of a graph's static buffers: the next replay (issued while this chunk is still in use)
                    # would overwrite them
                    out = _map_tensors(out, lambda v: v.clone())
                _map_tensors(out, lambda v: (v.record_stream(main), v)[1])
                return out, enc_side.record_event()

        if side is not None:
            side.wait_stream(main)                               # inputs the caller produced on the main stream
        if enc_side is not None:
            enc_side.wait_stream(main)
        ahead = prepare(0) if T > 0 else None
        next_chunk = encode(0) if T > 0 else None
        for t in range(T):
            extra = [n <= t for n in n_frames]
            x = frames[:, t]
            if t == 0:
                y_mask = first_masks.float().view(B, O, H * W)
            elif targets is not None:
                y_mask = targets[:, t].float().view(B, O, H * W)
            else:
                y_mask = first_masks.new_zeros((B, O, H * W), dtype=torch.float32)
            props, ready = ahead
            if ready is not None:
                main.wait_event(ready)
            # The encoder has no temporal state (the templates and the decoder carry it), so `encode_ahead` frames of
            # the clip go through it as ONE batch, time-major: G x B images per launch sequence instead of B -- the
            # ResNet at 4 x 255 x 448 is launch bound (~90 kernels of 5-20 us), at 16-48 images it is not.
            # `encode_overlap`: the chunk AFTER this one is encoded on its own stream while this chunk's steps -- a
            # dependent chain of small kernels that leaves most of the chip idle -- run on the main stream.
            if t % G == 0:
                chunk, enc_ready = next_chunk
                if enc_ready is not None:
                    main.wait_event(enc_ready)
                if t + G < T:
                    next_chunk = encode(t + G)
            j = t % G
            features = chunk if min(G, T - (t - j)) == 1 else _slice_batch(chunk, j * B, (j + 1) * B)
            if t == 0:                                                   # forward_timestep_init, :215-225
                tpl, valid = [], []
                for b in range(B):
                    bl, v = ohw_mask2boxlist(y_mask[b].view(O, H, W))
                    tpl.append(bl)
                    valid.append(v)
                tplt_valid = torch.stack(valid, 0)
                # once per clip (the templates are fixed): live counts + the non-prefix row scale of the reference's
                # OF_matrix (an object that is empty in frame 0 leaves a hole in the valid slots)
                n_tplt, row_scale = self.dmm._valid_layout(tplt_valid)
                tplt_dict = self.dmm.fill_template_dict(None, tpl, features, y_mask, tplt_valid)
                if getattr(self.encoder, "static_outputs", False):
                    # a GraphedEncoder returns views of its graph's static buffers; every later replay overwrites
                    # them, so what the template dictionary keeps across frames must own its memory
                    for b in tplt_dict:
                        tplt_dict[b]["refine_input_feat"] = [tuple(f.clone() for f in tup)
                                                             for tup in tplt_dict[b]["refine_input_feat"]]
                prev_mask = y_mask
            infos = {"extra_frame": extra, "valid": tplt_valid, "shape": [[H, W]] * B, "n_tplt": n_tplt,
                     "row_scale": row_scale,
                     "alias_ok": self.refine is None}             # nothing here edits init_pred / hist_new in place; a
                                                                  # decoder may (evaluator.py:205): it gets its own tensor
            hist_in = prev_mask.view(B, O, H, W) if mask_hist is None else mask_hist       # :168-169
            init_pred, tplt_dict, _, hist_new = self.dmm.inference(infos, props, features["backbone_feature"], hist_in,
                                                                   tplt_dict)
            if self.refine is not None:
                outs, hist_new, state = self.refine(features, prev_mask, y_mask, init_pred, hist_new, tplt_valid, state)
            else:
                outs = init_pred.reshape(B, O, H * W)
            if t == 0:                                                   # frame 0 only warms the decoder state, :119-128
                outs = y_mask
            else:
                mask_hist = hist_new
            prev_mask = outs.view(B, O, H * W)
            if on_labels is not None:
                labels = merge_labels(outs.view(B, O, H, W), tplt_valid)
                for b in range(B):
                    if not extra[b]:
                        on_labels(b, t, labels[b])
            history.append(outs)
            if t + 1 < T:
                ahead = prepare(t + 1)                           # host sync inside: frame t's work is already queued
        return history
