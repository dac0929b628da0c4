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
from .proposals import SimpleBoxList, filter_results, forward_mask_prop


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
    """

    def __init__(self, encoder: Callable, dmm, refine: Optional[Callable] = None, nms_thresh: float = 0.4,
                 max_proposals: int = 50, mask_thresh: float = 0.4, padding: int = 1, pasted: bool = False):
        self.encoder, self.dmm, self.refine = encoder, dmm, refine
        self.nms_thresh, self.max_proposals = float(nms_thresh), int(max_proposals)
        self.mask_thresh, self.padding, self.pasted = float(mask_thresh), int(padding), bool(pasted)
        self.lookahead = True                                    # proposals of frame t + 1 on a side stream (see run)
        self.encode_ahead = 4                                    # frames per encoder batch (see run); 1 = the reference's order
        self.encode_overlap = True                               # next chunk's encoder on its own stream (see run)
        self._side = {}

    def _side_stream(self, dev, role="proposals"):
        key = (role, dev.index if dev.index is not None else torch.cuda.current_device())
        if key not in self._side:
            self._side[key] = torch.cuda.Stream(device=dev)
        return self._side[key]

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
            on_labels: Optional[Callable] = None):
        """frames [B,T,3,H,W]; first_masks [B,O,H*W] (frame-0 annotation); proposals[b][t] (the last entry is reused
        for missing frames, evaluator.py:101-106); n_frames[b] = real length of video b (later frames are 'extra',
        :86); targets [B,T,O,HW] optional per-frame annotation (zeros otherwise).  Calls ``on_labels(b, t, uint8 [H,W])``
        for every real frame and returns the list over t of ``outs`` [B,O,HW]."""
        B, T, C, H, W = frames.shape
        O = first_masks.shape[1]
        dev = frames.device
        n_frames = list(n_frames) if n_frames is not None else [T] * B
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

        G = max(1, min(int(self.encode_ahead), T))
        enc_side = self._side_stream(dev, "encoder") if main is not None and self.encode_overlap else None

        def encode(t0):
            g = min(G, T - t0)

            def go():
                xs = frames[:, t0] if g == 1 else frames[:, t0:t0 + g].transpose(0, 1).reshape(g * B, C, H, W)
                return self.encoder(xs)
            if enc_side is None:
                return go(), None
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
                     "row_scale": row_scale}
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
