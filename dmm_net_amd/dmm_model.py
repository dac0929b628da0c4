"""``DMM_Model`` -- counterpart of the reference's per-video driver ``dmm/modules/dmm_model.py``.

The reference loops over the videos of a batch in Python (``for bid in range(B)``, dmm_model.py:62 / :115) and
calls ``MatchModel`` once per video with the first-O rows selected by 0/1 ``OF_matrix`` matmuls
(:144-158), scattering the O result rows back into F = maxseqlen slots (:78-80 / :133-135).

Here all videos of the step go through ONE ragged batched launch sequence of the HIP layer
(``n_valid`` = proposals per video, ``m_valid`` = live templates per video, valid templates are a prefix as in
the reference, :124); the select / scatter matmuls become the kernels' zero-filled rows.  Semantics kept:

  * O == 0 (or ``extra_frame`` at inference): output zeros, ``out_mask_last`` = the incoming
    ``mask_last_occurence[b]`` unchanged, loss 0 (dmm_model.py:66-69, :118-122);
  * otherwise ``output_mask[b]`` and ``out_mask_last[b]`` are BOTH the scattered ``full_outmask``
    (MatchModel returns it twice, match_model.py:47);
  * ``forward`` returns ``(output_mask, tplt_dict, match_loss list, out_mask_last)``; ``inference`` returns
    ``match_loss = []`` (:85).

Proposals are duck-typed like maskrcnn_benchmark ``BoxList``: ``len(p)``, ``p.get_field('mask')`` ([P,1,H,W]),
``p.fields()``, ``p.get_field('objectness' | 'scores')``.  ROI feature extraction (reference
``feature_extractor.py``, maskrcnn_benchmark Pooler) is injected as ``feature_extractor`` -- any callable
``(backbone_feature, proposals) -> [sum P, D]``.
"""
from __future__ import annotations

import weakref
from typing import Callable, List, Optional

import torch
import torch.nn as nn

from . import _lib
from .autograd import match_layer_batched, ragged_pad
from .match_model import MatchModel


def CHECK4D(t):
    assert len(t.shape) == 4, "get {} {}".format(t.shape, len(t.shape))
    return t.shape


_TF_MEMO = weakref.WeakKeyDictionary()              # DMM_Model instance -> (key, weak refs, stacked template features of the clip)
_TABLE_MEMO = weakref.WeakKeyDictionary()           # DMM_Model instance -> memo of its last small-table upload (_lib.small_to_device_many)
_VALID_CACHE = weakref.WeakKeyDictionary()          # DMM_Model instance -> (weakref of the clip's valid tensor, its version, layout)


class DMM_Model(nn.Module):
    def __init__(self, cfgs, is_test=0, feature_extractor: Optional[Callable] = None):
        super().__init__()
        self.match_layer = MatchModel(cfgs, is_test)
        self.feature_extractor = feature_extractor
        self.match_algo = cfgs["matching"]["algo"]
        self.cfgs = cfgs
        self.is_test = is_test

    # ---- dmm_model.py:22-46 ------------------------------------------------------------------------
    def fill_template_dict(self, args, proposals, features, y_mask, tplt_valid_batch):
        backbone_feature = features["backbone_feature"]
        refine_input_feat = features["refine_input_feat"]
        boxes_per_image = [len(box) for box in proposals]
        result_alllevel = self.feature_extractor(backbone_feature, proposals)
        feats = result_alllevel.split(boxes_per_image, dim=0)
        tplt_dict = {}
        for b, feat in enumerate(feats):
            tplt_dict[b] = {"feat": [feat],
                            "refine_input_feat": [tuple([f[b] for f in refine_input_feat])]}
        return tplt_dict

    # ---- shared batched core -----------------------------------------------------------------------
    def _match_batch(self, prop_feat: List[torch.Tensor], prop_m: List[torch.Tensor], prop_score: List[torch.Tensor],
                     tplt_feat: List[torch.Tensor], mask_last_occurence, n_tplt: List[int], targets, skip: List[bool],
                     row_scale=None, packed: Optional[List[torch.Tensor]] = None):
        """prop_feat[b] [P_b,D], prop_m[b] [P_b,H,W], tplt_feat[b] [F,D]; returns (full [B,F,H,W], loss [B])."""
        B, F, H, W = CHECK4D(mask_last_occurence)
        dev = mask_last_occurence.device
        D = prop_feat[0].shape[1]
        Pmax = max(int(p.shape[0]) for p in prop_m)
        for b in range(B):
            P = prop_m[b].shape[0]
            assert prop_m[b].shape[-2:] == mask_last_occurence[b].shape[-2:], \
                "get {} {}".format(prop_m[b].shape[-2:], mask_last_occurence[b].shape[-2:])
            assert prop_feat[b].shape[0] == P, "get {} {}".format(P, prop_feat[b].shape[0])
        m_counts = [0 if skip[b] else n_tplt[b] for b in range(B)]
        pm = list(prop_m)
        if dev.type == "cuda":
            # the per-video blocks are stacked by one launch each (a zero fill + one copy per video before), with a
            # gradient path for the feature rows (the scores carry none: the layer returns no score gradient); the two
            # count vectors and the three pointer tables of the call go up in ONE copy
            from . import ops
            # the feature rows of a batch of equal proposal counts are usually the row ranges of ONE tensor (the ROI
            # extractor's [sum P, D] output, split per video, dmm_model.py:53-54 / :106): then the batch IS that tensor,
            # viewed -- no launch, and the gradient goes straight to it
            pf = self._rows_of_one_tensor(prop_feat, B, Pmax, D)
            if pf is None:
                pf_blocks, pf_addr = ops.ragged_blocks([f.float() for f in prop_feat])
            else:
                pf_addr = []
            dense_rows = all(int(p.shape[0]) == Pmax for p in prop_m)
            if dense_rows:                            # every video has Pmax proposals: the score batch is a plain stack
                sc_addr = []
            else:
                sc_blocks, sc_addr = ops.ragged_blocks([s_.detach().float().reshape(-1, 1) for s_ in prop_score])
            i32, i64 = torch.int32, torch.int64
            specs = [([int(p.shape[0]) for p in prop_m], i32), (m_counts, i32), (pf_addr, i64), (sc_addr, i64)]
            direct = not (any(t.requires_grad for t in pm) or len({t.dtype for t in pm}) > 1 or pm[0].dtype not in ops._DT)
            if direct:                                # (else match_layer_batched stacks a copy, autograd.py)
                pm = ops.FramePlanes(pm, table=None)
                specs.append((pm.addresses(), i64))
            # (the tables of a clip's frames repeat: same counts, and the allocator hands out the same addresses -- then the
            # previous call's device tables are reused as they are)
            up = _lib.small_to_device_many(specs, dev, memo=_TABLE_MEMO.setdefault(self, {}))
            n_valid, m_valid = up[0], up[1]
            if direct:
                pm.table = up[4]
            if pf is None:
                pf = ragged_pad(pf_blocks, Pmax, n_valid, up[2])
            if dense_rows:
                sc = torch.stack([s_.detach().reshape(-1) for s_ in prop_score], 0).float()
            else:
                sc = ops.ragged_pad(sc_blocks, Pmax, n_valid, up[3]).view(B, Pmax)
        else:
            n_valid = torch.tensor([int(p.shape[0]) for p in prop_m], dtype=torch.int32, device=dev)
            m_valid = torch.tensor(m_counts, dtype=torch.int32, device=dev)
            pf = prop_feat[0].new_zeros((B, Pmax, D))
            sc = mask_last_occurence.new_zeros((B, Pmax))
            for b in range(B):
                P = prop_m[b].shape[0]
                pf[b, :P] = prop_feat[b]
                sc[b, :P] = prop_score[b]
        # the mask planes stay where they are: one tensor per video, handed to the kernels as a pointer table (the
        # round-1 driver copied them into a [B, Pmax, H, W] batch: 2 x 13 MB per video in front of a 15.6 MB cost pass)
        # the templates of a clip are fixed from frame 0 (dmm_model.py:44): when they carry no gradient, their stacked batch is
        # kept for as long as the same tensors (same objects, same versions) come back -- one launch less per frame step
        if any(t.requires_grad for t in tplt_feat):
            tf = torch.stack([t.view(F, -1) for t in tplt_feat], 0)
        else:
            key = tuple((id(t), t._version, t.data_ptr()) for t in tplt_feat)
            memo = _TF_MEMO.get(self)
            if memo is not None and memo[0] == key and all(r() is t for r, t in zip(memo[1], tplt_feat)):
                tf = memo[2]
            else:
                tf = torch.stack([t.view(F, -1) for t in tplt_feat], 0)
                try:
                    _TF_MEMO[self] = (key, [weakref.ref(t) for t in tplt_feat], tf)
                except TypeError:
                    _TF_MEMO.pop(self, None)
        if row_scale is not None:
            # valid templates that are NOT a prefix: the reference's OF_matrix = diag(valid)[:O] (dmm_model.py:151-156)
            # zeroes the feature rows -- and, transposed, the scattered output rows (:78-80) -- of slots i < O with
            # valid[i] == 0
            tf = tf * row_scale[:, :, None]
        counts = None
        if packed is not None and targets is None:
            # the proposals come with their 1-bit (mask > 0.5) planes (emitted by the paste kernel): the cost pass
            # counts on those -- 1/32 of the proposal bytes, identical integer tables -- and the templates are packed
            # on the fly (one read of the F planes, what the float kernel would have read anyway)
            from . import ops
            pk = ops.ragged_pad(list(packed), Pmax, n_valid)
            counts = ops.iou_counts_packed(pk, ops.pack_masks(mask_last_occurence.float()), H * W, n_valid, m_valid)
        cfg = self.match_layer
        if counts is None and all(int(p.shape[0]) == Pmax for p in prop_m) and all(c == F for c in m_counts):
            # every video has all of its proposals and templates (the usual training batch): a DENSE batch -- the layer then
            # takes the exact-row kernels and, for a handful of videos, the one-launch similarity + counts kernel
            n_valid = m_valid = None
        full, ms, ds, loss, _ = match_layer_batched(
            pf, pm, tf, mask_last_occurence, sc, targets, n_valid, m_valid, score_weight=cfg.cfgs["score_weight"],
            max_iter=cfg.max_iter, proj_iter=cfg.proj_iter, lr=cfg.relax_lr, is_test=int(bool(cfg.is_test)),
            counts=counts)
        if row_scale is not None:
            full = full * row_scale[:, :, None, None]
        return full, loss

    @staticmethod
    def _rows_of_one_tensor(blocks, B, P, D):
        """[B, P, D] view of the tensor the per-video blocks are consecutive row ranges of, or None: every block a
        contiguous fp32 [P, D] view of the SAME base (what ``.split`` of a contiguous [B * P, D] tensor gives), in order,
        covering it."""
        base = getattr(blocks[0], "_base", None)
        if base is None or base.dim() != 2 or base.dtype != torch.float32 or tuple(base.shape) != (B * P, D) \
                or not base.is_contiguous():
            return None
        p0, step = base.data_ptr(), P * D * 4
        for b, f in enumerate(blocks):
            if f._base is not base or tuple(f.shape) != (P, D) or not f.is_contiguous() or f.data_ptr() != p0 + b * step:
                return None
            if f.requires_grad != base.requires_grad:              # views made under no_grad of a base that requires grad: the
                return None                                        # gradient must not reach the base through this shortcut
        return base.view(B, P, D)

    def _match_single(self, prop_feat, prop_m, prop_score, tplt_feat, mask_last_occurence, O, targets):
        """ONE video (the evaluator's batch: scripts/eval/*.sh run ``-batch_size=1``) with its templates a prefix and none
        skipped: the reference's own call, ``self.match_layer`` on the first O rows (dmm_model.py:75-77 / :130-132) --
        three launches in the evaluator, one library call each way in the trainer; the ragged batch machinery (count
        vectors, pointer tables, batching launches) is for B > 1.  Returns (full [1,F,H,W], [loss])."""
        _, F, H, W = CHECK4D(mask_last_occurence)
        full, _, _, _, loss = self.match_layer(prop_feat, prop_m, [tplt_feat.view(F, -1)[:O]], mask_last_occurence[0, :O],
                                               prop_score, None if targets is None else targets[0][:O])
        if O < F:                                                  # the slots behind the live templates stay zero (:78-80)
            out = full.new_zeros((1, F, H, W))
            out[0, :O] = full
        else:
            out = full.unsqueeze(0)
        return out, [loss["cost_loss"]] if len(loss) > 0 else []

    def _valid_layout_of_clip(self, tplt_valid_batch):
        """``_valid_layout`` once per CLIP: the reference builds ``tplt_valid_batch`` when a clip starts and hands the same
        tensor to every frame step (trainer.py:113-121, evaluator.py:114-126), so the answer for this very tensor object at
        this very version (no in-place write since) is kept -- a frame step after the first costs no host sync.  The tensor
        is held weakly; a new tensor, or the same one after any in-place edit, is read again.  Contract: writes that do not
        move ``_version`` (through ``.data``, ``set_``, a raw device copy) are not seen -- hand such a clip a new tensor."""
        c = _VALID_CACHE.get(self)                                 # (outside the module's __dict__: deepcopy / pickle safe)
        if c is not None and c[0]() is tplt_valid_batch and c[1] == tplt_valid_batch._version:
            return list(c[2][0]), c[2][1]                          # (a copy of the list: a caller editing it must not edit the cache)
        got = self._valid_layout(tplt_valid_batch)
        try:
            _VALID_CACHE[self] = (weakref.ref(tplt_valid_batch), tplt_valid_batch._version, got)
        except TypeError:                                          # (an object that cannot be weakly referenced)
            _VALID_CACHE.pop(self, None)
        return list(got[0]), got[1]

    @staticmethod
    def _valid_layout(tplt_valid_batch, n_tplt_hint=None):
        """-> (live templates per video, row_scale [B,F] or None).  ONE host sync for the whole batch.  The reference
        selects ``mask_last_occurence[bid, :O]`` with O = valid.sum() (dmm_model.py:117,124) and builds
        ``OF_matrix = diag(valid)[:O]``: when the valid slots are a prefix (what ``ohw_mask2boxlist`` normally
        produces) that is a plain row selection and row_scale is None; otherwise slots i < O with valid[i] == 0 get
        zero template features and zero output rows, reproduced by ``row_scale``."""
        rows = [[int(round(float(v))) for v in r] for r in tplt_valid_batch.tolist()]
        n_tplt = [sum(r) for r in rows]
        if all(all(v == 1 for v in r[:o]) for r, o in zip(rows, n_tplt)):
            return n_tplt, None
        scale = [[float(v) if i < o else 0.0 for i, v in enumerate(r)] for r, o in zip(rows, n_tplt)]
        return n_tplt, _lib.small_to_device(scale, torch.float32, tplt_valid_batch.device)

    def _per_video(self, prop_feat, prop_m, prop_score, tplt_feat, mask_last_occurence, tplt_valid_batch, n_tplt,
                   targets, skip):
        """algo 'hun' (and any future non-batched solver): the reference's loop, one ``MatchModel`` call per video
        (dmm_model.py:62-82 / :115-139) with the OF/FO selection matmuls of ``prepare_tplt_feature`` (:144-158)."""
        B, F, H, W = CHECK4D(mask_last_occurence)
        outs, losses = [], []
        for b in range(B):
            O = n_tplt[b]
            if skip[b]:
                outs.append(mask_last_occurence.new_zeros(F, H, W))
                losses.append(prop_feat[b].sum() * 0)
                continue
            OF = torch.diag(tplt_valid_batch[b].float())[:O, :]
            tfv = torch.mm(OF, tplt_feat[b].view(F, -1))
            full, _, _, _, loss = self.match_layer(prop_feat[b], prop_m[b], [tfv], mask_last_occurence[b, :O],
                                                   prop_score[b], None if targets is None else targets[b][:O])
            outs.append(torch.mm(OF.t(), full.reshape(O, -1)).view(F, H, W))
            losses.append(loss["cost_loss"] if len(loss) > 0 else prop_feat[b].sum() * 0)
        return torch.stack(outs, 0), losses

    @staticmethod
    def _out_mask_last(full, mask_last_occurence, skip, alias_ok=False):
        """``out_mask_last`` (dmm_model.py:66-69, :78-80): the matched masks, except that a skipped video keeps its
        incoming planes.  The reference returns it as a tensor of its own, and its decoder step then writes the refined
        masks INTO it (evaluator.py:205) -- so by default this is a separate tensor too.  ``alias_ok`` (the caller
        promises not to edit either result in place; ``video.FrameLoop`` without a decoder does): with no video
        skipped it IS ``full`` -- MatchModel returns the same tensor twice (match_model.py:47) -- and the
        [B,F,H,W] copy is saved."""
        if not any(skip):
            return full if alias_ok else full.clone()
        keep = _lib.small_to_device([bool(s_) for s_ in skip], torch.bool, full.device)
        return torch.where(keep[:, None, None, None], mask_last_occurence.to(full.dtype), full)

    @staticmethod
    def _template_features(tplt_dict, B):
        """The template-feature entry of every video.  The reference passes ``tplt_dict[b]['feat']`` (a list) on
        (dmm_model.py:151-157) and MatchModel averages the cosines over its entries (match_model.py:71-76); DMM-Net only
        ever stores ONE entry (templates are fixed from frame 0, dmm_model.py:44) and the batched launch is built for
        that: a longer list is refused here rather than silently truncated ('hun' / per-video calls take any length)."""
        for b in range(B):
            assert len(tplt_dict[b]["feat"]) == 1, "the batched matching path takes one template-feature entry per video"
        return [tplt_dict[b]["feat"][0] for b in range(B)]

    @staticmethod
    def _proposal_fields(proposals):
        prop_m = [p.get_field("mask").squeeze(1) for p in proposals]
        prop_score = [p.get_field("objectness") if "objectness" in p.fields() else p.get_field("scores")
                      for p in proposals]
        return prop_m, prop_score

    # ---- dmm_model.py:48-86 ------------------------------------------------------------------------
    def inference(self, infos, proposals, backbone_feature, mask_last_occurence, tplt_dict, target=None):
        extra_frame, tplt_valid_batch = infos["extra_frame"], infos["valid"]
        B, F, H, W = CHECK4D(mask_last_occurence)
        boxes_per_image = [len(box) for box in proposals]
        prop_feat = self.feature_extractor(backbone_feature, proposals).split(boxes_per_image, dim=0)
        prop_m, prop_score = self._proposal_fields(proposals)
        # live templates per video: one host sync unless the caller (video.FrameLoop) already knows them
        if infos.get("n_tplt"):
            n_tplt, row_scale = list(infos["n_tplt"]), infos.get("row_scale")
        else:
            n_tplt, row_scale = self._valid_layout_of_clip(tplt_valid_batch)
        skip = [n_tplt[b] == 0 or bool(extra_frame[b]) for b in range(B)]
        tplt_feat = self._template_features(tplt_dict, B)
        tg = None
        if target is not None:
            tg = torch.stack([target[b] for b in range(B)], 0)
        if self.match_algo != "relax":
            full, _ = self._per_video(list(prop_feat), prop_m, prop_score, tplt_feat, mask_last_occurence,
                                      tplt_valid_batch, n_tplt, tg, skip)
        else:
            packed = None
            if all("mask_packed" in p.fields() for p in proposals):
                packed = [p.get_field("mask_packed") for p in proposals]
            if B == 1 and packed is None and row_scale is None and not skip[0]:
                full, _ = self._match_single(prop_feat[0], prop_m[0], prop_score[0], tplt_feat[0], mask_last_occurence,
                                             n_tplt[0], tg)
            else:
                full, _ = self._match_batch(list(prop_feat), prop_m, prop_score, tplt_feat, mask_last_occurence, n_tplt,
                                            tg, skip, row_scale, packed)
        return full, tplt_dict, [], self._out_mask_last(full, mask_last_occurence, skip, bool(infos.get("alias_ok")))

    # ---- dmm_model.py:88-142 -----------------------------------------------------------------------
    def forward(self, args, proposals, backbone_feature, mask_last_occurence, tplt_dict, tplt_valid_batch, targets):
        B, F, H, W = CHECK4D(mask_last_occurence)
        assert targets is not None                                  # training mode must have targets (:128)
        boxes_per_image = [len(box) for box in proposals]
        prop_feat = self.feature_extractor(backbone_feature, proposals).split(boxes_per_image, dim=0)
        prop_m, prop_score = self._proposal_fields(proposals)
        n_tplt, row_scale = self._valid_layout_of_clip(tplt_valid_batch)   # one host sync per clip for the whole batch
        skip = [n_tplt[b] == 0 for b in range(B)]
        tplt_feat = self._template_features(tplt_dict, B)
        if self.match_algo != "relax":
            full, loss = self._per_video(list(prop_feat), prop_m, prop_score, tplt_feat, mask_last_occurence,
                                         tplt_valid_batch, n_tplt, targets, skip)
        elif B == 1 and row_scale is None and not skip[0]:
            full, loss = self._match_single(prop_feat[0], prop_m[0], prop_score[0], tplt_feat[0], mask_last_occurence,
                                            n_tplt[0], targets)
        else:
            full, loss = self._match_batch(list(prop_feat), prop_m, prop_score, tplt_feat, mask_last_occurence,
                                           n_tplt, targets, skip, row_scale)
        per_video = loss.unbind(0) if torch.is_tensor(loss) else loss        # (one stack in backward, not B x zeros + copy)
        match_loss = [prop_feat[b].sum() * 0 if skip[b] else per_video[b] for b in range(B)]      # :121
        return full, tplt_dict, match_loss, self._out_mask_last(full, mask_last_occurence, skip)
