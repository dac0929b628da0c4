"""``MatchModel`` -- drop-in for the reference's ``dmm.modules.match_model.MatchModel``.

Same constructor (``MatchModel(cfgs, is_test)``), same ``forward`` signature and 5-tuple return as
reference ``dmm/modules/match_model.py:13-47``; the arithmetic runs in the gfx950 HIP library
(``libdmm_match.so``) through its C ABI instead of eager torch ops:

  compute_cost_matrix      (match_model.py:49-91)   -> ops.iou_counts + ops.feature_normalize +
                                                       the prologue of ops.relax_match
  match_with_first_frame   (match_model.py:93-148)  -> ops.relax_match + ops.mask_mix
  compute_matching_loss    (match_helper.py:30-49)  -> ops.iou_counts on the targets + device greedy init

``cfgs`` keys consumed (as the reference): ``cfgs['matching']['algo']`` in {'relax', 'hun'},
``relax_max_iter``, ``relax_proj_iter``, ``relax_learning_rate``, ``score_weight``.

Errors are ``AssertionError`` for rank / shape violations like the reference's CHECK* helpers
(``dmm/utils/checker.py:4-30``).  The layer has no parameters or buffers (checkpoint neutral).
There is no CPU path: CPU tensors raise ``DmmError``.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.nn as nn

from .autograd import match_layer_function


def CHECK3D(t):
    assert len(t.shape) == 3, "get {} {}".format(t.shape, len(t.shape))
    return t.shape


def CHECKEQ(a, b):
    assert a == b, "get {} {}".format(a, b)


class MatchModel(nn.Module):
    def __init__(self, cfgs={}, is_test=0):
        super().__init__()
        self.cfgs = cfgs
        self.is_test = is_test
        self.match_algo = cfgs["matching"]["algo"]
        self.max_iter = self.cfgs["relax_max_iter"]
        self.proj_iter = self.cfgs["relax_proj_iter"]
        self.relax_lr = self.cfgs["relax_learning_rate"]
        assert self.match_algo == "relax" or self.match_algo == "hun"

    def forward(self, proposed_feature, proposed_mask, template_feature: List[torch.Tensor], mask_last_occurence,
                proposal_score, targets: Optional[torch.Tensor] = None):
        """One frame of one video (reference match_model.py:24-47).

        proposed_feature [P,D]; proposed_mask [P,H,W]; template_feature: list of [O,D];
        mask_last_occurence [O,H,W]; proposal_score [P]; targets [O,H,W] or None.
        Returns (full_outmask [O,H,W], match_score [O], det_score [O], full_outmask, match_loss dict).
        """
        CHECK3D(proposed_mask)
        CHECK3D(mask_last_occurence)
        n_prop = proposed_mask.shape[0]
        n_tplt = template_feature[0].shape[0]
        CHECKEQ(proposed_mask.shape[-2:], mask_last_occurence.shape[-2:])
        CHECKEQ(proposal_score.shape[0], n_prop)
        CHECKEQ(mask_last_occurence.shape[0], n_tplt)
        CHECKEQ(proposed_feature.shape[0], n_prop)
        if targets is not None:
            CHECK3D(targets)
            CHECKEQ(proposed_mask.shape[-1], targets.shape[-1])
        # feature_sim is the MEAN over the template-feature list (match_model.py:71-76); the product
        # always passes a single entry (dmm_model.py:44), longer lists are averaged on the cosines.
        full_outmask, match_score, det_score, cost_loss = match_layer_function(
            proposed_feature, proposed_mask, list(template_feature), mask_last_occurence, proposal_score, targets,
            score_weight=float(self.cfgs["score_weight"]), max_iter=int(self.max_iter), proj_iter=int(self.proj_iter),
            lr=float(self.relax_lr), is_test=int(bool(self.is_test)), algo=self.match_algo)
        match_loss = {}
        if targets is not None:
            match_loss.update({"cost_loss": cost_loss})
        return full_outmask, match_score, det_score, full_outmask, match_loss
