"""Autograd glue of the matching layer (one frame = a batch of 1 for the HIP kernels).

Forward: ops.iou_counts -> ops.feature_normalize -> ops.relax_match -> ops.mask_mix, all on the
gfx950 library.  Gradients flow to ``proposed_feature`` and ``template_feature`` only, exactly as in
the reference (the IoU part of the cost is computed under ``no_grad``/detached, match_helper.py:20-28;
the greedy init carries no grad, relax_match.py:45-55).

Backward (reference: torch autograd through ~950 nodes at (10,5)):
  dRb      = dOut @ mask_p^T                        (bandwidth bound, ops.mask_mix_bwd)
  dsim     = reverse sweep through the relax iterations (ops.relax_match_bwd: the kernel re-runs the
             forward, taping 1 relu bit per element and 1 column bit per sweep, then walks it back)
  dfeat    = cosine / normalisation backward        (small dense algebra, torch on device)
"""
from __future__ import annotations

from typing import List, Optional

import torch

from . import ops


def _greedy_onehot(neg_cost: torch.Tensor) -> torch.Tensor:
    """Greedy init of relax_matching on a [B,n,m] cost batch (relax_match.py:45-55) as a one-hot
    matrix; device side via the solver kernel with max_iter = 0 (X_list == [X0])."""
    return ops.relax_solve(neg_cost, 0, 0, 0.0)["X"]


def hungarian_onehot(cost: torch.Tensor) -> torch.Tensor:
    """algo == 'hun' (relax_match.py:120-126): scipy on the host, like the reference (which also
    round-trips through numpy); not differentiable."""
    from scipy.optimize import linear_sum_assignment
    import numpy as np
    c = cost.detach().cpu().numpy()
    r, col = linear_sum_assignment(c)
    X = np.zeros_like(c)
    X[r, col] = 1
    return torch.from_numpy(X).float().to(cost.device)


class _MatchLayerFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pf, tf, pm, tm, sc, targets, score_weight, max_iter, proj_iter, lr, is_test):
        # pf [P,D], tf [O,D] (already averaged/normalised handling done by the caller), pm [P,H,W], tm [O,H,W]
        P, O = pm.shape[0], tm.shape[0]
        pm_b, tm_b = pm.unsqueeze(0), tm.unsqueeze(0)
        inter, ap, at = ops.iou_counts(pm_b, tm_b)
        pn, pnorm = ops.feature_normalize(pf.unsqueeze(0), want_norms=True)
        tn, tnorm = ops.feature_normalize(tf.unsqueeze(0), want_norms=True)
        cos = ops.cosine(tn, pn)
        r = ops.relax_match(cos, inter, ap, at, sc.unsqueeze(0), score_weight=score_weight, max_iter=max_iter,
                            proj_iter=proj_iter, lr=lr, is_test=is_test)
        full = ops.mask_mix(r["Rb"], pm_b)
        cost_loss = pf.new_zeros(())
        gt = None
        if targets is not None:
            # compute_matching_loss (match_helper.py:30-49): IoU(proposal>0.5, targets) -> greedy one-hot -> MSE
            tg = targets.unsqueeze(0).to(pm.dtype)
            gi, gap, gat = ops.iou_counts(pm_b, tg)
            union = (gap.unsqueeze(1) + gat.unsqueeze(2) - gi).float() + 1e-6
            gt_iou = gi.float() / union
            gt = _greedy_onehot(-gt_iou)
            diff = cos - gt
            cost_loss = (diff * diff).mean()
        ctx.save_for_backward(pn, tn, pnorm, tnorm, pf, tf, cos, r["sim"], r["Rb"], sc, pm,
                              gt if gt is not None else cos)
        ctx.has_targets = targets is not None
        ctx.cfg = (score_weight, max_iter, proj_iter, lr, is_test)
        ctx.mark_non_differentiable(r["iters"])
        return full[0], r["match_score"][0], r["det_score"][0], cost_loss, r["iters"]

    @staticmethod
    def backward(ctx, d_full, d_ms, d_ds, d_loss, _d_iters):
        from .backward import match_layer_backward
        return match_layer_backward(ctx, d_full, d_ms, d_ds, d_loss)


def match_layer_function(proposed_feature, proposed_mask, template_feature: List[torch.Tensor], mask_last_occurence,
                         proposal_score, targets: Optional[torch.Tensor], *, score_weight, max_iter, proj_iter, lr,
                         is_test, algo="relax"):
    pm = proposed_mask.float()
    tm = mask_last_occurence.float()
    pf = proposed_feature.float()
    sc = proposal_score.float()
    if len(template_feature) != 1:
        # feature_sim is the mean of the per-entry cosines (match_model.py:71-76); DMM-Net always passes one
        # entry (dmm_model.py:44, templates are fixed from frame 0), longer lists are rejected loudly.
        raise NotImplementedError("template_feature lists longer than 1 are never produced by DMM-Net "
                                  "(dmm_model.py:44); pass a single [O,D] tensor")
    tf = template_feature[0].float()
    if algo == "hun":
        return _hungarian_forward(pf, tf, pm, tm, sc, targets, score_weight, is_test)
    full, ms, ds, loss, _ = _MatchLayerFn.apply(pf, tf, pm, tm, sc, targets, score_weight, max_iter, proj_iter, lr,
                                                is_test)
    return full, ms, ds, loss


def _hungarian_forward(pf, tf, pm, tm, sc, targets, score_weight, is_test):
    """algo 'hun' slot (match_model.py:122-123): same cost matrix, assignment from scipy (host)."""
    P, O = pm.shape[0], tm.shape[0]
    pm_b, tm_b = pm.unsqueeze(0), tm.unsqueeze(0)
    inter, ap, at = ops.iou_counts(pm_b, tm_b)
    pn = ops.feature_normalize(pf.unsqueeze(0))
    tn = ops.feature_normalize(tf.unsqueeze(0))
    cos = ops.cosine(tn, pn)
    r = ops.relax_match(cos, inter, ap, at, sc.unsqueeze(0), score_weight=score_weight, max_iter=0, proj_iter=0,
                        lr=0.0, is_test=is_test)
    sim = r["sim"][0]
    Pp = ops.padded_width(P, O)
    simp = sim.new_zeros((O, Pp))
    simp[:, :P] = sim
    R = hungarian_onehot(-simp)
    maxv = R.max(dim=1, keepdim=True)[0]
    logic = (R == maxv).float() if is_test else (R > 0.01).float()
    Rb = R * logic
    full = ops.mask_mix(Rb.unsqueeze(0), pm_b)[0]
    ms = (R.clamp(0, 1) * simp).max(1)[0]
    scp = sc.new_zeros(Pp)
    scp[:P] = sc
    ds = (scp.view(1, -1) * Rb).sum(1)
    loss = pf.new_zeros(())
    if targets is not None:
        gi, gap, gat = ops.iou_counts(pm_b, targets.unsqueeze(0).to(pm.dtype))
        gt_iou = gi.float() / ((gap.unsqueeze(1) + gat.unsqueeze(2) - gi).float() + 1e-6)
        gt = _greedy_onehot(-gt_iou)
        loss = ((cos - gt) ** 2).mean()
    return full, ms, ds, loss
