"""Autograd glue of the matching layer: B frames per call (B = 1 reproduces one reference call).

Forward: ops.iou_counts -> ops.feature_normalize -> ops.cosine -> ops.relax_match -> ops.mask_mix, all on
the gfx950 library.  Gradients flow to ``proposed_feature`` and ``template_feature`` (and to
``proposed_mask`` if it requires grad), exactly as in the reference: the IoU part of the cost is computed
under ``no_grad``/detached (match_helper.py:20-28), the greedy init carries no grad (relax_match.py:45-55).

Backward: dmm_net_amd/backward.py.
"""
from __future__ import annotations

from typing import List, Optional

import torch

from . import _lib, ops


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


def matching_loss(pm_b, targets_b, cos, n_valid=None, m_valid=None, counts=None):
    """compute_matching_loss (match_helper.py:30-49) for B frames -> (loss [B], gt one-hot [B,O,P]).
    IoU(proposal > 0.5, targets) -> greedy one-hot of -IoU (relax_matching(..., 0, 0, 0)) -> MSE with cos.
    ``counts`` = (inter2, area_p, area_t2) when the cost pass already intersected the targets (iou_counts_dual)."""
    B, O, P = cos.shape
    if counts is not None:
        gi, gap, gat = counts
    else:
        gi, gap, gat = ops.iou_counts(pm_b, targets_b.to(pm_b.dtype), n_valid, m_valid)
    union = (gap.unsqueeze(1) + gat.unsqueeze(2) - gi).float() + 1e-6
    gt_iou = gi.float() / union
    gt = ops.relax_solve(-gt_iou, 0, 0, 0.0, rows_valid=m_valid, cols_valid=n_valid)["X"]
    diff = cos - gt
    if n_valid is None and m_valid is None:
        return (diff * diff).flatten(1).mean(1), gt, None
    nv = n_valid if n_valid is not None else torch.full((B,), P, dtype=torch.int32, device=cos.device)
    mv = m_valid if m_valid is not None else torch.full((B,), O, dtype=torch.int32, device=cos.device)
    live = (torch.arange(O, device=cos.device)[None, :, None] < mv[:, None, None]) & \
           (torch.arange(P, device=cos.device)[None, None, :] < nv[:, None, None])
    cnt = (nv * mv).clamp_min(1).float()
    sq = torch.where(live, diff * diff, torch.zeros_like(diff))
    return sq.flatten(1).sum(1) / cnt, gt, (live, cnt)


# the training call as one fused library call each way (tests pin the granular chain against it by switching this off)
_FUSED_TRAIN = True
_EMPTY = torch.zeros(())


def _no_grad_view(*tensors):
    """Under ``torch.no_grad()`` a Function's ``ctx.needs_input_grad`` still says what the inputs' ``requires_grad`` says (it
    ignores the grad mode): the layer would keep the solver's tape for a backward that cannot come.  Detached
    views make the two agree."""
    if torch.is_grad_enabled():
        return tensors
    return tuple(t.detach() if isinstance(t, torch.Tensor) else t for t in tensors)


class _MatchLayerFn(torch.autograd.Function):
    """pf [B,P,D], tf [T,B,O,D] (T template-feature entries, DMM-Net uses T = 1), pm [B,P,H,W], tm [B,O,H,W],
    sc [B,P], targets [B,O,H,W] | None."""

    @staticmethod
    def forward(ctx, pf, tf, pm, tm, sc, targets, n_valid, m_valid, score_weight, max_iter, proj_iter, lr, is_test,
                counts=None):
        T = tf.shape[0]
        tcounts = None
        ctx.set_materialize_grads(False)             # an unused output's gradient arrives as None, not as a zero tensor
        ctx.fused = False
        if counts is None and T == 1 and _FUSED_TRAIN and not (isinstance(pm, torch.Tensor) and pm.requires_grad):
            # the common case as ONE library call each way (dmm_match_train_forward / _backward): a one-frame call is
            # host bound otherwise (bench.py --config dropin)
            pf_c, tf_c, sc_c = pf.contiguous(), tf[0].contiguous(), sc.contiguous()
            got = ops.match_train_forward(pm, tm, targets, pf_c, tf_c, sc_c, n_valid, m_valid, score_weight=score_weight,
                                          max_iter=max_iter, proj_iter=proj_iter, lr=lr, is_test=is_test,
                                          want_tape=ctx.needs_input_grad[0] or ctx.needs_input_grad[1])
            if got is not None:
                full, ms, ds, loss, iters, saved, ctx.taped = got
                ctx.fused = True
                ctx.frame_planes = pm if isinstance(pm, ops.FramePlanes) else None
                empty = _EMPTY
                ctx.save_for_backward(pf_c, tf_c, sc_c, saved, empty if ctx.frame_planes is not None else pm,
                                      n_valid if n_valid is not None else empty, m_valid if m_valid is not None else empty,
                                      iters)
                ctx.has_targets = targets is not None
                ctx.ragged = (n_valid is not None, m_valid is not None)
                ctx.cfg = (score_weight, max_iter, proj_iter, lr, is_test)
                ctx.n_tplt = tm.shape[1]
                ctx.mark_non_differentiable(iters)
                if loss is None:
                    loss = pf.new_zeros((pf.shape[0],))
                return full, ms, ds, loss, iters
        if counts is not None:                       # integer tables from elsewhere (1-bit planes of the paste kernel)
            inter, ap, at = counts
        elif targets is not None and tm.shape[1] <= 16:
            # training: one pass over the proposal planes for both IoU tables (templates and targets)
            (inter, ap, at), (gi, gat) = ops.iou_counts_dual(pm, tm, targets.to(tm.dtype), n_valid, m_valid)
            tcounts = (gi, ap, gat)
        else:
            inter, ap, at = ops.iou_counts(pm, tm, n_valid, m_valid)
        pn, pnorm = ops.feature_normalize(pf, want_norms=True)
        tn, tnorm = ops.feature_normalize(tf, want_norms=True)              # [T,B,O,D], [T,B,O]
        # feature_sim = mean over the template-feature entries (match_model.py:71-76): zeros, += each, /= T
        cos = ops.cosine(tn[0], pn, n_valid, m_valid)
        if T > 1:
            cos = torch.zeros_like(cos) + cos
            for t in range(1, T):
                cos = cos + ops.cosine(tn[t], pn, n_valid, m_valid)
            cos = cos / T
        r = ops.relax_match(cos, inter, ap, at, sc, score_weight=score_weight, max_iter=max_iter, proj_iter=proj_iter,
                            lr=lr, is_test=is_test, n_valid=n_valid, m_valid=m_valid)
        full = ops.mask_mix(r["Rb"], pm, n_valid, m_valid, shared=not is_test)   # train mode: rows share planes
        B = pf.shape[0]
        cost_loss = pf.new_zeros((B,))
        gt, live, cnt = None, None, None
        if targets is not None:
            cost_loss, gt, lc = matching_loss(pm, targets, cos, n_valid, m_valid, counts=tcounts)
            if lc is not None:
                live, cnt = lc
        empty = pf.new_zeros(())
        ctx.frame_planes = pm if isinstance(pm, ops.FramePlanes) else None     # per-frame tensors + pointer table
        ctx.save_for_backward(pn, tn, pnorm, tnorm, pf, tf, cos, r["sim"], r["Rb"], sc,
                              empty if ctx.frame_planes is not None else pm,
                              gt if gt is not None else empty, live if live is not None else empty,
                              cnt if cnt is not None else empty,
                              n_valid if n_valid is not None else empty, m_valid if m_valid is not None else empty)
        ctx.has_targets = targets is not None
        ctx.ragged = (n_valid is not None, m_valid is not None)
        ctx.cfg = (score_weight, max_iter, proj_iter, lr, is_test)
        ctx.mark_non_differentiable(r["iters"])
        return full, r["match_score"], r["det_score"], cost_loss, r["iters"]

    @staticmethod
    def backward(ctx, d_full, d_ms, d_ds, d_loss, _d_iters):
        if ctx.fused:
            need_pf, need_tf = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
            if not (need_pf or need_tf):
                return (None,) * 14
            pf, tf, sc, saved, pm, n_valid, m_valid, iters = ctx.saved_tensors
            score_weight, max_iter, proj_iter, lr, is_test = ctx.cfg
            g_t, g_p = ops.match_train_backward(
                ctx.frame_planes if ctx.frame_planes is not None else pm, pf, tf, sc, saved, ctx.has_targets, d_full, d_ms,
                d_ds, d_loss, n_valid if ctx.ragged[0] else None, m_valid if ctx.ragged[1] else None, ctx.n_tplt,
                score_weight=score_weight, max_iter=max_iter, proj_iter=proj_iter, lr=lr, is_test=is_test, iters=iters,
                taped=ctx.taped)
            return (g_p if need_pf else None, g_t.unsqueeze(0) if need_tf else None) + (None,) * 12
        from .backward import match_layer_backward
        return match_layer_backward(ctx, d_full, d_ms, d_ds, d_loss)


class _MatchFrameFn(torch.autograd.Function):
    """ONE frame exactly as the reference's trainer hands it over (dmm_model.py:130-132): pf [P,D], tf [O,D], pm [P,H,W],
    tm [O,H,W], sc [P], targets [O,H,W] | None -> (full [O,H,W], match_score [O], det_score [O], cost_loss ()).  The
    fused library calls of ``_MatchLayerFn`` without the batch axis: no unsqueeze / select nodes around the layer in the
    autograd graph (a one-frame call is host bound; bench.py --config dropin)."""

    @staticmethod
    def forward(ctx, pf, tf, pm, tm, sc, targets, score_weight, max_iter, proj_iter, lr, is_test):
        ctx.set_materialize_grads(False)
        pf_c, tf_c, sc_c = pf.contiguous(), tf.contiguous(), sc.contiguous()
        got = ops.match_train_forward(pm, tm, targets, pf_c, tf_c, sc_c, None, None, score_weight=score_weight,
                                      max_iter=max_iter, proj_iter=proj_iter, lr=lr, is_test=is_test, one_frame=True,
                                      want_tape=ctx.needs_input_grad[0] or ctx.needs_input_grad[1])
        if got is None:
            raise _lib.DmmError("dmm_match_train_forward refused a shape frame_fused_ok() accepted")
        full, ms, ds, loss, iters, saved, ctx.taped = got
        ctx.save_for_backward(pf_c, tf_c, sc_c, saved, pm, iters)
        ctx.has_targets = targets is not None
        ctx.cfg = (score_weight, max_iter, proj_iter, lr, is_test)
        ctx.n_tplt = tm.shape[0]
        if loss is None:
            return full, ms, ds
        return full, ms, ds, loss

    @staticmethod
    def backward(ctx, d_full, d_ms, d_ds, d_loss=None):
        need_pf, need_tf = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        if not (need_pf or need_tf):
            return (None,) * 11
        pf, tf, sc, saved, pm, iters = ctx.saved_tensors
        score_weight, max_iter, proj_iter, lr, is_test = ctx.cfg
        g_t, g_p = ops.match_train_backward(pm, pf, tf, sc, saved, ctx.has_targets, d_full, d_ms, d_ds, d_loss, None, None,
                                            ctx.n_tplt, score_weight=score_weight, max_iter=max_iter, proj_iter=proj_iter,
                                            lr=lr, is_test=is_test, one_frame=True, iters=iters, taped=ctx.taped)
        return (g_p if need_pf else None, g_t if need_tf else None) + (None,) * 9


def frame_fused_ok(pm, tm) -> bool:
    """Shapes ``dmm_match_train_forward`` takes (the fast kernels' envelope) -- decided BEFORE the autograd function is
    chosen, from the shapes alone."""
    P, O = pm.shape[0], tm.shape[0]
    return (_FUSED_TRAIN and 0 < O <= _lib.MAX_TEMPLATES and 0 < P and ops.padded_width(P, O) <= _lib.MAX_PROPOSALS
            and not pm.requires_grad and _lib.get_option("FORCE_WIDE") != 1)


class _RaggedPadFn(torch.autograd.Function):
    """Per-video blocks [P_b, D] -> [B, P_max, D] by ONE launch (``ops.ragged_pad``), differentiable: the gradient of a
    block is the view ``g[b, :P_b]`` of the batch's gradient -- no kernel.  (The trainer's batching step used to be a
    zero fill + one in-place copy per video, each with a CopySlices node whose backward is a clone + fill + copy: ~35
    tensor-op kernels around the 10 of the layer itself for 4 videos, round-5 timeline of ``DMM_Model.forward``.)"""

    @staticmethod
    def forward(ctx, counts, P_max, table, *blocks):
        ctx.rows = [int(b.shape[0]) for b in blocks]
        return ops.ragged_pad(list(blocks), int(P_max), counts, table)

    @staticmethod
    def backward(ctx, g):
        return (None, None, None) + tuple(g[b, :p] for b, p in enumerate(ctx.rows))


def ragged_pad(blocks, P_max, counts, table=None):
    """``ops.ragged_pad`` with a gradient path to the blocks that ask for one (``table``: see there)."""
    if torch.is_grad_enabled() and any(b.requires_grad for b in blocks):
        return _RaggedPadFn.apply(counts, int(P_max), table, *blocks)
    return ops.ragged_pad(list(blocks), int(P_max), counts, table)


def match_layer_batched(pf, pm, tf, tm, sc, targets=None, n_valid=None, m_valid=None, *, score_weight, max_iter,
                        proj_iter, lr, is_test, counts=None):
    """B frames through the layer with autograd.  ``tf`` is [B,O,D] or, for several template-feature entries,
    [T,B,O,D].  Returns (full_outmask [B,O,H,W], match_score [B,O], det_score [B,O], cost_loss [B], iters [B])."""
    if tf.dim() == 3:
        tf = tf.unsqueeze(0)
    if isinstance(pm, ops.FramePlanes):                      # the caller built the pointer table (DMM_Model: one upload)
        tm = tm if tm.dtype == pm.dtype else tm.to(pm.dtype)
        if targets is not None and targets.dtype != pm.dtype:
            targets = targets.to(pm.dtype)
        if not torch.is_grad_enabled() and counts is None and tf.shape[0] == 1 and _FUSED_TRAIN:
            # no gradient can be asked for (the evaluator's call): the library call itself, without an autograd node around it
            # (Function.apply costs ~10-15 us of host time per call, and the zero loss tensor of the no-targets case a launch)
            got = ops.match_train_forward(pm, tm, targets, pf.detach().float().contiguous(), tf[0].detach().float().contiguous(),
                                          sc.detach().float().contiguous(), n_valid, m_valid, score_weight=float(score_weight),
                                          max_iter=int(max_iter), proj_iter=int(proj_iter), lr=float(lr), is_test=int(is_test),
                                          want_tape=False)
            if got is not None:
                return got[0], got[1], got[2], got[3], got[4]
        pf, tf = _no_grad_view(pf, tf)
        return _MatchLayerFn.apply(pf.float(), tf.float(), pm, tm, sc.float(), targets, n_valid, m_valid,
                                   float(score_weight), int(max_iter), int(proj_iter), float(lr), int(is_test), counts)
    if isinstance(pm, (list, tuple)):
        # one tensor per frame (DMM_Model's per-video proposal planes): matched in place through the pointer-table
        # entry points; only a mask gradient or mixed dtypes force the stacked copy
        if any(t.requires_grad for t in pm) or len({t.dtype for t in pm}) > 1 or pm[0].dtype not in ops._DT:
            pm = ops.FramePlanes([t.float() for t in pm]).stacked()
        else:
            pm = ops.FramePlanes(pm)
            if n_valid is None:
                n_valid = pm.n_valid()
            if tm.dtype != pm.dtype:
                tm = tm.to(pm.dtype)
            if targets is not None:
                targets = targets.to(pm.dtype)
            pf, tf = _no_grad_view(pf, tf)
            return _MatchLayerFn.apply(pf.float(), tf.float(), pm, tm, sc.float(), targets, n_valid, m_valid,
                                       float(score_weight), int(max_iter), int(proj_iter), float(lr), int(is_test), counts)
    # 16-bit mask planes go to the kernels as they are (half the bytes of the cost pass; values are only thresholded and
    # scaled); anything else is matched in fp32 like the reference
    if not (pm.dtype == tm.dtype and pm.dtype in (torch.float16, torch.bfloat16)):
        pm, tm = pm.float(), tm.float()
    pf, tf = _no_grad_view(pf, tf)
    return _MatchLayerFn.apply(pf.float(), tf.float(), pm, tm, sc.float(),
                               None if targets is None else targets.float(), n_valid, m_valid, float(score_weight),
                               int(max_iter), int(proj_iter), float(lr), int(is_test), counts)


def match_layer_function(proposed_feature, proposed_mask, template_feature: List[torch.Tensor], mask_last_occurence,
                         proposal_score, targets: Optional[torch.Tensor], *, score_weight, max_iter, proj_iter, lr,
                         is_test, algo="relax"):
    """One frame (the reference's call): unsqueeze to B = 1."""
    # feature_sim is the mean of the per-entry cosines (match_model.py:71-76); DMM-Net itself always passes one
    # entry (dmm_model.py:44, templates are fixed from frame 0)
    tf = template_feature[0] if len(template_feature) == 1 else torch.stack(list(template_feature), 0)
    if algo == "hun":
        assert len(template_feature) == 1, "algo 'hun' is wired for a single template-feature entry"
        return _hungarian_forward(proposed_feature.float(), tf.float(), proposed_mask.float(),
                                  mask_last_occurence.float(), proposal_score.float(), targets, score_weight, is_test)
    needs_grad = torch.is_grad_enabled() and (proposed_feature.requires_grad or tf.requires_grad
                                              or proposed_mask.requires_grad)
    if targets is None and not needs_grad and tf.dim() == 2:
        # inference (the evaluator's call, dmm_model.py:75-77): one fused C-ABI call, no autograd bookkeeping
        pm, tm = proposed_mask, mask_last_occurence
        if not (pm.dtype == tm.dtype and pm.dtype in (torch.float16, torch.bfloat16)):
            pm, tm = pm.float(), tm.float()
        full, ms, ds = ops.match_forward_frame(pm, tm, proposed_feature, tf, proposal_score, score_weight=score_weight,
                                               max_iter=max_iter, proj_iter=proj_iter, lr=lr, is_test=is_test)
        return full, ms, ds, None                            # (no targets: MatchModel returns an empty loss dict)
    if tf.dim() == 2:
        pm, tm, tg = proposed_mask, mask_last_occurence, targets
        if not (pm.dtype == tm.dtype and pm.dtype in (torch.float16, torch.bfloat16)):
            pm, tm = pm.float(), tm.float()
        if frame_fused_ok(pm, tm):
            proposed_feature, tf = _no_grad_view(proposed_feature, tf)
            out = _MatchFrameFn.apply(proposed_feature.float(), tf.float(), pm, tm, proposal_score.float(),
                                      None if tg is None else tg.to(pm.dtype), float(score_weight), int(max_iter),
                                      int(proj_iter), float(lr), int(is_test))
            return out if tg is not None else out + (None,)
    full, ms, ds, loss, _ = match_layer_batched(
        proposed_feature.unsqueeze(0), proposed_mask.unsqueeze(0),
        tf.unsqueeze(0) if tf.dim() == 2 else tf.unsqueeze(1), mask_last_occurence.unsqueeze(0),
        proposal_score.unsqueeze(0), None if targets is None else targets.unsqueeze(0), score_weight=score_weight,
        max_iter=max_iter, proj_iter=proj_iter, lr=lr, is_test=is_test)
    return full[0], ms[0], ds[0], loss[0]


class _CosineFn(torch.autograd.Function):
    """get_cosine_score (match_helper.py:51-64) for B frames with autograd: tf [B,O,D], pf [B,P,D] -> cos [B,O,P].
    Used where the cosine table is needed OUTSIDE the fused layer function (algo 'hun': the assignment comes from
    scipy and carries no gradient, but mse(cos, gt) still trains the features like in the reference)."""

    @staticmethod
    def forward(ctx, tf, pf):
        pn, pnorm = ops.feature_normalize(pf, want_norms=True)
        tn, tnorm = ops.feature_normalize(tf, want_norms=True)
        ctx.save_for_backward(pn, tn, pnorm, tnorm, pf, tf)
        return ops.cosine(tn, pn)

    @staticmethod
    def backward(ctx, dcos):
        pn, tn, pnorm, tnorm, pf, tf = ctx.saved_tensors
        # dmm_feature_sim_bwd_f32 with weight (1 - 0) = 1 and no loss term: both contractions + normalisation backward
        g_tf, g_pf = ops.feature_sim_bwd(dcos, None, None, None, 0.0, tf, pf, tn, pn, tnorm, pnorm)
        return (g_tf if ctx.needs_input_grad[0] else None), (g_pf if ctx.needs_input_grad[1] else None)


def _hungarian_forward(pf, tf, pm, tm, sc, targets, score_weight, is_test):
    """algo 'hun' slot (match_model.py:122-123): same cost matrix, assignment from scipy (host)."""
    P, O = pm.shape[0], tm.shape[0]
    pm_b, tm_b = pm.unsqueeze(0), tm.unsqueeze(0)
    inter, ap, at = ops.iou_counts(pm_b, tm_b)
    cos = _CosineFn.apply(tf.unsqueeze(0), pf.unsqueeze(0))          # differentiable: cost_loss reaches the features
    r = ops.relax_match(cos.detach(), inter, ap, at, sc.unsqueeze(0), score_weight=score_weight, max_iter=0,
                        proj_iter=0, lr=0.0, is_test=is_test)
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
        with torch.no_grad():                                        # gt one-hot carries no grad (match_helper.py:34-44)
            gt = matching_loss(pm_b, targets.unsqueeze(0).float(), cos.detach())[1]
        loss = ((cos - gt) ** 2).mean()                              # F.mse_loss(feature_sim, gt_matched) (:48)
    return full, ms, ds, loss
