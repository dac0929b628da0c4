"""Backward of the matching layer (counterpart of torch autograd through the reference's
``MatchModel.forward``; reference files: match_model.py:49-148, match_helper.py:30-64,
relax_match.py:36-105).  B frames per call.

Chain (tensors on the MI355X):

  d full_outmask [B,O,H,W]  --dmm_mask_mix_bwd----->  dRb [B,O,Pp]    (HIP, HBM bound: selected planes only)
  dRb, d match_score, d det_score  --dmm_relax_match_bwd_f32-->  dsim [B,O,P]   (HIP: taped reverse sweep)
  dsim * (1-w) + d cost_loss * 2 (cos - gt) / (O P)  =  dcos [B,O,P]          \
  dcos  -->  d template_n = dcos @ pn,  d proposal_n = dcos^T @ tn             >  dmm_feature_sim_bwd_f32 (HIP, one launch)
  normalisation backward  -->  d template_feature, d proposed_feature         /

Masks, scores and targets receive no gradient in the reference's use (they come from the frozen
proposal network); if ``proposed_mask`` requires grad its gradient Rb^T @ dOut is produced too.
"""
from __future__ import annotations

import torch

from . import ops


def _normalize_backward(g_hat: torch.Tensor, c: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """x_hat = x / c with c = max(||x||, eps), where torch clamps the VALUE in place under no_grad
    (cosine_similarity): autograd differentiates c as ||x||.  g_hat, x: [..., D]; c: [...]."""
    c = c.unsqueeze(-1)
    n = x.norm(dim=-1, keepdim=True)
    dot = (g_hat * x).sum(-1, keepdim=True)
    corr = torch.where(n > 0, x * dot / (c * c * n.clamp_min(1e-30)), torch.zeros_like(x))
    return g_hat / c - corr


def match_layer_backward(ctx, d_full, d_ms, d_ds, d_loss):
    (pn, tn, pnorm, tnorm, pf, tf, cos, sim, Rb, sc, pm, gt, live, cnt, n_valid, m_valid) = ctx.saved_tensors
    n_valid = n_valid if ctx.ragged[0] else None
    m_valid = m_valid if ctx.ragged[1] else None
    score_weight, max_iter, proj_iter, lr, is_test = ctx.cfg
    if ctx.frame_planes is not None:                 # per-frame tensors: the HIP kernels take the pointer table
        pm = ctx.frame_planes
        B, P, H, W = pm.B, pm.N, pm.H, pm.W
    else:
        B, P = pm.shape[0], pm.shape[1]
        H, W = pm.shape[-2], pm.shape[-1]
    O = sim.shape[1]
    Pp = Rb.shape[-1]
    need_pf, need_tf, need_pm = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2]
    g_pf = g_tf = g_pm = None
    none = (None,) * 11
    if not (need_pf or need_tf or need_pm):
        return (None, None, None) + none
    dOut = None if d_full is None else d_full.reshape(B, O, H * W).float()
    if need_pm and dOut is not None:                 # (never with frame planes: they carry no gradient by construction)
        g_pm = torch.bmm(Rb[:, :, :P].transpose(1, 2), dOut).view(B, P, H, W).to(pm.dtype)
    if need_pf or need_tf:
        dRb = None
        if dOut is not None:
            # HIP: only the planes selected by Rb are streamed (padded columns multiply zero planes -> stay 0)
            dRb = ops.mask_mix_bwd(Rb, pm, dOut, n_valid, m_valid)
        dsim = ops.relax_match_bwd(sim, sc, dRb, d_ms, d_ds, max_iter=max_iter, proj_iter=proj_iter, lr=lr,
                                   is_test=is_test, n_valid=n_valid, m_valid=m_valid)
        T = tn.shape[0]                                          # template-feature entries; cos = mean_t cos_t
        has_loss = ctx.has_targets and d_loss is not None
        if T == 1:
            # one HIP launch: (1 - w) mix + matching-loss term + both contractions + the normalisation backward
            g_t, g_p = ops.feature_sim_bwd(dsim, cos if has_loss else None, gt if has_loss else None,
                                           d_loss if has_loss else None, score_weight, tf[0], pf, tn[0], pn, tnorm[0],
                                           pnorm, n_valid, m_valid)
            g_pf = g_p if need_pf else None
            g_tf = g_t.unsqueeze(0) if need_tf else None
        else:
            # several template-feature entries (never in DMM-Net itself, dmm_model.py:44): plain torch ops
            w_feat = torch.tensor(1.0 - score_weight, dtype=torch.float32).item()
            dcos = dsim * w_feat
            if has_loss:
                diff = cos - gt
                if ctx.ragged[0] or ctx.ragged[1]:
                    dcos = dcos + torch.where(live, diff, torch.zeros_like(diff)) * (2.0 * d_loss / cnt)[:, None, None]
                else:
                    dcos = dcos + diff * (2.0 / (O * P)) * d_loss[:, None, None]
            dcos_t = dcos / T
            g_pn = torch.bmm(dcos_t.transpose(1, 2), tn[0])      # [B,P,D]
            for t in range(1, T):
                g_pn = g_pn + torch.bmm(dcos_t.transpose(1, 2), tn[t])
            if need_pf:
                g_pf = _normalize_backward(g_pn, pnorm, pf)
            if need_tf:
                g_tn = torch.stack([torch.bmm(dcos_t, pn) for _ in range(T)], 0)      # [T,B,O,D]
                g_tf = _normalize_backward(g_tn, tnorm, tf)
    return (g_pf, g_tf, g_pm) + none
