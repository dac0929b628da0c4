"""Batched device ops of the matching layer: thin tensor-level wrappers over the C ABI.

Every function takes CUDA(=HIP) tensors, enqueues on the current torch stream of their device and
returns fresh tensors.  B frames per call; ``n_valid`` / ``m_valid`` (int32 [B], optional) make the
batch ragged.  No function here has a CPU implementation: CPU tensors raise.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import _lib

_DT = {torch.float32: _lib.DTYPE_F32, torch.float16: _lib.DTYPE_F16, torch.bfloat16: _lib.DTYPE_BF16}


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _stream(t: torch.Tensor):
    return torch.cuda.current_stream(t.device).cuda_stream


def _need_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.DmmError("dmm_net_amd ops need tensors on an MI355X device (no CPU fallback)")


def _planes(t: torch.Tensor) -> Tuple[torch.Tensor, int, int]:
    """[B,K,H,W] with contiguous H*W planes -> (tensor, frame stride, plane stride) in elements."""
    assert t.dim() == 4, t.shape
    H, W = t.shape[2], t.shape[3]
    if t.shape[0] * t.shape[1] * H * W and not (t.stride(3) == 1 and t.stride(2) == W and t.stride(1) >= H * W):
        t = t.contiguous()
    return t, t.stride(0), t.stride(1)


def padded_width(N: int, M: int) -> int:
    """Pp = solver width after the reference's zero padding (match_model.py:109-113)."""
    return N if N > M else M + 1


def iou_counts(masks_p: torch.Tensor, masks_t: torch.Tensor, n_valid=None, m_valid=None):
    """-> inter [B,M,N] i32, area_p [B,N] i32, area_t [B,M] i32 (match_helper.py:9-28 on all pairs)."""
    _need_gpu(masks_p, masks_t)
    assert masks_p.dtype == masks_t.dtype and masks_p.dtype in _DT
    masks_p, sp_b, sp_n = _planes(masks_p)
    masks_t, st_b, st_m = _planes(masks_t)
    B, N, H, W = masks_p.shape
    M = masks_t.shape[1]
    assert masks_t.shape[0] == B and masks_t.shape[2:] == masks_p.shape[2:]
    dev = masks_p.device
    inter = torch.empty((B, M, N), dtype=torch.int32, device=dev)
    ap = torch.empty((B, N), dtype=torch.int32, device=dev)
    at = torch.empty((B, M), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        rc = _lib.load().dmm_iou_counts(_ptr(masks_p), _ptr(masks_t), _DT[masks_p.dtype], B, N, M, H * W, sp_b, sp_n,
                                        st_b, st_m, _ptr(n_valid), _ptr(m_valid), _ptr(inter), _ptr(ap), _ptr(at),
                                        _stream(masks_p))
    _lib.check(rc, "dmm_iou_counts")
    return inter, ap, at


def feature_normalize(x: torch.Tensor, want_norms: bool = False):
    """x [..., D] fp32 -> x / max(||x||, 1e-8) (and the clamped norms)."""
    _need_gpu(x)
    x = x.contiguous().float()
    D = x.shape[-1]
    rows = x.numel() // max(D, 1)
    out = torch.empty_like(x)
    norms = torch.empty(x.shape[:-1], dtype=torch.float32, device=x.device) if want_norms else None
    with torch.cuda.device(x.device):
        rc = _lib.load().dmm_feature_normalize_f32(_ptr(x), rows, D, _ptr(out), _ptr(norms), _stream(x))
    _lib.check(rc, "dmm_feature_normalize_f32")
    return (out, norms) if want_norms else out


def cosine(featn_t: torch.Tensor, featn_p: torch.Tensor, n_valid=None, m_valid=None) -> torch.Tensor:
    """cos [B,M,N] of normalised rows featn_t [B,M,D], featn_p [B,N,D] (match_helper.py:59-63)."""
    _need_gpu(featn_t, featn_p)
    featn_t, featn_p = featn_t.contiguous().float(), featn_p.contiguous().float()
    B, M, D = featn_t.shape
    N = featn_p.shape[1]
    out = torch.empty((B, M, N), dtype=torch.float32, device=featn_t.device)
    with torch.cuda.device(featn_t.device):
        rc = _lib.load().dmm_cosine_f32(_ptr(featn_t), _ptr(featn_p), B, N, M, D, _ptr(n_valid), _ptr(m_valid),
                                        _ptr(out), _stream(featn_t))
    _lib.check(rc, "dmm_cosine_f32")
    return out


def relax_match(cos, inter, area_p, area_t, score_p, *, score_weight, max_iter, proj_iter, lr, is_test,
                n_valid=None, m_valid=None, want_x=False):
    """Similarity mix + relaxed assignment + scores for B frames.  cos [B,M,N] = feature_sim.
    Returns dict(sim, R, Rb, match_score, det_score, iters, X)."""
    _need_gpu(cos, inter)
    B, M, N = cos.shape
    Pp = padded_width(N, M)
    dev = cos.device
    f32 = dict(dtype=torch.float32, device=dev)
    out = dict(sim=torch.empty((B, M, N), **f32), R=torch.empty((B, M, Pp), **f32), Rb=torch.empty((B, M, Pp), **f32),
               match_score=torch.empty((B, M), **f32), det_score=torch.empty((B, M), **f32),
               iters=torch.empty((B,), dtype=torch.int32, device=dev),
               X=torch.empty((B, M, Pp), **f32) if want_x else None)
    cos, score_p = cos.contiguous().float(), score_p.contiguous().float()
    with torch.cuda.device(dev):
        rc = _lib.load().dmm_relax_match_f32(
            _ptr(cos), _ptr(inter), _ptr(area_p), _ptr(area_t), _ptr(score_p), B, N, M, _ptr(n_valid), _ptr(m_valid),
            float(score_weight), int(max_iter), int(proj_iter), float(lr), int(is_test), _ptr(out["sim"]),
            _ptr(out["R"]), _ptr(out["Rb"]), _ptr(out["match_score"]), _ptr(out["det_score"]), _ptr(out["iters"]),
            _ptr(out["X"]), _stream(cos))
    _lib.check(rc, "dmm_relax_match_f32")
    return out


def relax_solve(C: torch.Tensor, max_iter: int, proj_iter: int, lr: float):
    """relax_matching on cost matrices C [B,n,m] -> dict(X, R, cost [B,max_iter+1], iters [B])."""
    _need_gpu(C)
    C = C.contiguous().float()
    B, n, m = C.shape
    X = torch.empty_like(C)
    R = torch.empty_like(C)
    cost = torch.zeros((B, max_iter + 1), dtype=torch.float32, device=C.device)
    iters = torch.empty((B,), dtype=torch.int32, device=C.device)
    with torch.cuda.device(C.device):
        rc = _lib.load().dmm_relax_solve_f32(_ptr(C), B, n, m, int(max_iter), int(proj_iter), float(lr), _ptr(X),
                                             _ptr(R), _ptr(cost), _ptr(iters), _stream(C))
    _lib.check(rc, "dmm_relax_solve_f32")
    return dict(X=X, R=R, cost=cost, iters=iters)


def mask_mix(Rb: torch.Tensor, masks_p: torch.Tensor, n_valid=None, m_valid=None) -> torch.Tensor:
    """full_outmask [B,M,H,W] = Rb [B,M,Pp] @ masks_p [B,N,H*W] (zero planes for the padded columns)."""
    _need_gpu(Rb, masks_p)
    masks_p, sp_b, sp_n = _planes(masks_p)
    B, N, H, W = masks_p.shape
    M, Pp = Rb.shape[1], Rb.shape[2]
    Rb = Rb.contiguous().float()
    out = torch.empty((B, M, H, W), dtype=torch.float32, device=Rb.device)
    with torch.cuda.device(Rb.device):
        rc = _lib.load().dmm_mask_mix(_ptr(Rb), _ptr(masks_p), _DT[masks_p.dtype], B, N, M, Pp, H * W, sp_b, sp_n,
                                      _ptr(n_valid), _ptr(m_valid), _ptr(out), M * H * W, H * W, _stream(Rb))
    _lib.check(rc, "dmm_mask_mix")
    return out


class ForwardPlan:
    """Pre-allocated outputs + workspace for the fused forward of B same-shaped frames
    (``dmm_match_forward``): one ctypes call per batch, nothing allocated in the timed region."""

    def __init__(self, B, N, M, H, W, D, device, mask_dtype=torch.float32, want_tables=False):
        self.B, self.N, self.M, self.H, self.W, self.D = B, N, M, H, W, D
        self.Pp = padded_width(N, M)
        self.device = torch.device(device)
        self.mask_dtype = mask_dtype
        L = _lib.load()
        f32 = dict(dtype=torch.float32, device=self.device)
        self.ws_bytes = int(L.dmm_workspace_bytes(B, N, M, D))
        self.workspace = torch.empty((self.ws_bytes,), dtype=torch.uint8, device=self.device)
        self.full_outmask = torch.empty((B, M, H, W), **f32)
        self.match_score = torch.empty((B, M), **f32)
        self.det_score = torch.empty((B, M), **f32)
        self.iters = torch.empty((B,), dtype=torch.int32, device=self.device)
        self.sim = torch.empty((B, M, N), **f32) if want_tables else None
        self.R = torch.empty((B, M, self.Pp), **f32) if want_tables else None
        self.Rb = torch.empty((B, M, self.Pp), **f32) if want_tables else None

    def run(self, masks_p, masks_t, feat_p, feat_t, score_p, *, score_weight=0.3, max_iter=20, proj_iter=5, lr=0.1,
            is_test=1, n_valid=None, m_valid=None):
        _need_gpu(masks_p, masks_t, feat_p, feat_t, score_p)
        masks_p, sp_b, sp_n = _planes(masks_p)
        masks_t, st_b, st_m = _planes(masks_t)
        assert masks_p.shape == (self.B, self.N, self.H, self.W) and masks_t.shape == (self.B, self.M, self.H, self.W)
        assert masks_p.dtype == self.mask_dtype and masks_t.dtype == self.mask_dtype
        assert feat_p.is_contiguous() and feat_t.is_contiguous() and score_p.is_contiguous()
        assert feat_p.dtype == torch.float32 and feat_t.dtype == torch.float32 and score_p.dtype == torch.float32
        with torch.cuda.device(self.device):
            rc = _lib.load().dmm_match_forward(
                _ptr(masks_p), _ptr(masks_t), _DT[self.mask_dtype], _ptr(feat_p), _ptr(feat_t), _ptr(score_p), self.B,
                self.N, self.M, self.H * self.W, self.D, sp_b, sp_n, st_b, st_m, _ptr(n_valid), _ptr(m_valid),
                float(score_weight), int(max_iter), int(proj_iter), float(lr), int(is_test), _ptr(self.full_outmask),
                _ptr(self.match_score), _ptr(self.det_score), _ptr(self.sim), _ptr(self.R), _ptr(self.Rb),
                _ptr(self.iters), _ptr(self.workspace), self.ws_bytes, _stream(masks_p))
        _lib.check(rc, "dmm_match_forward")
        return self.full_outmask, self.match_score, self.det_score
