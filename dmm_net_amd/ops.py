"""Batched device ops of the matching layer: thin tensor-level wrappers over the C ABI.

Every function takes CUDA(=HIP) tensors, enqueues on the current torch stream of their device and
returns fresh tensors.  B frames per call; ``n_valid`` / ``m_valid`` (int32 [B], optional) make the
batch ragged.  No function here has a CPU implementation: CPU tensors raise.
"""
from __future__ import annotations

import ctypes
import threading
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


def alloc_planes(B: int, K: int, H: int, W: int, dtype, device, align_bytes: int = 128, fill=None) -> torch.Tensor:
    """[B,K,H,W] mask planes whose PLANE STRIDE is rounded up to ``align_bytes`` (every plane then starts on a 128-byte
    line).  255 x 255 planes are an odd number of elements: contiguous 16-bit planes alternate between dword-aligned and
    2-bytes-off starts, and every 2 KiB run of a misaligned plane straddles one extra 128-byte line (+5.4 % HBM traffic in
    the count kernel at BASELINE configs[4], profiles/r03).  The C ABI takes the plane stride (``sp_n`` / ``so_m``), so a
    producer that owns its buffers just allocates them this way; H*W elements of a plane stay contiguous."""
    es = torch.empty((), dtype=dtype).element_size()
    q = max(1, align_bytes // es)
    S = (H * W + q - 1) // q * q
    buf = torch.empty((B, K, S), dtype=dtype, device=device)
    if fill is not None:
        buf.fill_(fill)
    return torch.as_strided(buf, (B, K, H, W), (K * S, S, W, 1))


class FramePlanes:
    """Proposal planes of B frames held as ONE TENSOR PER FRAME (the reference's ``prop_m[bid]``, dmm_model.py:58 / :111)
    plus the device pointer table the ``*_frames`` entry points of the C ABI take -- the frames go through one launch
    without being copied into a [B, Nmax, H, W] batch first.  Frame b has ``planes[b].shape[0]`` planes (ragged: pass
    ``n_valid``); all frames share H, W, dtype and the plane stride."""

    def __init__(self, planes, table="build"):
        """``table``: "build" uploads the pointer table here; None defers it -- the caller uploads ``addresses()`` together
        with its other small tables (``_lib.small_to_device_many``) and assigns ``.table``."""
        planes = list(planes)
        assert len(planes) > 0
        H, W = int(planes[0].shape[-2]), int(planes[0].shape[-1])
        self.dtype, self.device = planes[0].dtype, planes[0].device
        assert self.dtype in _DT, self.dtype
        fixed = []
        for t in planes:
            _need_gpu(t)
            assert t.dim() == 3 and tuple(t.shape[-2:]) == (H, W) and t.dtype == self.dtype and t.device == self.device
            if t.shape[0] and not (t.stride(2) == 1 and t.stride(1) == W and t.stride(0) >= H * W):
                t = t.contiguous()
            fixed.append(t)
        strides = {int(t.stride(0)) for t in fixed if t.shape[0] > 1}
        if len(strides) > 1:                         # mixed plane strides: densify the odd ones
            fixed = [t if (t.shape[0] <= 1 or t.stride(0) == H * W) else t.contiguous() for t in fixed]
            strides = {int(t.stride(0)) for t in fixed if t.shape[0] > 1}
            if len(strides) > 1:
                fixed = [t.contiguous() for t in fixed]
                strides = {H * W}
        self.planes = fixed                          # keeps the storage alive
        self.plane_stride = strides.pop() if strides else H * W
        self.H, self.W, self.B = H, W, len(fixed)
        self.N = max(int(t.shape[0]) for t in fixed)
        self.counts = [int(t.shape[0]) for t in fixed]
        # one small H2D copy; frames without planes get a valid dummy address (never dereferenced: n_valid = 0)
        self.table = _lib.small_to_device(self.addresses(), torch.int64, self.device) if table == "build" else table

    def addresses(self):
        any_ptr = next((t.data_ptr() for t in self.planes if t.shape[0]), 0)
        return [t.data_ptr() if t.shape[0] else any_ptr for t in self.planes]

    def n_valid(self) -> torch.Tensor:
        return _lib.small_to_device(self.counts, torch.int32, self.device)

    def stacked(self) -> torch.Tensor:
        """[B, N, H, W] copy (zero planes beyond a frame's count) -- only for paths without a pointer-table kernel."""
        out = self.planes[0].new_zeros((self.B, self.N, self.H, self.W))
        for b, t in enumerate(self.planes):
            out[b, :t.shape[0]] = t
        return out


def padded_width(N: int, M: int) -> int:
    """Pp = solver width after the reference's zero padding (match_model.py:109-113)."""
    return N if N > M else M + 1


def iou_counts(masks_p: torch.Tensor, masks_t: torch.Tensor, n_valid=None, m_valid=None):
    """-> inter [B,M,N] i32, area_p [B,N] i32, area_t [B,M] i32 (match_helper.py:9-28 on all pairs)."""
    if isinstance(masks_p, FramePlanes):
        fp = masks_p
        _need_gpu(masks_t)
        assert fp.dtype == masks_t.dtype and n_valid is not None
        masks_t, st_b, st_m = _planes(masks_t)
        B, N, H, W, M = fp.B, fp.N, fp.H, fp.W, masks_t.shape[1]
        assert masks_t.shape[0] == B and tuple(masks_t.shape[2:]) == (H, W)
        dev = fp.device
        inter = torch.empty((B, M, N), dtype=torch.int32, device=dev)
        ap = torch.empty((B, N), dtype=torch.int32, device=dev)
        at = torch.empty((B, M), dtype=torch.int32, device=dev)
        with _lib.device_guard(dev):
            rc = _lib.load().dmm_iou_counts_frames(_ptr(fp.table), _ptr(masks_t), _DT[fp.dtype], B, N, M, H * W,
                                                   fp.plane_stride, st_b, st_m, _ptr(n_valid), _ptr(m_valid),
                                                   _ptr(inter), _ptr(ap), _ptr(at), _stream(masks_t))
        _lib.check(rc, "dmm_iou_counts_frames")
        return inter, ap, at
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
    with _lib.device_guard(dev):
        rc = _lib.load().dmm_iou_counts(_ptr(masks_p), _ptr(masks_t), _DT[masks_p.dtype], B, N, M, H * W, sp_b, sp_n,
                                        st_b, st_m, _ptr(n_valid), _ptr(m_valid), _ptr(inter), _ptr(ap), _ptr(at),
                                        _stream(masks_p))
    _lib.check(rc, "dmm_iou_counts")
    return inter, ap, at


def pack_words(HW: int) -> int:
    """uint64 words per packed plane of HW pixels (DMM_PACKED1 ballot layout)."""
    return 4 * ((HW + 255) // 256)


def ragged_blocks(blocks):
    """-> (contiguous blocks, their addresses as the pointer table of ``ragged_pad`` wants them)."""
    blocks = [b.contiguous() for b in blocks]
    any_ptr = next((b.data_ptr() for b in blocks if b.shape[0]), 0)
    return blocks, [b.data_ptr() if b.shape[0] else any_ptr for b in blocks]


def ragged_pad(blocks, P_max: int, counts: torch.Tensor, table: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Per-video blocks [P_b, ...] (same trailing shape and dtype, 4-byte multiples per row) -> [B, P_max, ...] with
    zeros from row P_b on, in ONE launch (``dmm_ragged_pad``): the batching step of the per-video driver.
    counts: [B] int32 on the device (= the n_valid the ragged kernels take anyway).  ``table`` (optional): the blocks'
    addresses already on the device (``ragged_blocks`` + ``_lib.small_to_device_many``; the blocks must then be the
    contiguous ones ``ragged_blocks`` returned)."""
    if table is None:
        blocks, addrs = ragged_blocks(blocks)
    first = blocks[0]
    _need_gpu(first, counts)
    tail = tuple(first.shape[1:])
    row_bytes = first.element_size()
    for d in tail:
        row_bytes *= int(d)
    out = torch.empty((len(blocks), int(P_max)) + tail, dtype=first.dtype, device=first.device)
    if row_bytes % 4 != 0 or any(tuple(b.shape[1:]) != tail or b.dtype != first.dtype for b in blocks):
        raise ValueError("ragged_pad: blocks must share dtype and trailing shape, rows of a multiple of 4 bytes")
    if table is None:
        table = _lib.small_to_device(addrs, torch.int64, first.device)
    with _lib.device_guard(first.device):
        rc = _lib.load().dmm_ragged_pad(_ptr(table), _ptr(counts), len(blocks), int(P_max), row_bytes, _ptr(out),
                                        _stream(first))
    _lib.check(rc, "dmm_ragged_pad")
    return out                                        # (the blocks are read in stream order: no keep-alive needed)


def pack_masks(masks: torch.Tensor) -> torch.Tensor:
    """[B,K,H,W] soft masks -> [B,K,words] int64 bit planes of (x > 0.5) in the library's ballot layout."""
    _need_gpu(masks)
    masks, s_b, s_k = _planes(masks)
    B, K, H, W = masks.shape
    if s_b != K * s_k:
        masks = masks.contiguous()
        s_k = H * W
    wd = pack_words(H * W)
    out = torch.empty((B, K, wd), dtype=torch.int64, device=masks.device)
    with _lib.device_guard(masks.device):
        rc = _lib.load().dmm_pack_masks(_ptr(masks), _DT[masks.dtype], B * K, H * W, s_k, _ptr(out), wd, _stream(masks))
    _lib.check(rc, "dmm_pack_masks")
    return out


def iou_counts_packed(packed_p: torch.Tensor, packed_t: torch.Tensor, HW: int, n_valid=None, m_valid=None):
    """iou_counts on DMM_PACKED1 planes ([B,N,words], [B,M,words] int64): identical integer tables, 1/32 of the bytes."""
    _need_gpu(packed_p, packed_t)
    assert packed_p.dtype == torch.int64 and packed_t.dtype == torch.int64
    packed_p, packed_t = packed_p.contiguous(), packed_t.contiguous()
    B, N, wd = packed_p.shape
    M = packed_t.shape[1]
    assert wd == pack_words(HW) and packed_t.shape[2] == wd
    dev = packed_p.device
    inter = torch.empty((B, M, N), dtype=torch.int32, device=dev)
    ap = torch.empty((B, N), dtype=torch.int32, device=dev)
    at = torch.empty((B, M), dtype=torch.int32, device=dev)
    with _lib.device_guard(dev):
        rc = _lib.load().dmm_iou_counts(_ptr(packed_p), _ptr(packed_t), _lib.DTYPE_PACKED1, B, N, M, HW, N * wd, wd,
                                        M * wd, wd, _ptr(n_valid), _ptr(m_valid), _ptr(inter), _ptr(ap), _ptr(at),
                                        _stream(packed_p))
    _lib.check(rc, "dmm_iou_counts (packed)")
    return inter, ap, at


def iou_counts_dual(masks_p: torch.Tensor, masks_t: torch.Tensor, masks_t2: torch.Tensor, n_valid=None, m_valid=None):
    """One pass over the proposal planes against TWO template sets (templates + training targets).
    -> (inter, area_p, area_t), (inter2, area_t2)."""
    if isinstance(masks_p, FramePlanes):
        fp = masks_p
        _need_gpu(masks_t, masks_t2)
        assert fp.dtype == masks_t.dtype == masks_t2.dtype and masks_t.shape == masks_t2.shape and n_valid is not None
        masks_t, st_b, st_m = _planes(masks_t)
        masks_t2, st2_b, st2_m = _planes(masks_t2)
        B, N, H, W, M = fp.B, fp.N, fp.H, fp.W, masks_t.shape[1]
        i32 = dict(dtype=torch.int32, device=fp.device)
        inter, inter2 = torch.empty((B, M, N), **i32), torch.empty((B, M, N), **i32)
        ap, at, at2 = torch.empty((B, N), **i32), torch.empty((B, M), **i32), torch.empty((B, M), **i32)
        with _lib.device_guard(fp.device):
            rc = _lib.load().dmm_iou_counts_dual_frames(_ptr(fp.table), _ptr(masks_t), _ptr(masks_t2), _DT[fp.dtype], B,
                                                        N, M, H * W, fp.plane_stride, st_b, st_m, st2_b, st2_m,
                                                        _ptr(n_valid), _ptr(m_valid), _ptr(inter), _ptr(ap), _ptr(at),
                                                        _ptr(inter2), _ptr(at2), _stream(masks_t))
        _lib.check(rc, "dmm_iou_counts_dual_frames")
        return (inter, ap, at), (inter2, at2)
    _need_gpu(masks_p, masks_t, masks_t2)
    assert masks_p.dtype == masks_t.dtype == masks_t2.dtype and masks_p.dtype in _DT
    assert masks_t.shape == masks_t2.shape
    masks_p, sp_b, sp_n = _planes(masks_p)
    masks_t, st_b, st_m = _planes(masks_t)
    masks_t2, st2_b, st2_m = _planes(masks_t2)
    B, N, H, W = masks_p.shape
    M = masks_t.shape[1]
    dev = masks_p.device
    i32 = dict(dtype=torch.int32, device=dev)
    inter, inter2 = torch.empty((B, M, N), **i32), torch.empty((B, M, N), **i32)
    ap, at, at2 = torch.empty((B, N), **i32), torch.empty((B, M), **i32), torch.empty((B, M), **i32)
    with _lib.device_guard(dev):
        rc = _lib.load().dmm_iou_counts_dual(_ptr(masks_p), _ptr(masks_t), _ptr(masks_t2), _DT[masks_p.dtype], B, N, M,
                                             H * W, sp_b, sp_n, st_b, st_m, st2_b, st2_m, _ptr(n_valid), _ptr(m_valid),
                                             _ptr(inter), _ptr(ap), _ptr(at), _ptr(inter2), _ptr(at2), _stream(masks_p))
    _lib.check(rc, "dmm_iou_counts_dual")
    return (inter, ap, at), (inter2, at2)


def feature_normalize(x: torch.Tensor, want_norms: bool = False):
    """x [..., D] fp32 -> x / max(||x||, 1e-8) (and the clamped norms)."""
    _need_gpu(x)
    x = x.contiguous().float()
    D = x.shape[-1]
    rows = x.numel() // max(D, 1)
    out = torch.empty_like(x)
    norms = torch.empty(x.shape[:-1], dtype=torch.float32, device=x.device) if want_norms else None
    with _lib.device_guard(x.device):
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
    with _lib.device_guard(featn_t.device):
        rc = _lib.load().dmm_cosine_f32(_ptr(featn_t), _ptr(featn_p), B, N, M, D, _ptr(n_valid), _ptr(m_valid),
                                        _ptr(out), _stream(featn_t))
    _lib.check(rc, "dmm_cosine_f32")
    return out


def cosine_features(feat_t: torch.Tensor, feat_p: torch.Tensor) -> torch.Tensor:
    """get_cosine_score (match_helper.py:51-64) from RAW features feat_t [B,M,D], feat_p [B,N,D] -> cos [B,M,N]: one
    fused launch inside its envelope (dense, N >= 2, D % 64 == 0; D = 256 / 512 / 1024 with the D axis over the lanes),
    else normalise x 2 + cosine.  Bit identical either way."""
    _need_gpu(feat_t, feat_p)
    feat_t, feat_p = feat_t.contiguous().float(), feat_p.contiguous().float()
    B, M, D = feat_t.shape
    N = feat_p.shape[1]
    out = torch.empty((B, M, N), dtype=torch.float32, device=feat_t.device)
    with _lib.device_guard(feat_t.device):
        rc = _lib.load().dmm_cosine_features_f32(_ptr(feat_t), _ptr(feat_p), B, N, M, D, _ptr(out), _stream(feat_t))
    if rc == 2:                                               # DMM_ERR_UNSUPPORTED: outside the fused kernel's envelope
        return cosine(feature_normalize(feat_t), feature_normalize(feat_p))
    _lib.check(rc, "dmm_cosine_features_f32")
    return out


def relax_match(cos, inter, area_p, area_t, score_p, *, score_weight, max_iter, proj_iter, lr, is_test,
                n_valid=None, m_valid=None, want_x=False, state="f32"):
    """Similarity mix + relaxed assignment + scores for B frames.  cos [B,M,N] = feature_sim.
    Returns dict(sim, R, Rb, match_score, det_score, iters, X).  ``state="f16"``: the opt-in tolerance mode with the
    solver state in packed fp16 and fp32 sums (``dmm_relax_match_f16s``, BASELINE configs[4]); the default reproduces
    the reference bit for bit.  Tables outside the fast kernels' envelope (M > 32 or Pp > 256) go through
    ``dmm_relax_match_any_f32`` (general solver, state in a scratch tensor allocated here) -- fp32 state only."""
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
    assert state in ("f32", "f16")
    L = _lib.load()
    fn = L.dmm_relax_match_f32 if state == "f32" else L.dmm_relax_match_f16s
    args = (_ptr(cos), _ptr(inter), _ptr(area_p), _ptr(area_t), _ptr(score_p), B, N, M, _ptr(n_valid), _ptr(m_valid),
            float(score_weight), int(max_iter), int(proj_iter), float(lr), int(is_test), _ptr(out["sim"]),
            _ptr(out["R"]), _ptr(out["Rb"]), _ptr(out["match_score"]), _ptr(out["det_score"]), _ptr(out["iters"]),
            _ptr(out["X"]))
    with _lib.device_guard(dev):
        if state == "f32" and (M > _lib.MAX_TEMPLATES or Pp > _lib.MAX_PROPOSALS):
            scratch = torch.empty((int(L.dmm_relax_any_scratch_bytes(B, N, M)),), dtype=torch.uint8, device=dev)
            rc = L.dmm_relax_match_any_f32(*args, _ptr(scratch), scratch.numel(), _stream(cos))
            scratch.record_stream(torch.cuda.current_stream(dev))
        else:
            rc = fn(*args, _stream(cos))
    _lib.check(rc, "dmm_relax_match_" + ("f32" if state == "f32" else "f16s"))
    return out


def relax_match_bwd(sim, score_p, dRb, d_match_score, d_det_score, *, max_iter, proj_iter, lr, is_test,
                    n_valid=None, m_valid=None) -> torch.Tensor:
    """d loss / d sim [B,M,N] of ``relax_match`` (reverse sweep through the taped solver iterations)."""
    _need_gpu(sim)
    sim = sim.contiguous().float()
    B, M, N = sim.shape
    dev = sim.device
    cf = lambda t: None if t is None else t.contiguous().float()
    dRb, d_match_score, d_det_score, score_p = cf(dRb), cf(d_match_score), cf(d_det_score), cf(score_p)
    L = _lib.load()
    nbytes = int(L.dmm_relax_bwd_workspace_bytes(B, N, M, int(max_iter), int(proj_iter)))
    ws = torch.empty((max(nbytes, 8),), dtype=torch.uint8, device=dev)
    out = torch.empty((B, M, N), dtype=torch.float32, device=dev)
    with _lib.device_guard(dev):
        rc = L.dmm_relax_match_bwd_f32(_ptr(sim), _ptr(score_p), B, N, M, _ptr(n_valid), _ptr(m_valid), int(max_iter),
                                       int(proj_iter), float(lr), int(is_test), _ptr(dRb), _ptr(d_match_score),
                                       _ptr(d_det_score), _ptr(out), _ptr(ws), ws.numel(), _stream(sim))
    _lib.check(rc, "dmm_relax_match_bwd_f32")
    return out


def feature_sim_bwd(dsim, cos, gt, d_loss, score_weight, feat_t, feat_p, featn_t, featn_p, norm_t, norm_p,
                    n_valid=None, m_valid=None):
    """-> (g_feat_t [B,M,D], g_feat_p [B,N,D]): backward of cosine + (1 - w) mix + matching-loss mse in one launch
    (``dmm_feature_sim_bwd_f32``).  gt / d_loss None = no matching loss."""
    _need_gpu(dsim, feat_t, feat_p)
    cf = lambda t: None if t is None else t.contiguous().float()
    dsim, cos, gt, d_loss = cf(dsim), cf(cos), cf(gt), cf(d_loss)
    feat_t, feat_p, featn_t, featn_p = cf(feat_t), cf(feat_p), cf(featn_t), cf(featn_p)
    norm_t, norm_p = cf(norm_t), cf(norm_p)
    B, M, N = dsim.shape
    D = feat_p.shape[-1]
    g_t, g_p = torch.empty_like(feat_t), torch.empty_like(feat_p)
    if gt is None or d_loss is None:
        gt = d_loss = cos_arg = None
    else:
        cos_arg = cos
    with _lib.device_guard(dsim.device):
        rc = _lib.load().dmm_feature_sim_bwd_f32(_ptr(dsim), _ptr(cos_arg), _ptr(gt), _ptr(d_loss), float(score_weight),
                                                 _ptr(feat_t), _ptr(feat_p), _ptr(featn_t), _ptr(featn_p), _ptr(norm_t),
                                                 _ptr(norm_p), B, N, M, D, _ptr(n_valid), _ptr(m_valid), _ptr(g_t),
                                                 _ptr(g_p), _stream(dsim))
    _lib.check(rc, "dmm_feature_sim_bwd_f32")
    return g_t, g_p


def _greedy_init_any(C: torch.Tensor, rows_valid=None, cols_valid=None):
    """relax_matching(C, 0, 0, 0) -- the greedy one-hot initialisation alone (relax_match.py:45-55), what
    compute_matching_loss asks for (match_helper.py:44) -- for tables OUTSIDE the solver kernels' envelope: a handful of
    device tensor ops (order-free: max, first argmin over rows per column, first argmin over columns per row; torch's
    argmin returns the first minimal index).  Inside the envelope ``dmm_relax_solve_f32`` does it in its prologue."""
    B, n, m = C.shape
    dev = C.device
    rv = torch.full((B,), n, dtype=torch.int64, device=dev) if rows_valid is None else rows_valid.long()
    cv = torch.full((B,), m, dtype=torch.int64, device=dev) if cols_valid is None else cols_valid.long()
    live = (torch.arange(n, device=dev)[None, :, None] < rv[:, None, None]) & \
           (torch.arange(m, device=dev)[None, None, :] < cv[:, None, None])
    inf = torch.tensor(float("inf"), device=dev)
    cmax = torch.where(live, C, -inf).flatten(1).max(1)[0]                               # C.max() over the live block
    col_best = torch.where(live, C, inf).argmin(1)                                       # [B, m]: first argmin over rows
    keep = torch.arange(n, device=dev)[None, :, None] == col_best[:, None, :]
    crm = torch.where(live, torch.where(keep, C, cmax[:, None, None].expand_as(C)), inf)
    idx = crm.argmin(2)                                                                  # [B, n]: first argmin over columns
    X = torch.zeros_like(C)
    X.scatter_(2, idx[:, :, None], 1.0)
    X = torch.where(live & (torch.arange(n, device=dev)[None, :, None] < rv[:, None, None]), X, torch.zeros_like(X))
    X = X * ((rv > 0) & (cv > 0))[:, None, None].to(X.dtype)
    return dict(X=X, R=X.clone(), cost=torch.zeros((B, 1), dtype=torch.float32, device=dev),
                iters=torch.zeros((B,), dtype=torch.int32, device=dev))


def relax_solve(C: torch.Tensor, max_iter: int, proj_iter: int, lr: float, rows_valid=None, cols_valid=None):
    """relax_matching on cost matrices C [B,n,m] -> dict(X, R, cost [B,max_iter+1], iters [B]).
    rows_valid / cols_valid (int32 [B]) restrict each frame to its top-left live block."""
    _need_gpu(C)
    C = C.contiguous().float()
    B, n, m = C.shape
    if (n > _lib.MAX_TEMPLATES or m > _lib.MAX_PROPOSALS) and max_iter == 0:
        return _greedy_init_any(C, rows_valid, cols_valid)
    X = torch.empty_like(C)
    R = torch.empty_like(C)
    cost = torch.zeros((B, max_iter + 1), dtype=torch.float32, device=C.device)
    iters = torch.empty((B,), dtype=torch.int32, device=C.device)
    with _lib.device_guard(C.device):
        rc = _lib.load().dmm_relax_solve_f32(_ptr(C), B, n, m, _ptr(rows_valid), _ptr(cols_valid), int(max_iter),
                                             int(proj_iter), float(lr), _ptr(X), _ptr(R), _ptr(cost), _ptr(iters),
                                             _stream(C))
    _lib.check(rc, "dmm_relax_solve_f32")
    return dict(X=X, R=R, cost=cost, iters=iters)


def mask_mix(Rb: torch.Tensor, masks_p: torch.Tensor, n_valid=None, m_valid=None, out_dtype=None,
             shared: bool = False) -> torch.Tensor:
    """full_outmask [B,M,H,W] = Rb [B,M,Pp] @ masks_p [B,N,H*W] (zero planes for the padded columns).
    ``shared``: the rows of Rb share planes (train mode keeps every R > 0.01): every plane of the union of the rows'
    supports is streamed once (``dmm_mask_mix_shared_*``); same result bit for bit."""
    if isinstance(masks_p, FramePlanes):
        fp = masks_p
        _need_gpu(Rb)
        B, N, H, W = fp.B, fp.N, fp.H, fp.W
        M, Pp = Rb.shape[1], Rb.shape[2]
        Rb = Rb.contiguous().float()
        if n_valid is None:                          # frames are ragged by construction: never read past a short one
            n_valid = fp.n_valid()
        if out_dtype not in (None, torch.float32):
            raise ValueError("mask_mix on per-frame plane tables writes fp32 (dmm_mask_mix_frames)")
        out = torch.empty((B, M, H, W), dtype=torch.float32, device=Rb.device)
        with _lib.device_guard(Rb.device):
            L = _lib.load()
            rc = (L.dmm_mask_mix_shared_frames if shared else L.dmm_mask_mix_frames)(_ptr(Rb), _ptr(fp.table), _DT[fp.dtype], B, N, M, Pp, H * W,
                                                 fp.plane_stride, _ptr(n_valid), _ptr(m_valid), _ptr(out), M * H * W,
                                                 H * W, _stream(Rb))
        _lib.check(rc, "dmm_mask_mix_frames")
        return out
    _need_gpu(Rb, masks_p)
    masks_p, sp_b, sp_n = _planes(masks_p)
    B, N, H, W = masks_p.shape
    M, Pp = Rb.shape[1], Rb.shape[2]
    Rb = Rb.contiguous().float()
    out_dtype = out_dtype or torch.float32
    out = torch.empty((B, M, H, W), dtype=out_dtype, device=Rb.device)
    with _lib.device_guard(Rb.device):
        L = _lib.load()
        rc = (L.dmm_mask_mix_shared_to if shared else L.dmm_mask_mix_to)(_ptr(Rb), _ptr(masks_p), _DT[masks_p.dtype], B, N, M, Pp, H * W, sp_b, sp_n,
                                         _ptr(n_valid), _ptr(m_valid), _ptr(out), _DT[out_dtype], M * H * W, H * W,
                                         _stream(Rb))
    _lib.check(rc, "dmm_mask_mix_to")
    return out


def mask_mix_bwd(Rb: torch.Tensor, masks_p: torch.Tensor, dout: torch.Tensor, n_valid=None, m_valid=None):
    """dRb [B,M,Pp] = dout [B,M,H,W] . masks_p [B,N,H,W] on the support of Rb (zeros elsewhere)."""
    if isinstance(masks_p, FramePlanes):
        fp = masks_p
        _need_gpu(Rb, dout)
        B, N, H, W = fp.B, fp.N, fp.H, fp.W
        M, Pp = Rb.shape[1], Rb.shape[2]
        Rb = Rb.contiguous().float()
        if n_valid is None:
            n_valid = fp.n_valid()
        dout = dout.contiguous().float().view(B, M, H * W)
        dRb = torch.empty((B, M, Pp), dtype=torch.float32, device=Rb.device)
        with _lib.device_guard(Rb.device):
            rc = _lib.load().dmm_mask_mix_bwd_frames(_ptr(Rb), _ptr(fp.table), _DT[fp.dtype], _ptr(dout), B, N, M, Pp,
                                                     H * W, fp.plane_stride, _ptr(n_valid), _ptr(m_valid), _ptr(dRb),
                                                     _stream(Rb))
        _lib.check(rc, "dmm_mask_mix_bwd_frames")
        return dRb
    _need_gpu(Rb, masks_p, dout)
    masks_p, sp_b, sp_n = _planes(masks_p)
    B, N, H, W = masks_p.shape
    M, Pp = Rb.shape[1], Rb.shape[2]
    Rb = Rb.contiguous().float()
    dout = dout.contiguous().float().view(B, M, H * W)
    dRb = torch.empty((B, M, Pp), dtype=torch.float32, device=Rb.device)
    with _lib.device_guard(Rb.device):
        rc = _lib.load().dmm_mask_mix_bwd(_ptr(Rb), _ptr(masks_p), _DT[masks_p.dtype], _ptr(dout), B, N, M, Pp, H * W,
                                          sp_b, sp_n, _ptr(n_valid), _ptr(m_valid), _ptr(dRb), _stream(Rb))
    _lib.check(rc, "dmm_mask_mix_bwd")
    return dRb


_WORKSPACES = {}
_WS_STATE = {}                       # (device, stream) -> ((B, N, M, D), ctypes.c_int): dmm_match_forward_ws's note
# HIP stream capture is process-global by default: a second host thread that touches the runtime while one captures
# aborts the process.  Captures are serialised and run in thread-local capture mode (nn.DataParallel-style callers).
_CAPTURE_LOCK = threading.Lock()


def match_forward(masks_p, masks_t, feat_p, feat_t, score_p, *, score_weight, max_iter, proj_iter, lr, is_test,
                  n_valid=None, m_valid=None, return_tables=False):
    """The whole forward of B frames as ONE C-ABI call (``dmm_match_forward``: counts -> normalise -> cosine -> solver
    -> mix on the current stream): fresh output tensors, intermediates in a workspace cached per (device, stream).
    The inference path of ``MatchModel`` (no autograd bookkeeping, 1 ctypes call instead of 7, no per-call
    intermediate allocations).  Returns (full_outmask [B,M,H,W], match_score [B,M], det_score [B,M], iters [B]) and, with
    ``return_tables``, a dict of sim [B,M,N], R and Rb [B,M,Pp].  ANY N and M: tables outside the envelope of the fast
    kernels (M <= 32, Pp = max(N, M + 1) <= 256) take the general kernels of ``dmm_wide.hip`` inside the same call."""
    _need_gpu(masks_p, masks_t, feat_p, feat_t, score_p)
    assert masks_p.dtype == masks_t.dtype and masks_p.dtype in _DT
    masks_p, sp_b, sp_n = _planes(masks_p)
    masks_t, st_b, st_m = _planes(masks_t)
    B, N, H, W = masks_p.shape
    M, D = masks_t.shape[1], feat_p.shape[-1]
    feat_p, feat_t, score_p = feat_p.contiguous().float(), feat_t.contiguous().float(), score_p.contiguous().float()
    dev = masks_p.device
    L = _lib.load()
    stream = _stream(masks_p)
    need = int(L.dmm_workspace_bytes(B, N, M, D))
    key = (dev.index, stream)
    ws = _WORKSPACES.get(key)
    if ws is None or ws.numel() < need:
        ws = _WORKSPACES[key] = torch.empty((need,), dtype=torch.uint8, device=dev)
        _WS_STATE.pop(key, None)
    # dmm_match_forward_ws: the library's note about what the previous call left in THIS workspace; it only holds for the
    # same table layout (B, N, M, D), so another shape starts from "unknown"
    # (a call recorded into a graph is replayed behind our back, on this workspace: from then on no note is kept for it)
    note = None
    if torch.cuda.is_current_stream_capturing() or _WS_STATE.get(key) == "captured":
        _WS_STATE[key] = "captured"
    else:
        shape, note = _WS_STATE.get(key, (None, None))
        if shape != (B, N, M, D):
            note = ctypes.c_int(0)
            _WS_STATE[key] = ((B, N, M, D), note)
    f32 = dict(dtype=torch.float32, device=dev)
    full = torch.empty((B, M, H, W), **f32)
    ms, ds = torch.empty((B, M), **f32), torch.empty((B, M), **f32)
    iters = torch.empty((B,), dtype=torch.int32, device=dev)
    tables = None
    if return_tables:
        Pp = padded_width(N, M)
        tables = {"sim": torch.empty((B, M, N), **f32), "R": torch.empty((B, M, Pp), **f32),
                  "Rb": torch.empty((B, M, Pp), **f32)}
    tp = (lambda k: _ptr(tables[k])) if tables else (lambda k: None)
    with _lib.device_guard(dev):
        rc = L.dmm_match_forward_ws(_ptr(masks_p), _ptr(masks_t), _DT[masks_p.dtype], _ptr(feat_p), _ptr(feat_t),
                                    _ptr(score_p), B, N, M, H * W, D, sp_b, sp_n, st_b, st_m, _ptr(n_valid), _ptr(m_valid),
                                    float(score_weight), int(max_iter), int(proj_iter), float(lr), int(is_test), _ptr(full),
                                    _ptr(ms), _ptr(ds), tp("sim"), tp("R"), tp("Rb"), _ptr(iters), _ptr(ws), ws.numel(),
                                    None if note is None else ctypes.byref(note), stream)
    _lib.check(rc, "dmm_match_forward_ws")
    if return_tables:
        return full, ms, ds, iters, tables
    return full, ms, ds, iters


def match_forward_frame(proposed_mask, mask_last, feat_p, feat_t, score_p, *, score_weight, max_iter, proj_iter, lr, is_test):
    """ONE frame exactly as the reference's evaluator hands it over (``MatchModel.forward`` without targets,
    dmm_model.py:75-77): proposed_mask [P,H,W], mask_last [O,H,W], feat_p [P,D], feat_t [O,D], score_p [P] ->
    (full_outmask [O,H,W], match_score [O], det_score [O]).  ``match_forward`` with B = 1 minus everything a one-frame
    call does not need on the HOST (the call is host bound otherwise): no unsqueeze / index views around the C call, no
    iteration-count or table outputs, three allocations."""
    _need_gpu(proposed_mask, mask_last, feat_p, feat_t, score_p)
    pm, tm = proposed_mask, mask_last
    assert pm.dtype == tm.dtype and pm.dtype in _DT and pm.dim() == 3 and tm.dim() == 3
    N, H, W = pm.shape
    M, D = tm.shape[0], feat_p.shape[-1]
    if N * H * W and not (pm.stride(2) == 1 and pm.stride(1) == W and pm.stride(0) >= H * W):
        pm = pm.contiguous()
    if M * H * W and not (tm.stride(2) == 1 and tm.stride(1) == W and tm.stride(0) >= H * W):
        tm = tm.contiguous()
    feat_p, feat_t, score_p = feat_p.contiguous().float(), feat_t.contiguous().float(), score_p.contiguous().float()
    dev = pm.device
    L = _lib.load()
    stream = torch.cuda.current_stream(dev).cuda_stream
    key = (dev.index, stream)
    ws = _WORKSPACES.get(key)
    need = _WS_NEED.get((1, N, M, D))
    if need is None:
        need = _WS_NEED[(1, N, M, D)] = int(L.dmm_workspace_bytes(1, N, M, D))
    if ws is None or ws.numel() < need:
        ws = _WORKSPACES[key] = torch.empty((need,), dtype=torch.uint8, device=dev)
        _WS_STATE.pop(key, None)
    note = None
    if torch.cuda.is_current_stream_capturing() or _WS_STATE.get(key) == "captured":
        _WS_STATE[key] = "captured"
    else:
        shape, note = _WS_STATE.get(key, (None, None))
        if shape != (1, N, M, D):
            note = ctypes.c_int(0)
            _WS_STATE[key] = ((1, N, M, D), note)
    full = torch.empty((M, H, W), dtype=torch.float32, device=dev)
    ms = torch.empty((M,), dtype=torch.float32, device=dev)
    ds = torch.empty((M,), dtype=torch.float32, device=dev)
    with _lib.device_guard(dev):
        rc = L.dmm_match_forward_ws(pm.data_ptr(), tm.data_ptr(), _DT[pm.dtype], feat_p.data_ptr(), feat_t.data_ptr(),
                                    score_p.data_ptr(), 1, N, M, H * W, D, N * pm.stride(0) if N else 0, pm.stride(0),
                                    M * tm.stride(0) if M else 0, tm.stride(0), None, None, score_weight, max_iter,
                                    proj_iter, lr, is_test, full.data_ptr(), ms.data_ptr(), ds.data_ptr(), None, None, None,
                                    None, ws.data_ptr(), ws.numel(), None if note is None else ctypes.byref(note), stream)
    if rc:
        _lib.check(rc, "dmm_match_forward_ws")
    return full, ms, ds


_WS_NEED = {}                        # (B, N, M, D) [+ solver setting] -> workspace bytes (a C call saved per layer call)


def _cached_ws(key, need, dev):
    ws = _WORKSPACES.get(key)
    if ws is None or ws.numel() < need:
        ws = _WORKSPACES[key] = torch.empty((max(need, 256),), dtype=torch.uint8, device=dev)
    return ws


def _planes3(t: torch.Tensor):
    """[K,H,W] with contiguous H*W planes -> (tensor, plane stride in elements)."""
    K, H, W = t.shape
    if K * H * W and not (t.stride(2) == 1 and t.stride(1) == W and t.stride(0) >= H * W):
        t = t.contiguous()
    return t, t.stride(0)


def match_train_forward(masks_p, masks_t, targets, feat_p, feat_t, score_p, n_valid, m_valid, *, score_weight, max_iter,
                        proj_iter, lr, is_test, one_frame=False, want_tape=True):
    """The training call of B frames as ONE C-ABI call (``dmm_match_train_forward``, include/dmm_match.h (5d)): feature
    similarity -> counts against templates AND targets -> solver -> mix -> matching loss.  masks_p: [B,N,H,W] tensor or
    ``FramePlanes``; targets [B,M,H,W] of the masks' dtype or None.  ``one_frame``: the tensors are ONE frame without the
    batch axis, as the reference's trainer hands them over (masks_p [N,H,W], masks_t / targets [M,H,W], feat_p [N,D],
    feat_t [M,D], score_p [N]) and so are the results -- no unsqueeze / select views around the call.  Returns None when the
    shape is outside the entry's envelope (the caller then runs the granular ops), else
    (full [B,M,H,W], match_score [B,M], det_score [B,M], cost_loss [B] | None, iters [B], saved, taped) with ``saved`` = the
    flat fp32 block cos | sim | Rb | gt (| the solver's tape, 256-byte aligned) that ``match_train_backward`` reads and
    ``taped`` = whether the forward's solver kernel kept its tape there (tables of <= 64 solver columns: the backward then
    does not re-run the solver).  ``want_tape=False`` (no gradient will be asked for): the untaped solver kernels."""
    fp = masks_p if isinstance(masks_p, FramePlanes) else None
    if one_frame:
        masks_p, sp_n = _planes3(masks_p)
        B, (N, H, W) = 1, masks_p.shape
        sp_b = N * sp_n
        dt, p_ptr, dev = masks_p.dtype, masks_p.data_ptr(), masks_p.device
        masks_t, st_m = _planes3(masks_t)
        M = masks_t.shape[0]
        st_b = M * st_m
    else:
        if fp is not None:
            B, N, H, W, dt = fp.B, fp.N, fp.H, fp.W, fp.dtype
            sp_b, sp_n, p_ptr, dev = _lib.FRAME_TABLE, fp.plane_stride, fp.table.data_ptr(), fp.device
        else:
            masks_p, sp_b, sp_n = _planes(masks_p)
            B, N, H, W = masks_p.shape
            dt, p_ptr, dev = masks_p.dtype, masks_p.data_ptr(), masks_p.device
        masks_t, st_b, st_m = _planes(masks_t)
        M = masks_t.shape[1]
    D = feat_p.shape[-1]
    Pp = padded_width(N, M)
    if M > _lib.MAX_TEMPLATES or Pp > _lib.MAX_PROPOSALS or B > 65535 or N == 0 or M == 0 or B == 0 or dt not in _DT \
            or masks_t.dtype != dt:
        return None
    sg_b = sg_m = 0
    g_ptr = None
    if targets is not None:
        if targets.dtype != dt:
            targets = targets.to(dt)
        if one_frame:
            targets, sg_m = _planes3(targets)
            sg_b = M * sg_m
        else:
            targets, sg_b, sg_m = _planes(targets)
        g_ptr = targets.data_ptr()
    L = _lib.load()
    stream = torch.cuda.current_stream(dev).cuda_stream
    need = _WS_NEED.get(("tf", B, N, M, D))
    if need is None:
        need = _WS_NEED[("tf", B, N, M, D)] = int(L.dmm_match_train_forward_workspace_bytes(B, N, M, D))
    ws = _cached_ws((dev.index, stream, "train_fwd"), need, dev)
    f32 = dict(dtype=torch.float32, device=dev)
    lead = () if one_frame else (B,)
    full = torch.empty(lead + (M, H, W), **f32)
    ms, ds = torch.empty(lead + (M,), **f32), torch.empty(lead + (M,), **f32)
    loss = torch.empty(lead, **f32) if targets is not None else None
    iters = torch.empty((B,), dtype=torch.int32, device=dev)
    n_cs, n_rb = B * M * N, B * M * Pp
    tape_off, tape_bytes = _train_tape_layout(L, B, N, M, max_iter, proj_iter)
    if not want_tape:
        tape_bytes = 0
    saved = torch.empty((tape_off // 4 + tape_bytes // 4,), **f32)
    sp = saved.data_ptr()
    taped = ctypes.c_int(0)
    with _lib.device_guard(dev):
        rc = L.dmm_match_train_forward(p_ptr, masks_t.data_ptr(), g_ptr, _DT[dt], feat_p.data_ptr(), feat_t.data_ptr(),
                                       score_p.data_ptr(), B, N, M, H * W, D, sp_b, sp_n, st_b, st_m, sg_b, sg_m,
                                       _ptr(n_valid), _ptr(m_valid), score_weight, max_iter, proj_iter, lr, is_test,
                                       full.data_ptr(), ms.data_ptr(), ds.data_ptr(), _ptr(loss), iters.data_ptr(), sp,
                                       sp + 4 * n_cs, sp + 8 * n_cs, (sp + 8 * n_cs + 4 * n_rb) if targets is not None else None,
                                       ws.data_ptr(), ws.numel(), (sp + tape_off) if tape_bytes else None, tape_bytes,
                                       ctypes.byref(taped), stream)
    if rc == 2:                                               # DMM_ERR_UNSUPPORTED: nothing was launched
        return None
    if rc:
        _lib.check(rc, "dmm_match_train_forward")
    return full, ms, ds, loss, iters, saved, int(taped.value)


def _train_tape_layout(L, B, N, M, max_iter, proj_iter):
    """-> (byte offset of the solver's tape inside the training call's saved block, its size; 0 = not taped)."""
    k = ("tt", B, N, M, max_iter, proj_iter)
    got = _WS_NEED.get(k)
    if got is None:
        n_cs, n_rb = B * M * N, B * M * padded_width(N, M)
        off = -(-(4 * (3 * n_cs + n_rb)) // 256) * 256
        nbytes = int(L.dmm_match_train_tape_bytes(B, N, M, int(max_iter), int(proj_iter)))
        # the tape is sized for max_iter x proj_iter sweeps whatever the solver executes (the shipped default 400 x 50 is
        # 10 MB per frame, held until the backward): past _TAPE_MAX_BYTES per call the backward re-runs the solver instead
        got = _WS_NEED[k] = (off, nbytes if nbytes <= _TAPE_MAX_BYTES else 0)
    return got


_TAPE_MAX_BYTES = 32 << 20          # (held until the backward, per frame step alive in the autograd graph)


def match_train_backward(masks_p, feat_p, feat_t, score_p, saved, has_loss, d_full, d_ms, d_ds, d_loss, n_valid, m_valid,
                         M, *, score_weight, max_iter, proj_iter, lr, is_test, one_frame=False, iters=None, taped=0):
    """-> (g_feat_t [B,M,D], g_feat_p [B,N,D]): the whole backward of ``match_train_forward`` as ONE C-ABI call
    (``dmm_match_train_backward``, (5e)): normalise both feature sets -> mix backward -> taped solver backward ->
    feature-similarity backward.  ``saved`` is the forward's block; d_* may be None.  ``one_frame``: as in the forward.
    ``iters`` / ``taped``: the forward's iteration counts and its ``taped`` flag -- with them the solver's backward walks the
    tape inside ``saved`` instead of re-running the solver."""
    fp = masks_p if isinstance(masks_p, FramePlanes) else None
    if one_frame:
        masks_p, sp_n = _planes3(masks_p)
        B, (N, H, W) = 1, masks_p.shape
        sp_b = N * sp_n
        dt, p_ptr, dev = masks_p.dtype, masks_p.data_ptr(), masks_p.device
    elif fp is not None:
        B, N, H, W, dt = fp.B, fp.N, fp.H, fp.W, fp.dtype
        sp_b, sp_n, p_ptr, dev = _lib.FRAME_TABLE, fp.plane_stride, fp.table.data_ptr(), fp.device
    else:
        masks_p, sp_b, sp_n = _planes(masks_p)
        B, N, H, W = masks_p.shape
        dt, p_ptr, dev = masks_p.dtype, masks_p.data_ptr(), masks_p.device
    D = feat_p.shape[-1]
    Pp = padded_width(N, M)
    n_cs, n_rb = B * M * N, B * M * Pp
    L = _lib.load()
    stream = torch.cuda.current_stream(dev).cuda_stream
    k = ("tb", B, N, M, D, max_iter, proj_iter)
    need = _WS_NEED.get(k)
    if need is None:
        need = _WS_NEED[k] = int(L.dmm_match_train_backward_workspace_bytes(B, N, M, D, max_iter, proj_iter))
    ws = _cached_ws((dev.index, stream, "train_bwd"), need, dev)
    cf = lambda t: None if t is None else t.contiguous().float()
    d_full, d_ms, d_ds, d_loss = cf(d_full), cf(d_ms), cf(d_ds), cf(d_loss)
    use_loss = has_loss and d_loss is not None
    g_t, g_p = torch.empty_like(feat_t), torch.empty_like(feat_p)
    sp = saved.data_ptr()
    tape_off, tape_bytes = _train_tape_layout(L, B, N, M, max_iter, proj_iter)
    walk = bool(taped) and iters is not None and tape_bytes > 0 and saved.numel() * 4 >= tape_off + tape_bytes
    with _lib.device_guard(dev):
        rc = L.dmm_match_train_backward(p_ptr, _DT[dt], feat_p.data_ptr(), feat_t.data_ptr(), score_p.data_ptr(),
                                        sp if use_loss else None, sp + 4 * n_cs, sp + 8 * n_cs,
                                        (sp + 8 * n_cs + 4 * n_rb) if use_loss else None, _ptr(d_full), _ptr(d_ms),
                                        _ptr(d_ds), _ptr(d_loss) if use_loss else None, B, N, M, H * W, D, sp_b, sp_n,
                                        _ptr(n_valid), _ptr(m_valid), score_weight, max_iter, proj_iter, lr, is_test,
                                        g_t.data_ptr(), g_p.data_ptr(), ws.data_ptr(), ws.numel(),
                                        (sp + tape_off) if walk else None, _ptr(iters) if walk else None, int(walk), stream)
    if rc:
        _lib.check(rc, "dmm_match_train_backward")
    return g_t, g_p


def match_forward_packed(masks_p, packed_p, masks_t, feat_p, feat_t, score_p, n_valid, m_valid, *, score_weight, max_iter,
                         proj_iter, lr, is_test, out=None, workspace=None):
    """``match_forward`` with the proposal side of the cost pass on the 1-bit planes ``packed_p`` [B,N,words] the caller
    holds next to the soft planes (``proposals.ProposalSlots``); ``dmm_match_forward_packed``.  ``out`` = (full [B,M,H,W],
    match_score [B,M], det_score [B,M], iters [B]) and ``workspace`` (uint8) may be caller-owned: then nothing is
    allocated (a captured frame step).  Returns the ``out`` tuple."""
    _need_gpu(masks_p, packed_p, masks_t, feat_p, feat_t, score_p)
    assert masks_p.dtype == masks_t.dtype and masks_p.dtype in _DT and packed_p.dtype == torch.int64
    masks_p, sp_b, sp_n = _planes(masks_p)
    masks_t, st_b, st_m = _planes(masks_t)
    B, N, H, W = masks_p.shape
    M, D = masks_t.shape[1], feat_p.shape[-1]
    assert packed_p.is_contiguous() and packed_p.shape == (B, N, pack_words(H * W))
    assert feat_p.is_contiguous() and feat_t.is_contiguous() and score_p.is_contiguous()
    assert feat_p.dtype == feat_t.dtype == score_p.dtype == torch.float32
    dev = masks_p.device
    L = _lib.load()
    need = int(L.dmm_workspace_bytes_packed(B, N, M, D, H * W))
    if workspace is None:
        key = (dev.index, _stream(masks_p), "packed")
        workspace = _WORKSPACES.get(key)
        if workspace is None or workspace.numel() < need:
            workspace = _WORKSPACES[key] = torch.empty((need,), dtype=torch.uint8, device=dev)
    assert workspace.numel() >= need
    if out is None:
        f32 = dict(dtype=torch.float32, device=dev)
        out = (torch.empty((B, M, H, W), **f32), torch.empty((B, M), **f32), torch.empty((B, M), **f32),
               torch.empty((B,), dtype=torch.int32, device=dev))
    full, ms, ds, iters = out
    assert full.is_contiguous() and full.shape == (B, M, H, W) and full.dtype == torch.float32
    wd = packed_p.shape[2]
    with _lib.device_guard(dev):
        rc = L.dmm_match_forward_packed(_ptr(masks_p), _ptr(packed_p), _ptr(masks_t), _DT[masks_p.dtype], _ptr(feat_p),
                                        _ptr(feat_t), _ptr(score_p), B, N, M, H * W, D, sp_b, sp_n, N * wd, wd, st_b, st_m,
                                        _ptr(n_valid), _ptr(m_valid), float(score_weight), int(max_iter), int(proj_iter),
                                        float(lr), int(is_test), _ptr(full), _ptr(ms), _ptr(ds), None, None, None,
                                        _ptr(iters), _ptr(workspace), workspace.numel(), _stream(masks_p))
    _lib.check(rc, "dmm_match_forward_packed")
    return out


def match_solve_packed(packed_p, packed_t, feat_p, feat_t, score_p, n_valid, m_valid, HW, *, score_weight, max_iter,
                        proj_iter, lr, is_test, out, workspace):
    """Cost + assignment on 1-bit planes of BOTH sides, no mix (``dmm_match_solve_packed``): packed_p [B,N,words],
    packed_t [B,M,words]; ``out`` = (Rb [B,M,Pp], match_score [B,M], det_score [B,M], iters [B]) caller-owned, like the
    workspace (>= dmm_workspace_bytes): nothing is allocated -- the middle of a captured frame step."""
    _need_gpu(packed_p, packed_t, feat_p, feat_t, score_p)
    B, N, wd = packed_p.shape
    M, D = packed_t.shape[1], feat_p.shape[-1]
    assert packed_p.is_contiguous() and packed_t.is_contiguous() and packed_t.shape == (B, M, wd) and wd == pack_words(HW)
    assert feat_p.is_contiguous() and feat_t.is_contiguous() and score_p.is_contiguous()
    Rb, ms, ds, iters = out
    assert Rb.is_contiguous() and Rb.shape == (B, M, padded_width(N, M))
    L = _lib.load()
    assert workspace.numel() >= int(L.dmm_workspace_bytes(B, N, M, D))
    with _lib.device_guard(packed_p.device):
        rc = L.dmm_match_solve_packed(_ptr(packed_p), _ptr(packed_t), _ptr(feat_p), _ptr(feat_t), _ptr(score_p), B, N, M,
                                      int(HW), D, _ptr(n_valid), _ptr(m_valid), float(score_weight), int(max_iter),
                                      int(proj_iter), float(lr), int(is_test), _ptr(Rb), _ptr(ms), _ptr(ds), None, None,
                                      _ptr(iters), _ptr(workspace), workspace.numel(), _stream(packed_p))
    _lib.check(rc, "dmm_match_solve_packed")
    return out


class ForwardPlan:
    """Pre-allocated forward of B same-shaped frames: nothing is allocated or synchronised per call.

    Two execution shapes:

    * ``pipeline=False`` -- one ``dmm_match_forward`` C call, all kernels back to back on the current stream;
      With ``graph=True`` (small batches, where a frame step is launch / latency bound) the sequence is captured into
      a HIP graph the second time in a row ``run`` sees the same tensors (same addresses and strides) and replayed
      from then on -- one graph launch instead of 7 kernel launches (``graph_fork=True`` also runs the two independent
      branches of the layer, IoU counts (masks) | feature similarity, side by side before the solver joins them).
      Up to 4 tensor sets are kept (each captured graph holds references to its tensors); other calls launch directly.
    * ``pipeline=True``  -- "streaming lane + latency lane".  The batch is split in two halves A, B.  The
      current stream runs only the HBM-bound kernels, serialised at full bandwidth:
      cost(A) -> cost(B) -> mix(A) -> mix(B)  (A, B = the two halves of the batch).  A side stream runs the latency-bound ones:
      normalise + cosine (all frames) -> solver(A) (after cost(A)) -> solver(B) (after cost(B)).
      solver(A) hides under cost(B), solver(B) under mix(A); HIP events carry the dependencies, there is no
      host synchronisation, and every output is complete in current-stream order when ``run`` returns.
    """

    def __init__(self, B, N, M, H, W, D, device, mask_dtype=torch.float32, want_tables=False, pipeline=None,
                 split=0.5, time_kernels=False, graph=None, out_dtype=None, parts=2, graph_fork=False,
                 solver_state="f32", out_plane_align=0):
        self.B, self.N, self.M, self.H, self.W, self.D = B, N, M, H, W, D
        self.Pp = padded_width(N, M)
        self.device = torch.device(device)
        self.mask_dtype = mask_dtype
        # "f16": dmm_relax_match_f16s (packed-fp16 solver state, fp32 sums; opt-in tolerance mode of BASELINE configs[4]).
        # Its <= 128 VGPRs let it share a SIMD with the streaming kernels, so wide tables can take the 2-lane schedule.
        assert solver_state in ("f32", "f16")
        self.solver_state = solver_state
        # default: two lanes for large batches whose solver fits beside the streaming kernels (one wave per frame, exact
        # row count: M <= 16, Pp <= 64); the multi-wave solvers of wide tables hold up to 256 VGPRs per wave and only
        # serialise with them (config 5, fp32 state: 2.26 ms single stream vs 2.37 ms two lanes per 256 frames)
        # (round 5, tools/config5_probe.py: with the fp16-state solver the wide tables of config 5 run 3.05 ms per 512 frames
        # on one stream against 3.08-3.46 on two lanes -- the solver's waves slow the CUs they land on and the statically
        # partitioned count launch waits for its slowest CU -- so wide tables take one stream whatever the solver state)
        auto = B >= 64 and M <= 16 and self.Pp <= 64
        self.pipeline = auto if pipeline is None else bool(pipeline)
        # time_kernels: the single-stream form issues the granular C-ABI calls (same kernels as dmm_match_forward) so
        # that HIP events can bracket the cost and mix launches; bench.py sets kernel_events = {} per timed step
        # out_dtype: fp32 (default, the nn.Module boundary) or the planes' own 16-bit type (config 5: the matched masks
        # are the next frame's fp16 templates); the fused single C call writes fp32 only -> granular launches otherwise
        self.out_dtype = out_dtype or torch.float32
        assert self.out_dtype in (torch.float32, mask_dtype)
        self.time_kernels = bool(time_kernels) or self.out_dtype != torch.float32 or solver_state != "f32"
        self.kernel_events = None
        # opt-in: it only pays for callers that present the SAME tensors again (static buffers: bench loops, serving
        # loops over GraphedEncoder outputs); a frame loop with fresh proposal tensors would capture and never replay
        self.graph_mode = bool(graph) and not self.pipeline and not self.time_kernels
        # graph_fork: feature branch on a second stream inside the captured graph.  It paid while the feature similarity
        # took as long as the counts (18 us vs 19 us at B = 1); with the lanes kernel (10 us) and the small-chunk count
        # launch (round 2) the ~18 us a cross-queue fork + join costs is more than the branch saves: default off
        self.graph_fork = bool(graph_fork)
        self._graphs, self._last_key = {}, None                   # key -> (graph, tensors it holds addresses of, ws in / out)
        # dmm_match_forward_ws: what the last call of this plan left in ITS workspace (DMM_WS_UNKNOWN / _TABLES_ZERO).
        # From _TABLES_ZERO a handful of dense frames start their counts without a clearing launch (include/dmm_match.h 5a').
        self._ws_state = ctypes.c_int(0)
        L = _lib.load()
        f32 = dict(dtype=torch.float32, device=self.device)
        i32 = dict(dtype=torch.int32, device=self.device)
        # out_plane_align: plane stride of full_outmask rounded up to that many bytes (alloc_planes) -- the matched masks of
        # config 5 are the next frame's 16-bit templates, so their producer writes them line aligned
        if out_plane_align:
            self.full_outmask = alloc_planes(B, M, H, W, self.out_dtype, self.device, int(out_plane_align))
        else:
            self.full_outmask = torch.empty((B, M, H, W), dtype=self.out_dtype, device=self.device)
        self.so_b, self.so_m = self.full_outmask.stride(0), self.full_outmask.stride(1)
        # the fused C call writes contiguous fp32 only: aligned output planes take the granular launches -- which cannot be
        # the graph-replay form, so asking for both is an error rather than a silent downgrade
        if out_plane_align and graph:
            raise ValueError("ForwardPlan(out_plane_align=..., graph=True): line-aligned output planes are written by the "
                             "granular launches only (no HIP-graph replay); drop one of the two")
        self.time_kernels = self.time_kernels or bool(out_plane_align)
        self.graph_mode = self.graph_mode and not out_plane_align
        self.match_score = torch.empty((B, M), **f32)
        self.det_score = torch.empty((B, M), **f32)
        self.iters = torch.empty((B,), **i32)
        self.sim = torch.empty((B, M, N), **f32)
        self.R = torch.empty((B, M, self.Pp), **f32) if want_tables else None
        self.Rb = torch.empty((B, M, self.Pp), **f32)
        if not self.pipeline:
            self.ws_bytes = int(L.dmm_workspace_bytes(B, N, M, D))
            self.workspace = torch.empty((self.ws_bytes,), dtype=torch.uint8, device=self.device)
            if self.time_kernels or self.graph_mode:
                if self.graph_mode:
                    with _lib.device_guard(self.device):
                        self.side = torch.cuda.Stream(device=self.device)
                self.counts = [torch.empty((B * (M * N + N + M),), **i32)]
                self.halves = [(0, B)]
                self.pn = torch.empty((B, N, D), **f32)
                self.tn = torch.empty((B, M, D), **f32)
                self.cos = torch.empty((B, M, N), **f32)
            return
        # ``parts`` equal slices of the batch (2 = halves A, B; with split != 0.5 the first slice takes that fraction).
        # Measured at B = 1024, config 2: 1:1 gives 255 k frames/s, 7:1 only 245 k (the normalise / cosine kernels then
        # overlap one long cost launch and slow it down by what they cost alone).
        parts = max(1, min(int(parts), B))
        if parts == 2 and B > 1:
            cut = min(B - 1, max(1, int(round(B * split))))
            self.halves = [(0, cut), (cut, B)]
        else:
            edges = [round(B * k / parts) for k in range(parts + 1)]
            self.halves = [(edges[k], edges[k + 1]) for k in range(parts) if edges[k + 1] > edges[k]]
        # inter | area_p | area_t of one half are contiguous -> one memset per cost launch
        self.counts = [torch.empty(((e - b) * (M * N + N + M),), **i32) for (b, e) in self.halves]
        self.pn = torch.empty((B, N, D), **f32)
        self.tn = torch.empty((B, M, D), **f32)
        self.cos = torch.empty((B, M, N), **f32)
        with _lib.device_guard(self.device):
            # (stream priorities make no measurable difference here: 261.1 / 262.0 k frames/s at priority 0 / -1)
            self.side = torch.cuda.Stream(device=self.device)
            self.ev_start = torch.cuda.Event()
            self.ev_cost = [torch.cuda.Event() for _ in self.halves]
            self.ev_solved = [torch.cuda.Event() for _ in self.halves]

    def schedule_name(self) -> str:
        if self.pipeline:
            return "streaming lane (cost, mix) + latency lane (normalise, cosine, solver) on 2 HIP streams" + \
                (" [fp16-state solver]" if self.solver_state == "f16" else "")
        if self.graph_mode:
            if not self._graphs:
                return "single stream (HIP graph armed: captured on the 2nd call with the same tensors)"
            return "HIP graph replay (IoU counts | feature similarity in parallel -> solver -> mix)" if self.graph_fork \
                else "HIP graph replay (feature similarity + IoU counts [one launch at <= 8 dense frames] -> solver -> mix, one chain)"
        return "single stream"

    def _launch_forked(self, L, masks_p, masks_t, feat_p, feat_t, score_p, dt, strides, n_valid, m_valid, cfg):
        """Granular launches, the form that is captured into the HIP graph (``graph_fork``: feature branch on the side
        stream, fork / join by stream waits)."""
        B, N, M, D, Pp, HW = self.B, self.N, self.M, self.D, self.Pp, self.H * self.W
        sp_b, sp_n, st_b, st_m = strides
        score_weight, max_iter, proj_iter, lr, is_test = cfg
        main = torch.cuda.current_stream(self.device)
        if not self.graph_fork:
            # one chain: the fused C call (its feature-similarity launch also clears the count tables: no memset node)
            _lib.check(L.dmm_match_forward_ws(
                _ptr(masks_p), _ptr(masks_t), dt, _ptr(feat_p), _ptr(feat_t), _ptr(score_p), B, N, M, HW, D, sp_b, sp_n,
                st_b, st_m, _ptr(n_valid), _ptr(m_valid), float(score_weight), int(max_iter), int(proj_iter), float(lr),
                int(is_test), _ptr(self.full_outmask), _ptr(self.match_score), _ptr(self.det_score), _ptr(self.sim),
                _ptr(self.R), _ptr(self.Rb), _ptr(self.iters), _ptr(self.workspace), self.ws_bytes,
                ctypes.byref(self._ws_state), main.cuda_stream),
                "dmm_match_forward_ws (graph capture)")
            return
        side = self.side
        inter, ap, at = self._tables(0)
        if self.graph_fork:
            side.wait_stream(main)
        ss, ms = side.cuda_stream, main.cuda_stream
        rc = self._feature_sim(L, feat_p, feat_t, n_valid, m_valid, ss)
        rc |= L.dmm_iou_counts(_ptr(masks_p), _ptr(masks_t), dt, B, N, M, HW, sp_b, sp_n, st_b, st_m, _ptr(n_valid),
                               _ptr(m_valid), _ptr(inter), _ptr(ap), _ptr(at), ms)
        if self.graph_fork:
            main.wait_stream(side)
        rc |= self._solver(L)(_ptr(self.cos), _ptr(inter), _ptr(ap), _ptr(at), _ptr(score_p), B, N, M,
                               _ptr(n_valid), _ptr(m_valid), float(score_weight), int(max_iter), int(proj_iter),
                               float(lr), int(is_test), _ptr(self.sim), _ptr(self.R), _ptr(self.Rb),
                               _ptr(self.match_score), _ptr(self.det_score), _ptr(self.iters), None, ms)
        rc |= L.dmm_mask_mix_to(_ptr(self.Rb), _ptr(masks_p), dt, B, N, M, Pp, HW, sp_b, sp_n, _ptr(n_valid),
                                _ptr(m_valid), _ptr(self.full_outmask), _DT[self.out_dtype], self.so_b, self.so_m, ms)
        _lib.check(rc, "ForwardPlan.run (forked)")

    def _solver(self, L):
        return L.dmm_relax_match_f32 if self.solver_state == "f32" else L.dmm_relax_match_f16s

    def _mark(self, name, stream, begin):
        """HIP event on ``stream`` before / after a kernel launch when bench.py asked for kernel timing."""
        if self.kernel_events is None:
            return
        e = torch.cuda.Event(enable_timing=True)
        e.record(stream)
        if begin:
            self.kernel_events.setdefault(name, []).append([e, None])
        else:
            self.kernel_events[name][-1][1] = e

    def _feature_sim(self, L, feat_p, feat_t, n_valid, m_valid, stream) -> int:
        """cos[B,M,N] into self.cos on ``stream``: the fused one-launch kernel for dense batches inside its envelope,
        normalise + normalise + cosine otherwise."""
        B, N, M, D = self.B, self.N, self.M, self.D
        if n_valid is None and m_valid is None:
            rc = L.dmm_cosine_features_f32(_ptr(feat_t), _ptr(feat_p), B, N, M, D, _ptr(self.cos), stream)
            if rc != 2:
                return rc
        rc = L.dmm_feature_normalize_f32(_ptr(feat_p), B * N, D, _ptr(self.pn), None, stream)
        rc |= L.dmm_feature_normalize_f32(_ptr(feat_t), B * M, D, _ptr(self.tn), None, stream)
        rc |= L.dmm_cosine_f32(_ptr(self.tn), _ptr(self.pn), B, N, M, D, _ptr(n_valid), _ptr(m_valid), _ptr(self.cos),
                               stream)
        return rc

    def _tables(self, h):
        (b, e) = self.halves[h]
        nb, M, N = e - b, self.M, self.N
        c = self.counts[h]
        return c[:nb * M * N], c[nb * M * N:nb * (M * N + N)], c[nb * (M * N + N):]

    def run(self, masks_p, masks_t, feat_p, feat_t, score_p, *, score_weight=0.3, max_iter=20, proj_iter=5, lr=0.1,
            is_test=1, n_valid=None, m_valid=None):
        _need_gpu(masks_p, masks_t, feat_p, feat_t, score_p)
        masks_p, sp_b, sp_n = _planes(masks_p)
        masks_t, st_b, st_m = _planes(masks_t)
        B, N, M, H, W, D, Pp = self.B, self.N, self.M, self.H, self.W, self.D, self.Pp
        HW = H * W
        assert masks_p.shape == (B, N, H, W) and masks_t.shape == (B, M, H, W)
        assert masks_p.dtype == self.mask_dtype and masks_t.dtype == self.mask_dtype
        assert feat_p.is_contiguous() and feat_t.is_contiguous() and score_p.is_contiguous()
        assert feat_p.dtype == torch.float32 and feat_t.dtype == torch.float32 and score_p.dtype == torch.float32
        L = _lib.load()
        dt = _DT[self.mask_dtype]
        if self.graph_mode and not torch.cuda.is_current_stream_capturing():
            cfg = (float(score_weight), int(max_iter), int(proj_iter), float(lr), int(is_test))
            key = (masks_p.data_ptr(), masks_t.data_ptr(), feat_p.data_ptr(), feat_t.data_ptr(), score_p.data_ptr(),
                   sp_b, sp_n, st_b, st_m, _ptr(n_valid), _ptr(m_valid), cfg)
            hit = self._graphs.get(key)
            # a captured call must find the workspace in the state it was captured with; if something else ran in between
            # (another tensor set, a direct call that took a different path) this call goes directly -- which heals it
            if hit is not None and hit[2] in (0, self._ws_state.value):
                hit[0].replay()
                self._ws_state.value = hit[3]
                return self.full_outmask, self.match_score, self.det_score
            if hit is None and key == self._last_key and len(self._graphs) < 4:
                # second call in a row on the same tensors: capture (the first one ran directly = the warm-up)
                ws_in = self._ws_state.value
                with _CAPTURE_LOCK, torch.cuda.device(self.device):
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, capture_error_mode="thread_local"):
                        self._launch_forked(L, masks_p, masks_t, feat_p, feat_t, score_p, dt, (sp_b, sp_n, st_b, st_m),
                                            n_valid, m_valid, cfg)
                ws_out = self._ws_state.value                       # what a replay leaves (nothing ran during the capture)
                self._ws_state.value = ws_in
                # the graph holds raw addresses: keep the tensors alive with it
                self._graphs[key] = (g, (masks_p, masks_t, feat_p, feat_t, score_p, n_valid, m_valid), ws_in, ws_out)
                g.replay()
                self._ws_state.value = ws_out
                return self.full_outmask, self.match_score, self.det_score
            self._last_key = key
        if not self.pipeline and self.time_kernels:
            with _lib.device_guard(self.device):
                main = torch.cuda.current_stream(self.device)
                ms = main.cuda_stream
                inter, ap, at = self._tables(0)
                self._mark("cost", main, True)
                rc = L.dmm_iou_counts(_ptr(masks_p), _ptr(masks_t), dt, B, N, M, HW, sp_b, sp_n, st_b, st_m,
                                      _ptr(n_valid), _ptr(m_valid), _ptr(inter), _ptr(ap), _ptr(at), ms)
                self._mark("cost", main, False)
                rc |= self._feature_sim(L, feat_p, feat_t, n_valid, m_valid, ms)
                self._mark("solver", main, True)
                rc |= self._solver(L)(_ptr(self.cos), _ptr(inter), _ptr(ap), _ptr(at), _ptr(score_p), B, N, M,
                                       _ptr(n_valid), _ptr(m_valid), float(score_weight), int(max_iter),
                                       int(proj_iter), float(lr), int(is_test), _ptr(self.sim), _ptr(self.R),
                                       _ptr(self.Rb), _ptr(self.match_score), _ptr(self.det_score),
                                       _ptr(self.iters), None, ms)
                self._mark("solver", main, False)
                self._mark("mix", main, True)
                rc |= L.dmm_mask_mix_to(_ptr(self.Rb), _ptr(masks_p), dt, B, N, M, Pp, HW, sp_b, sp_n, _ptr(n_valid),
                                        _ptr(m_valid), _ptr(self.full_outmask), _DT[self.out_dtype], self.so_b, self.so_m, ms)
                self._mark("mix", main, False)
            _lib.check(rc, "ForwardPlan.run (granular, timed)")
            return self.full_outmask, self.match_score, self.det_score
        if not self.pipeline:
            with _lib.device_guard(self.device):
                rc = L.dmm_match_forward_ws(
                    _ptr(masks_p), _ptr(masks_t), dt, _ptr(feat_p), _ptr(feat_t), _ptr(score_p), B, N, M, HW, D, sp_b,
                    sp_n, st_b, st_m, _ptr(n_valid), _ptr(m_valid), float(score_weight), int(max_iter), int(proj_iter),
                    float(lr), int(is_test), _ptr(self.full_outmask), _ptr(self.match_score), _ptr(self.det_score),
                    _ptr(self.sim), _ptr(self.R), _ptr(self.Rb), _ptr(self.iters), _ptr(self.workspace), self.ws_bytes,
                    ctypes.byref(self._ws_state), _stream(masks_p))
            _lib.check(rc, "dmm_match_forward_ws")
            return self.full_outmask, self.match_score, self.det_score

        es = masks_p.element_size()
        nv = lambda t, b: None if t is None else t.data_ptr() + 4 * b
        with _lib.device_guard(self.device):
            main = torch.cuda.current_stream(self.device)
            side = self.side
            ms, ss = main.cuda_stream, side.cuda_stream
            self.ev_start.record(main)
            # ---- latency lane: normalise + cosine for every frame --------------------------------------------
            side.wait_event(self.ev_start)
            rc = self._feature_sim(L, feat_p, feat_t, n_valid, m_valid, ss)
            # ---- streaming lane: cost(A), cost(B) ------------------------------------------------------------
            for h, (b, e) in enumerate(self.halves):
                inter, ap, at = self._tables(h)
                self._mark("cost", main, True)
                rc |= L.dmm_iou_counts(masks_p.data_ptr() + es * b * sp_b, masks_t.data_ptr() + es * b * st_b, dt,
                                       e - b, N, M, HW, sp_b, sp_n, st_b, st_m, nv(n_valid, b), nv(m_valid, b),
                                       _ptr(inter), _ptr(ap), _ptr(at), ms)
                self._mark("cost", main, False)
                self.ev_cost[h].record(main)
            # ---- latency lane: solver(h) as soon as cost(h) is done ------------------------------------------
            for h, (b, e) in enumerate(self.halves):
                inter, ap, at = self._tables(h)
                side.wait_event(self.ev_cost[h])
                rc |= self._solver(L)(
                    self.cos.data_ptr() + 4 * b * M * N, _ptr(inter), _ptr(ap), _ptr(at), score_p.data_ptr() + 4 * b * N,
                    e - b, N, M, nv(n_valid, b), nv(m_valid, b), float(score_weight), int(max_iter), int(proj_iter),
                    float(lr), int(is_test), self.sim.data_ptr() + 4 * b * M * N,
                    None if self.R is None else self.R.data_ptr() + 4 * b * M * Pp, self.Rb.data_ptr() + 4 * b * M * Pp,
                    self.match_score.data_ptr() + 4 * b * M, self.det_score.data_ptr() + 4 * b * M,
                    self.iters.data_ptr() + 4 * b, None, ss)
                self.ev_solved[h].record(side)
            # ---- streaming lane: mix(A) after solver(A), mix(B) after solver(B) ------------------------------
            for h, (b, e) in enumerate(self.halves):
                main.wait_event(self.ev_solved[h])
                self._mark("mix", main, True)
                rc |= L.dmm_mask_mix_to(self.Rb.data_ptr() + 4 * b * M * Pp, masks_p.data_ptr() + es * b * sp_b, dt,
                                        e - b, N, M, Pp, HW, sp_b, sp_n, nv(n_valid, b), nv(m_valid, b),
                                        self.full_outmask.data_ptr() + self.full_outmask.element_size() * b * self.so_b,
                                        _DT[self.out_dtype], self.so_b, self.so_m, ms)
                self._mark("mix", main, False)
        _lib.check(rc, "ForwardPlan.run (pipelined)")
        return self.full_outmask, self.match_score, self.det_score
