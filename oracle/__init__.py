"""CPU oracle for the DMM-Net matching layer -- TEST INFRASTRUCTURE ONLY.

ctypes/numpy front end of ``oracle/dmm_oracle.c`` (a plain-C restatement of the reference's
``dmm/modules/match_model.py``, ``dmm/utils/match_helper.py`` and
``dmm/modules/submodules/relax_match.py``; each C function cites the lines it follows).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package.  The product (``dmm_net_amd``) never does: it fails loudly if the HIP library is
missing instead of falling back to anything here.

Parity status: pinned by ``tests/golden/*.npz`` (captured from the imported reference by
``tests/golden/gen_golden.py``) in ``tests/test_oracle_golden.py``.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libdmm_oracle.so")
_lib = None

_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_vp = ctypes.c_void_p


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "dmm_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libdmm_oracle.so"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        c_int, c_float = ctypes.c_int, ctypes.c_float
        L.dmmo_iou_counts.argtypes = [_f32p, c_int, _f32p, c_int, c_int, _i32p, _i32p, _i32p]
        L.dmmo_iou_counts.restype = None
        L.dmmo_iou_from_counts.argtypes = [_i32p, _i32p, _i32p, c_int, c_int, _f32p]
        L.dmmo_iou_from_counts.restype = None
        L.dmmo_cosine.argtypes = [_f32p, _f32p, c_int, c_int, c_int, _f32p]
        L.dmmo_cosine.restype = None
        L.dmmo_greedy_init.argtypes = [_f32p, c_int, c_int, _i32p]
        L.dmmo_greedy_init.restype = None
        L.dmmo_relax.argtypes = [_f32p, c_int, c_int, c_int, c_int, c_float, _vp, _vp, _vp, _vp, _vp]
        L.dmmo_relax.restype = c_int
        L.dmmo_matching_loss.argtypes = [_f32p, c_int, _f32p, c_int, c_int, _f32p, _vp, _vp]
        L.dmmo_matching_loss.restype = c_float
        L.dmmo_match_forward.argtypes = [_f32p, _f32p, _f32p, _f32p, _f32p, c_int, c_int, c_int, c_int,
                                         c_float, c_int, c_int, c_float, c_int] + [_vp] * 11
        L.dmmo_match_forward.restype = c_int
        L.dmmo_roialign4_mean.argtypes = [ctypes.c_void_p, c_int, c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                          _f32p, c_int, c_int, c_int, _f32p]
        L.dmmo_roialign4_mean.restype = None
        L.dmmo_paste_mask.argtypes = [_f32p, c_int, _f32p, c_int, c_int, c_float, c_int, _f32p, _f32p]
        L.dmmo_paste_mask.restype = None
        L.dmmo_nms.argtypes = [_f32p, _f32p, c_int, c_float, c_int, _i32p]
        L.dmmo_nms.restype = c_int
        L.dmmo_mask_box.argtypes = [_f32p, c_int, c_int, c_float, _f32p]
        L.dmmo_mask_box.restype = c_int
        L.dmmo_merge_labels.argtypes = [_f32p, c_int, c_int, np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")]
        L.dmmo_merge_labels.restype = None
        L.dmmo_check_div_by_const.argtypes = [c_int, ctypes.c_long]
        L.dmmo_check_div_by_const.restype = ctypes.c_long
        _lib = L
    return _lib


def _c(a, dt=np.float32):
    return np.ascontiguousarray(a, dtype=dt)


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def iou_counts(prop_mask, tplt_mask):
    """-> inter[M,N] i32, area_p[N] i32, area_t[M] i32 (match_helper.py:9-28 on all pairs)."""
    P, T = _c(prop_mask), _c(tplt_mask)
    N, M = P.shape[0], T.shape[0]
    HW = int(np.prod(P.shape[1:]))
    assert int(np.prod(T.shape[1:])) == HW
    inter = np.zeros((M, N), np.int32)
    ap, at = np.zeros(N, np.int32), np.zeros(M, np.int32)
    lib().dmmo_iou_counts(P.reshape(N, HW), N, T.reshape(M, HW), M, HW, inter, ap, at)
    return inter, ap, at


def iou_from_counts(inter, ap, at):
    M, N = inter.shape
    out = np.zeros((M, N), np.float32)
    lib().dmmo_iou_from_counts(_c(inter, np.int32), _c(ap, np.int32), _c(at, np.int32), N, M, out)
    return out


def cosine(tplt_feat, prop_feat):
    q, k = _c(tplt_feat), _c(prop_feat)
    M, D = q.shape
    N = k.shape[0]
    out = np.zeros((M, N), np.float32)
    lib().dmmo_cosine(q, k, M, N, D, out)
    return out


def greedy_init(C):
    C = _c(C)
    n, m = C.shape
    idx = np.zeros(n, np.int32)
    lib().dmmo_greedy_init(C, n, m, idx)
    return idx


def relax(C, max_iter, proj_iter, lr, want_xlist=False):
    """relax_matching (relax_match.py:36-105).  Returns dict(X, R, cost, iters, xlist, inner)."""
    C = _c(C)
    n, m = C.shape
    X = np.zeros((n, m), np.float32)
    R = np.zeros((n, m), np.float32)
    cost = np.zeros(max_iter + 1, np.float32)
    xl = np.zeros((max_iter + 1, n, m), np.float32) if want_xlist else None
    inner = np.zeros(max(max_iter, 1), np.int32)
    iters = lib().dmmo_relax(C, n, m, int(max_iter), int(proj_iter), float(lr),
                             _ptr(X), _ptr(R), _ptr(cost), _ptr(xl), _ptr(inner))
    return dict(X=X, R=R, cost=cost[:iters + 1], iters=iters,
                xlist=None if xl is None else xl[:iters + 1], inner=inner[:iters])


def matching_loss(prop_mask, targets, feature_sim):
    P, T = _c(prop_mask), _c(targets)
    N, M = P.shape[0], T.shape[0]
    HW = int(np.prod(P.shape[1:]))
    gi = np.zeros((M, N), np.float32)
    go = np.zeros((M, N), np.float32)
    loss = lib().dmmo_matching_loss(P.reshape(N, HW), N, T.reshape(M, HW), M, HW, _c(feature_sim),
                                    _ptr(gi), _ptr(go))
    return float(loss), gi, go


def hungarian(C):
    """hungarian_matching (relax_match.py:120-126): scipy linear_sum_assignment -> one-hot."""
    from scipy.optimize import linear_sum_assignment
    C = np.asarray(C)
    r, c = linear_sum_assignment(C)
    X = np.zeros_like(C, dtype=np.float32)
    X[r, c] = 1.0
    return X


def match_forward(prop_mask, tplt_mask, prop_feat, tplt_feat, prop_score, *, score_weight=0.3,
                  max_iter=20, proj_iter=5, lr=0.1, is_test=1, want_outmask=True):
    """MatchModel.forward for one frame, algo 'relax' (match_model.py:24-148)."""
    pm, tm = _c(prop_mask), _c(tplt_mask)
    P, O = pm.shape[0], tm.shape[0]
    H, W = pm.shape[1], pm.shape[2]
    HW = H * W
    pf, tf, sc = _c(prop_feat), _c(tplt_feat), _c(prop_score)
    D = pf.shape[1]
    Pp = max(P, O + 1)
    out = dict(
        full_outmask=np.zeros((O, H, W), np.float32) if want_outmask else None,
        match_score=np.zeros(O, np.float32), det_score=np.zeros(O, np.float32),
        sim=np.zeros((O, P), np.float32), R=np.zeros((O, Pp), np.float32),
        logic=np.zeros((O, Pp), np.float32), Rb=np.zeros((O, Pp), np.float32),
        cos=np.zeros((O, P), np.float32), inter=np.zeros((O, P), np.int32),
        area_p=np.zeros(P, np.int32), area_t=np.zeros(O, np.int32))
    iters = lib().dmmo_match_forward(
        pm.reshape(P, HW), tm.reshape(O, HW), pf, tf, sc, P, O, HW, D, float(score_weight),
        int(max_iter), int(proj_iter), float(lr), int(is_test),
        _ptr(out["full_outmask"]), _ptr(out["match_score"]), _ptr(out["det_score"]), _ptr(out["sim"]),
        _ptr(out["R"]), _ptr(out["logic"]), _ptr(out["Rb"]), _ptr(out["cos"]), _ptr(out["inter"]),
        _ptr(out["area_p"]), _ptr(out["area_t"]))
    out["iters"] = iters
    return out


def check_div_by_const(bmax=256, samples=100000):
    """Mismatches of the device's reciprocal-refinement division against IEEE division (expect 0)."""
    return int(lib().dmmo_check_div_by_const(int(bmax), int(samples)))


def roialign4_mean(feats, rois, scales=(0.25, 0.125, 0.0625, 0.03125), pooled=14, sampling=2):
    """Reference ROI feature extractor (feature_extractor.py:20-52) on 4 NCHW fp32 maps -> [R, 4*C].
    Literal legacy-ROIAlign restatement (third-party algorithm, parity un-pinned)."""
    feats = [_c(f) for f in feats]
    rois = _c(rois)
    B, C = feats[0].shape[0], feats[0].shape[1]
    R = rois.shape[0]
    ptrs = (ctypes.c_void_p * 4)(*[f.ctypes.data for f in feats])
    Hs = (ctypes.c_int * 4)(*[f.shape[2] for f in feats])
    Ws = (ctypes.c_int * 4)(*[f.shape[3] for f in feats])
    sc = (ctypes.c_float * 4)(*scales)
    out = np.zeros((R, 4 * C), np.float32)
    lib().dmmo_roialign4_mean(ptrs, B, C, Hs, Ws, sc, rois, R, int(pooled), int(sampling), out)
    return out


def paste_masks(prob, boxes, im_h, im_w, thresh=0.4, padding=1):
    """paste_mask_in_image + binmask_to_box (masker.py:110-173) for P proposals -> (masks [P,H,W], boxes [P,4])."""
    prob, boxes = _c(prob), _c(boxes)
    P, M = prob.shape[0], prob.shape[-1]
    masks = np.zeros((P, im_h, im_w), np.float32)
    nb = np.zeros((P, 4), np.float32)
    for p in range(P):
        lib().dmmo_paste_mask(prob[p].reshape(M, M), M, boxes[p], int(im_h), int(im_w), float(thresh), int(padding),
                              masks[p], nb[p])
    return masks, nb


def nms(boxes, scores, thresh, max_keep=0):
    """maskrcnn_benchmark-style greedy NMS (+1 areas) -> kept indices in descending score order."""
    boxes, scores = _c(boxes), _c(scores)
    n = boxes.shape[0]
    keep = np.zeros(max(n, 1), np.int32)
    cnt = lib().dmmo_nms(boxes.reshape(-1, 4), scores, n, float(thresh), int(max_keep), keep)
    return keep[:cnt]


def mask_boxes(masks, thresh=0.0):
    """ohw_mask2boxlist (utils.py:179-210) for planes [O,H,W] -> (boxes [O,4] f32, template_valid [O] i32)."""
    masks = _c(masks)
    O, H, W = masks.shape
    boxes = np.zeros((O, 4), np.float32)
    valid = np.zeros(O, np.int32)
    for o in range(O):
        valid[o] = lib().dmmo_mask_box(masks[o].reshape(-1), H, W, float(thresh), boxes[o])
    return boxes, valid


def merge_labels(masks, o_valid=None):
    """Label maps of evaluator.py:134-139 for masks [B,O,HW] (o_valid[b] live templates) -> uint8 [B,HW]."""
    masks = _c(masks)
    B, O, HW = masks.shape
    out = np.zeros((B, HW), np.uint8)
    for b in range(B):
        ob = O if o_valid is None else int(o_valid[b])
        lib().dmmo_merge_labels(np.ascontiguousarray(masks[b, :ob]).reshape(-1), ob, HW, out[b])
    return out
