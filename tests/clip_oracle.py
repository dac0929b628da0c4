"""A T-frame clip of the evaluator's frame loop, computed on the CPU from the oracle's pieces -- TEST INFRASTRUCTURE.

What the reference does per frame around the matching layer (none of it importable as a whole: ``Evaler.__init__`` opens
dataset files, evaluator.py:55-59), restated as a numpy chain over ``oracle.*``:

  frame 0   templates from the annotation (``forward_timestep_init``, evaluator.py:215-225): ``ohw_mask2boxlist`` ->
            boxes + ``template_valid``; template features = ROI features of those boxes on frame 0's maps, fixed for the
            clip (``fill_template_dict``, dmm_model.py:22-46); ``outs`` of frame 0 IS the annotation (:119-121)
  frame t   proposals: paste every raw 28x28 mask, re-box tightly (``forward_mask_prop``, masker.py:27-50,110-173), NMS +
            top-k on the tight boxes (``filter_results``, boxlist_ops.py:15-29; model_encoder.py:115-134)
            ROI features of the kept proposals on frame t's maps (feature_extractor.py:20-52)
            per video (dmm_model.py:62-82): O = #valid templates; O == 0 or an 'extra' frame -> zeros out, the history is
            carried over (:66-69); else ``MatchModel`` on the first O rows, template features through ``OF_matrix =
            diag(valid)[:O]`` (:151-157), results scattered back through its transpose (:78-80)
            without a decoder ``outs`` = the matched masks and ``mask_hist`` = ``out_mask_last`` (evaluator.py:131-134)
            label map over the first ``valid.sum()`` rows (evaluator.py:134-139)

Only tests import this file; it calls nothing but ``oracle`` and numpy.
"""
import numpy as np

import oracle


def run_clip(feats, first, raw, n_frames, *, max_iter=40, proj_iter=5, lr=0.1, score_weight=0.3, nms_thresh=0.4,
             max_proposals=50, mask_thresh=0.4, padding=1, roi_fn=None):
    """feats[t]: 4 arrays [B,C,h_l,w_l] (the encoder's backbone features of frame t, strides 4..32);
    first [B,O,H,W] frame-0 annotation; raw[b][t] = (prob [R,M,M], boxes [R,4], scores [R]) (the last entry is reused for
    missing frames, evaluator.py:101-106); n_frames[b] = real length of video b.
    ``roi_fn(t, rois [R,5]) -> [R, D]``: where the ROI features come from; default the oracle's own ROIAlign on
    ``feats[t]``.  (relax_matching leaves its loops on EXACT fp32 equalities, relax_match.py:88-89,96-98: a 1e-7 difference
    between two correct ROIAlign implementations can move an exit by dozens of steps, and the layer returns the MEAN of the
    iterates -- so a test that wants equal iteration counts in every frame hands the SAME feature rows to both sides.)
    Returns hist [T,B,O,H,W] fp32 (``outs`` per frame), labels [T,B,H,W] uint8 (meaningful where t < n_frames[b]),
    iters [T,B] int32 (solver iterations executed; -1 where the layer did not run), kept [T,B] proposals after NMS."""
    T = len(feats)
    B, O, H, W = first.shape
    if roi_fn is None:
        roi_fn = lambda t, rois: oracle.roialign4_mean([f for f in feats[t]], rois)
    hist_out = np.zeros((T, B, O, H, W), np.float32)
    labels = np.zeros((T, B, H, W), np.uint8)
    iters = -np.ones((T, B), np.int32)
    kept = np.zeros((T, B), np.int32)
    for b in range(B):
        y0 = np.ascontiguousarray(first[b], np.float32)
        tboxes, valid = oracle.mask_boxes(y0, 0.0)                      # utils.py:179-210
        n_live = int(valid.sum())
        rois = np.concatenate([np.full((O, 1), b, np.float32), tboxes], 1)
        tfeat = roi_fn(0, rois)                                         # [O, D], fixed from frame 0
        tfv = tfeat[:n_live] * valid[:n_live, None].astype(np.float32)  # OF_matrix @ feat: rows i < O scaled by valid[i]
        mask_hist = y0.copy()
        for t in range(T):
            if t == 0:
                outs = y0
            elif n_live == 0 or n_frames[b] <= t:
                outs = np.zeros((O, H, W), np.float32)                  # dmm_model.py:66-69; mask_hist carried over
            else:
                prob, boxes, scores = raw[b][t] if len(raw[b]) > t else raw[b][-1]
                planes, tight = oracle.paste_masks(prob, boxes, H, W, mask_thresh, padding)
                keep = oracle.nms(tight, scores, nms_thresh, max_proposals)
                kept[t, b] = len(keep)
                pm, sc = planes[keep], np.ascontiguousarray(scores[keep], np.float32)
                prois = np.concatenate([np.full((len(keep), 1), b, np.float32), tight[keep]], 1)
                pfeat = roi_fn(t, prois)
                o = oracle.match_forward(pm, mask_hist[:n_live], pfeat, tfv, sc, score_weight=score_weight,
                                         max_iter=max_iter, proj_iter=proj_iter, lr=lr, is_test=1)
                iters[t, b] = o["iters"]
                outs = np.zeros((O, H, W), np.float32)
                outs[:n_live] = o["full_outmask"] * valid[:n_live, None, None].astype(np.float32)   # FO_matrix @ full
                mask_hist = outs
            hist_out[t, b] = outs
            if t < n_frames[b]:
                labels[t, b] = oracle.merge_labels(outs.reshape(1, O, H * W), [n_live]).reshape(H, W)
    return hist_out, labels, iters, kept
