#!/usr/bin/env python3
"""The kernels of one fixed-slot frame step (video.StepPlan), each timed ALONE (back-to-back launches, HIP events):
B videos of 255x448, R raw proposals of 28x28 per video, K slots, O templates, eval solver setting 40 x 5."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dmm_net_amd import _lib, ops, proposals as prop
from dmm_net_amd.roi_features import roialign4_mean_into

dev = "cuda:0"
B, R, K, O, H, W, C = int(os.environ.get("B", "4")), 50, 50, 5, 255, 448, 128
rng = np.random.default_rng(0)
L = _lib.load()


def t_us(fn, n=200, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def raw(n):
    x1, y1 = rng.uniform(0, W - 40, n), rng.uniform(0, H - 40, n)
    boxes = np.stack([x1, y1, np.minimum(x1 + rng.uniform(10, 150, n), W - 1), np.minimum(y1 + rng.uniform(10, 100, n), H - 1)], 1)
    bl = prop.SimpleBoxList(torch.from_numpy(boxes.astype(np.float32)), (W, H))
    bl.add_field("scores", torch.from_numpy(rng.random(n).astype(np.float32)))
    bl.add_field("mask", torch.from_numpy((rng.random((n, 1, 28, 28)) * 0.6 + 0.4).astype(np.float32)))
    return bl


clip = prop.ClipProposals.from_boxlists([[raw(R)] for _ in range(B)], 1, H, W, dev)
slots = prop.ProposalSlots(B, K, H, W, R, dev)
step = torch.zeros(1, dtype=torch.int32, device=dev)
s = torch.cuda.current_stream().cuda_stream
sp = step.data_ptr()
feats = [torch.randn(2 * 4 * B, C, (H + k - 1) // k, (W + k - 1) // k, device=dev).to(torch.bfloat16)
         .contiguous(memory_format=torch.channels_last) for k in (4, 8, 16, 32)]
feat_p = torch.zeros(B * K, 4 * C, device=dev)
tplt = torch.randn(B, O, 4 * C, device=dev)
hist = (torch.rand(B, O, H, W, device=dev) > 0.7).float()
full = torch.zeros(B, O, H, W, device=dev)
out = (full, torch.zeros(B, O, device=dev), torch.zeros(B, O, device=dev), torch.zeros(B, dtype=torch.int32, device=dev))
mval = torch.tensor([3 + b % 3 for b in range(B)], dtype=torch.int32, device=dev)
commit = torch.ones(B, dtype=torch.int32, device=dev)
labels = torch.zeros(B, H * W, dtype=torch.uint8, device=dev)
Pp = ops.padded_width(K, O)
Rb = torch.zeros(B, O, Pp, device=dev)
packed_hist = ops.pack_masks(hist)
hist2 = hist.clone()
ws = torch.empty(int(L.dmm_workspace_bytes_packed(B, K, O, 4 * C, H * W)), dtype=torch.uint8, device=dev)
prop.prepare_slots(clip, slots, 0.4, 0.4, 1, step=step)
print("kept per video:", slots.count.tolist())
rows = {
    "proposal_boxes": lambda: L.dmm_proposal_boxes_f32(clip.prob.data_ptr(), clip.boxes.data_ptr(), clip.counts.data_ptr(), B, R, 28, H, W, 0.4, 1, sp, slots.tight.data_ptr(), s),
    "nms_slots": lambda: L.dmm_nms_slots_f32(slots.tight.data_ptr(), clip.scores.data_ptr(), clip.counts.data_ptr(), B, R, 0.4, K, sp, slots.keep.data_ptr(), slots.count.data_ptr(), s),
    "paste_kept": lambda: L.dmm_paste_kept_f32(clip.prob.data_ptr(), clip.boxes.data_ptr(), clip.scores.data_ptr(), slots.tight.data_ptr(), slots.keep.data_ptr(), slots.count.data_ptr(), B, R, 28, K, H, W, 1, sp, None, slots.planes.data_ptr(), H * W, slots.packed.data_ptr(), slots.boxes.data_ptr(), slots.scores.data_ptr(), slots.rois.data_ptr(), s),
    "roialign4_mean (nhwc bf16)": lambda: roialign4_mean_into(slots.rois, feats, feat_p),
    "match_forward_packed 40x5": lambda: ops.match_forward_packed(slots.planes, slots.packed, hist, feat_p.view(B, K, -1), tplt, slots.scores, slots.count, mval, score_weight=0.3, max_iter=40, proj_iter=5, lr=0.1, is_test=1, out=out),
    "match_forward_packed 0x0": lambda: ops.match_forward_packed(slots.planes, slots.packed, hist, feat_p.view(B, K, -1), tplt, slots.scores, slots.count, mval, score_weight=0.3, max_iter=0, proj_iter=0, lr=0.1, is_test=1, out=out),
    "paste_kept (1-bit planes only)": lambda: L.dmm_paste_kept_f32(clip.prob.data_ptr(), clip.boxes.data_ptr(), clip.scores.data_ptr(), slots.tight.data_ptr(), slots.keep.data_ptr(), slots.count.data_ptr(), B, R, 28, K, H, W, 1, sp, None, None, H * W, slots.packed.data_ptr(), slots.boxes.data_ptr(), slots.scores.data_ptr(), slots.rois.data_ptr(), s),
    "match_solve_packed 40x5": lambda: ops.match_solve_packed(slots.packed, packed_hist, feat_p.view(B, K, -1), tplt, slots.scores, slots.count, mval, H * W, score_weight=0.3, max_iter=40, proj_iter=5, lr=0.1, is_test=1, out=(Rb, out[1], out[2], out[3]), workspace=ws),
    "step_finish": lambda: L.dmm_step_finish_f32(Rb.data_ptr(), Pp, clip.prob.data_ptr(), clip.boxes.data_ptr(), slots.keep.data_ptr(), slots.count.data_ptr(), B, R, 28, K, O, H, W, 1, sp, mval.data_ptr(), commit.data_ptr(), mval.data_ptr(), full.data_ptr(), hist2.data_ptr(), packed_hist.data_ptr(), labels.data_ptr(), s),
    "commit_masks": lambda: L.dmm_commit_masks_f32(full.data_ptr(), hist.data_ptr(), commit.data_ptr(), B, O * H * W, s),
    "merge_labels": lambda: L.dmm_merge_labels_f32(full.data_ptr(), B, O, H * W, O * H * W, H * W, mval.data_ptr(), labels.data_ptr(), s),
    "step_advance": lambda: L.dmm_step_advance(step.data_ptr(), s) or step.zero_(),
}
for k, fn in rows.items():
    print(f"{k:32s} {t_us(fn):8.1f} us")
