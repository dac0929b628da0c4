#!/usr/bin/env python3
"""Inference frame loop (video.FrameLoop) wall time per frame on one MI355X: B videos, 255x448 frames, ResNet-50
encoder (BatchNorm folded + HIP graph), raw 28x28 proposal masks pasted + NMS-filtered on the device, ROI features,
DMM_Model.inference, label merge.  The refine decoder is not part of this package (refine=None)."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dmm_net_amd import video
from dmm_net_amd.dmm_model import DMM_Model
from dmm_net_amd.encoder import FastEncoder, FeatureEncoder, GraphedEncoder, fold_batchnorm
from dmm_net_amd.proposals import SimpleBoxList
from dmm_net_amd.roi_features import FeatureExtractor

dev = "cuda:0"
B, T, O, H, W = int(os.environ.get("B", "4")), int(os.environ.get("T", "12")), 5, 255, 448
rng = np.random.default_rng(0)
cfgs = {"matching": {"algo": "relax"}, "relax_max_iter": 40, "relax_proj_iter": 5, "relax_learning_rate": 0.1,
        "score_weight": 0.3}
if os.environ.get("NOENC"):                                 # isolate this package's share: pooling-only stand-in encoder
    import torch.nn.functional as F

    def enc(x):
        g = x.mean(1, keepdim=True)
        lv = tuple(F.avg_pool2d(g, s, ceil_mode=True).expand(-1, 128, -1, -1).contiguous() for s in (4, 8, 16, 32))
        return {"backbone_feature": lv, "refine_input_feat": lv}
elif os.environ.get("NCHW"):                                # round-1 encoder form: NCHW, every convolution through MIOpen
    enc = GraphedEncoder(fold_batchnorm(FeatureEncoder("resnet50").to(dev).eval()), weights_dtype=torch.bfloat16)
else:                                                       # channels-last bf16, 1x1 convolutions on hipBLASLt, fused epilogues
    enc = GraphedEncoder(FastEncoder(fold_batchnorm(FeatureEncoder("resnet50").to(dev).eval())),
                         miopen_find=bool(os.environ.get("FIND")))


def raw(n):
    x1, y1 = rng.uniform(0, W - 40, n), rng.uniform(0, H - 40, n)
    boxes = np.stack([x1, y1, np.minimum(x1 + rng.uniform(10, 150, n), W - 1), np.minimum(y1 + rng.uniform(10, 100, n), H - 1)], 1)
    bl = SimpleBoxList(torch.from_numpy(boxes.astype(np.float32)), (W, H))
    bl.add_field("scores", torch.from_numpy(rng.random(n).astype(np.float32)))
    bl.add_field("mask", torch.from_numpy((rng.random((n, 1, 28, 28)) * 0.6 + 0.4).astype(np.float32)))
    return bl


frames = torch.randn(B, T, 3, H, W, device=dev)
props = [[raw(50).to(dev) for _ in range(T)] for _ in range(B)]
first = torch.zeros(B, O, H, W, device=dev)
for b in range(B):
    for o in range(3 + b % 3):
        y0, x0 = int(rng.integers(0, H - 60)), int(rng.integers(0, W - 60))
        first[b, o, y0:y0 + 50, x0:x0 + 55] = 1.0
first = first.view(B, O, H * W)
loop = video.FrameLoop(enc, DMM_Model(cfgs, is_test=1, feature_extractor=FeatureExtractor()), nms_thresh=0.4, max_proposals=50)
loop.encode_ahead = int(os.environ.get("AHEAD", "0"))            # 0 = by clip length
loop.encode_first = int(os.environ.get("FIRST", "0"))
loop.encoder_priority = int(os.environ.get("ENCPRIO", "0"))
T = int(os.environ.get("T", "12"))
loop.slots = os.environ.get("SLOTS", "1") != "0"             # fixed-slot frame step (two-phase paste, no host sync)
loop.graph = os.environ.get("GRAPH", "1") != "0"             # ... replayed from one HIP graph per frame
labels = []
loop.run(frames, first, props, on_labels=lambda b, t, lab: None)   # warm-up (graph capture, MIOpen find for every chunk shape)
torch.cuda.synchronize()
t0 = time.perf_counter()
loop.run(frames, first, props, on_labels=lambda b, t, lab: labels.append(lab))
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"FrameLoop B={B} videos x {T} frames: {dt / T * 1e3:.2f} ms per frame step = {B * T / dt:.0f} frames/s")
# clips back to back, the next clip's first encoder chunk issued under this clip's last steps (run(next_frames=...))
loop.run(frames, first, props, on_labels=lambda b, t, lab: None, next_frames=frames)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    loop.run(frames, first, props, on_labels=lambda b, t, lab: None, next_frames=frames)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
print(f"  back to back with the next clip prefetched: {dt / T * 1e3:.2f} ms per frame step = {B * T / dt:.0f} frames/s")
if os.environ.get("PROFILE"):
    import cProfile
    import pstats
    pr = cProfile.Profile()
    pr.enable()
    loop.run(frames, first, props, on_labels=lambda b, t, lab: None)
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats(os.environ.get("SORT", "cumulative")).print_stats(32)
if os.environ.get("KERNELS"):
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        loop.run(frames, first, props, on_labels=lambda b, t, lab: None)
        torch.cuda.synchronize()
    rows = [(e.key, e.self_device_time_total / T / 1e3, e.count / T) for e in prof.key_averages()]
    tot = sum(r[1] for r in rows)
    print(f"GPU kernel time per frame step: {tot:.3f} ms")
    for k, t, c in sorted(rows, key=lambda r: -r[1])[:16]:
        print(f"  {k[:100]:100s} {t:7.3f} ms/frame  x{c:.1f}")
