#!/usr/bin/env python3
"""Fused 4-level ROIAlign+mean timing (config-3 maps: 8 frames, 128 channels, strides 4..32 of 255x255)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dmm_net_amd.roi_features import FeatureExtractor
from dmm_net_amd.proposals import SimpleBoxList

dev = "cuda:0"
B, C, H, W = 8, 128, 255, 255
g = torch.Generator(device=dev).manual_seed(0)
fe = FeatureExtractor()


def boxes(n, lo, hi):
    x1 = torch.rand(n, generator=g, device=dev) * (W - hi)
    y1 = torch.rand(n, generator=g, device=dev) * (H - hi)
    w = lo + torch.rand(n, generator=g, device=dev) * (hi - lo)
    h = lo + torch.rand(n, generator=g, device=dev) * (hi - lo)
    return torch.stack([x1, y1, (x1 + w).clamp(max=W - 1), (y1 + h).clamp(max=H - 1)], 1)


def timed(fn, n=20, warm=5):
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


for dt in (torch.float32, torch.bfloat16):
    maps = tuple(torch.randn(B, C, (H + s - 1) // s, (W + s - 1) // s, device=dev, generator=g).to(dt) for s in (4, 8, 16, 32))
    for per, lo, hi in ((50, 8, 108), (10, 8, 108), (50, 8, 24), (50, 100, 200)):
        bl = [SimpleBoxList(boxes(per, lo, hi), (W, H)) for _ in range(B)]
        t = timed(lambda: fe(maps, bl))
        print(f"{str(dt):15} {B * per:4d} rois, sizes {lo}-{hi} px: {t:8.1f} us")
