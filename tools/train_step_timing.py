#!/usr/bin/env python3
"""Training-mode matching layer (BASELINE config 4 per-GPU shape: 4 videos x 3 frames = 12 layer calls batched,
50 proposals, 5 templates, 255x448, targets present): forward and forward+backward wall / GPU time per step,
plus a torch profiler table of the kernels of one step."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time
import torch
from dmm_net_amd.autograd import match_layer_batched

dev = "cuda:0"
B, N, M, H, W, D = 12, 50, 5, 255, 448, 512
g = torch.Generator(device=dev).manual_seed(0)
pm = torch.rand((B, N, H, W), generator=g, device=dev)
tm = torch.rand((B, M, H, W), generator=g, device=dev)
tg = (torch.rand((B, M, H, W), generator=g, device=dev) > 0.5).float()
pf = torch.randn((B, N, D), generator=g, device=dev, requires_grad=True)
tf = torch.randn((B, M, D), generator=g, device=dev, requires_grad=True)
sc = torch.rand((B, N), generator=g, device=dev)
nv = torch.full((B,), N, dtype=torch.int32, device=dev)
mv = torch.full((B,), M, dtype=torch.int32, device=dev)
kw = dict(score_weight=0.3, max_iter=20, proj_iter=5, lr=0.1, is_test=0)


def fwd():
    return match_layer_batched(pf, pm, tf, tm, sc, tg, nv, mv, **kw)


def step():
    full, ms, ds, loss, _ = fwd()
    (full.sum() * 1e-3 + loss.sum()).backward()
    pf.grad = None
    tf.grad = None


def timed(fn, n=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, a.elapsed_time(b) / n


with torch.no_grad():
    w, gt = timed(fwd)
print(f"forward only      : wall {w:.3f} ms, GPU {gt:.3f} ms per step of {B} frames")
w, gt = timed(step)
print(f"forward + backward: wall {w:.3f} ms, GPU {gt:.3f} ms per step of {B} frames")
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(5):
        step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=14, max_name_column_width=60))
