#!/usr/bin/env python3
"""The general (any table size) forward, timed: dmm_match_forward outside the fast kernels' envelope, 20 x 5 iterations,
96 x 96 fp32 masks, D = 512.  Correctness path -- this is what it costs."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dmm_net_amd import ops

dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
for (N, M) in [(300, 40), (512, 64), (50, 40), (200, 20)]:
    for B in (1, 64):
        pm = torch.rand((B, N, 96, 96), generator=g, device=dev)
        tm = torch.rand((B, M, 96, 96), generator=g, device=dev)
        pf = torch.randn((B, N, 512), generator=g, device=dev)
        tf = torch.randn((B, M, 512), generator=g, device=dev)
        sc = torch.rand((B, N), generator=g, device=dev)
        kw = dict(score_weight=0.3, max_iter=20, proj_iter=5, lr=0.1, is_test=1)
        for _ in range(2):
            out = ops.match_forward(pm, tm, pf, tf, sc, **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            out = ops.match_forward(pm, tm, pf, tf, sc, **kw)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        wide = M > 32 or max(N, M + 1) > 256
        print(f"N={N:4d} M={M:3d} B={B:3d}: {dt * 1e3:8.3f} ms per call ({'general' if wide else 'fast'} kernels), "
              f"iters {int(out[3].float().mean())}")
