#!/usr/bin/env python3
"""BASELINE config 5 on one MI355X: N=200 proposals x M=20 templates, 255x255, fp16 mask planes (fp32 accumulate),
20 x 5 solver iterations, forward is_test=1; whole layer through ops.ForwardPlan, single-stream and 2-lane."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dmm_net_amd import ops, synth

dev = "cuda:0"
c = synth.CONFIGS[5]
N, M, H, W, D = c["P"], c["O"], c["H"], c["W"], c["D"]
B = int(os.environ.get("B", "256"))
g = torch.Generator(device=dev).manual_seed(5)
pm = torch.rand((B, N, H, W), generator=g, device=dev).half()
tm = torch.rand((B, M, H, W), generator=g, device=dev).half()
pf = torch.randn((B, N, D), generator=g, device=dev)
tf = torch.randn((B, M, D), generator=g, device=dev)
sc = torch.rand((B, N), generator=g, device=dev)
for pipe, pad in ((False, 0), (True, 0)):
    plan = ops.ForwardPlan(B, N, M, H, W, D, dev, mask_dtype=torch.float16, pipeline=pipe)
    run = lambda: plan.run(pm, tm, pf, tf, sc, score_weight=0.3, max_iter=20, proj_iter=5, lr=0.1, is_test=1)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        run()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    byt = B * ((N + M) * H * W * 2 + 2 * M * H * W * 3)          # planes read once (fp16) + selected read (fp16) + fp32 out
    print(f"config 5 B={B} {'2-lane' if pipe else 'single stream'} tl LDS pad {pad}: {ms:.3f} ms per step = {B / ms * 1e3:.0f} frames/s "
          f"(mean iters {float(plan.iters.float().mean()):.2f}; ~{byt / ms / 1e9:.2f} TB/s of algorithmic layer bytes)")
