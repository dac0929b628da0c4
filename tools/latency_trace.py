#!/usr/bin/env python3
"""B = 1 / B = 4 launch sequence of the layer (ForwardPlan in HIP-graph mode) repeated, for rocprofv3 --kernel-trace:
    rocprofv3 --kernel-trace --output-format csv -d DIR -- python tools/latency_trace.py 1"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dmm_net_amd import ops, synth

dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
c = synth.CONFIGS[2]
N, M, H, W, D = c["P"], c["O"], c["H"], c["W"], c["D"]
g = torch.Generator(device=dev).manual_seed(1)
pm = torch.rand((B, N, H, W), generator=g, device=dev)
tm = torch.rand((B, M, H, W), generator=g, device=dev)
pf = torch.randn((B, N, D), generator=g, device=dev)
tf = torch.randn((B, M, D), generator=g, device=dev)
sc = torch.rand((B, N), generator=g, device=dev)
plan = ops.ForwardPlan(B, N, M, H, W, D, dev, graph=True)
for _ in range(30):
    plan.run(pm, tm, pf, tf, sc, score_weight=0.3, max_iter=20, proj_iter=5, lr=0.1, is_test=1)
torch.cuda.synchronize()
print(plan.schedule_name())
