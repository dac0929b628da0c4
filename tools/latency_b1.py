#!/usr/bin/env python3
"""Wall-clock latency of one matching-layer call at small batch (the product's per-frame use): MatchModel.forward
(autograd path), ForwardPlan eager (one fused C call) and ForwardPlan replayed from a captured HIP graph."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dmm_net_amd import ops, synth
from dmm_net_amd.match_model import MatchModel

dev = "cuda:0"
c = synth.CONFIGS[2]
cfg = {"matching": {"algo": "relax"}, "relax_max_iter": 20, "relax_proj_iter": 5, "relax_learning_rate": 0.1,
       "score_weight": 0.3}


def wall(fn, n=200, warm=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for B in (1, 4):
    g = torch.Generator(device=dev).manual_seed(0)
    pm = torch.rand((B, c["P"], c["H"], c["W"]), generator=g, device=dev)
    tm = torch.rand((B, c["O"], c["H"], c["W"]), generator=g, device=dev)
    pf = torch.randn((B, c["P"], c["D"]), generator=g, device=dev)
    tf = torch.randn((B, c["O"], c["D"]), generator=g, device=dev)
    sc = torch.rand((B, c["P"]), generator=g, device=dev)
    model = MatchModel(cfg, 1)
    with torch.no_grad():
        t_mod = wall(lambda: [model(pf[b], pm[b], [tf[b]], tm[b], sc[b]) for b in range(B)])
    plan = ops.ForwardPlan(B, c["P"], c["O"], c["H"], c["W"], c["D"], dev, pipeline=False)
    t_plan = wall(lambda: plan.run(pm, tm, pf, tf, sc))
    graph = torch.cuda.CUDAGraph()
    plan.run(pm, tm, pf, tf, sc)
    torch.cuda.synchronize()
    with torch.cuda.graph(graph):
        plan.run(pm, tm, pf, tf, sc)
    t_graph = wall(graph.replay)
    print(f"B={B}: MatchModel.forward x{B} {t_mod:7.1f} us | ForwardPlan eager {t_plan:7.1f} us | HIP graph replay {t_graph:7.1f} us")
