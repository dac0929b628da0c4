#!/usr/bin/env python3
"""Replay the config-3 encoder (FastEncoder in a HIP graph) 50 times for rocprofv3 --kernel-trace --stats; with
`eager` as argument run it un-graphed under torch.profiler and print the kernel table (launches, us per forward)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dmm_net_amd.encoder import FastEncoder, FeatureEncoder, GraphedEncoder, fold_batchnorm

dev = "cuda:0"
if os.environ.get("BENCH"):
    torch.backends.cudnn.benchmark = True      # MIOpen "find": time every applicable solver once per shape
torch.manual_seed(0)
enc = FastEncoder(fold_batchnorm(FeatureEncoder("resnet50").to(dev).eval()))
img = torch.randn(*[int(v) for v in os.environ.get("SHAPE", "8,3,255,255").split(",")], device=dev)
if len(sys.argv) > 1 and sys.argv[1] == "eager":
    from torch.profiler import ProfilerActivity, profile
    for _ in range(5):
        enc(img)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(10):
            enc(img)
        torch.cuda.synchronize()
    kern = [(e.key, e.count // 10, e.self_device_time_total / 10.0) for e in prof.key_averages() if e.self_device_time_total > 0]
    kern.sort(key=lambda r: -r[2])
    print("| kernel | launches | us per forward |\n|---|---|---|")
    for k, c, t in kern[:int(os.environ.get('ROWS', '40'))]:
        print(f"| {k[:120]} | {c} | {t:.1f} |")
    print(f"\ntotal: {sum(t for _, _, t in kern):.1f} us in {sum(c for _, c, _ in kern)} launches per forward")
else:
    g = GraphedEncoder(enc)
    for _ in range(50):
        g(img)
    torch.cuda.synchronize()
