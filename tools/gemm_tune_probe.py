#!/usr/bin/env python3
"""Does timing hipBLASLt's heuristic candidates per shape (option GEMM_TUNE = n) pay at the frame loop's shapes?
Every 1x1 convolution shape of the inference encoder at 16 and 48 images of 255 x 448, alone, 20 launches per graph replay:
us with the heuristic's first pick vs the fastest of its first n candidates."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn
from dmm_net_amd import _lib
from dmm_net_amd.encoder import FastEncoder, FeatureEncoder, fold_batchnorm
dev = "cuda:0"
torch.manual_seed(0)
folded = fold_batchnorm(FeatureEncoder("resnet50").to(dev).eval())
def table(B, tune):
    _lib.set_option("GEMM_TUNE", tune)
    fast = FastEncoder(folded)                       # fresh plans: the option is read when a shape is first planned
    img = torch.randn(B, 3, 255, 448, device=dev)
    shapes, hooks = {}, []
    def hook(m, inp, o):
        if m.kernel_size[0] == 1:
            shapes.setdefault((m.in_channels, m.out_channels, m.stride[0], tuple(inp[0].shape[2:])), [m, 0])[1] += 1
    for m in folded.modules():
        if isinstance(m, nn.Conv2d):
            hooks.append(m.register_forward_hook(hook))
    with torch.no_grad():
        folded(img)
    for h in hooks: h.remove()
    res = {}
    for key, (conv, count) in shapes.items():
        cin, cout, st, hw = key
        x = torch.randn(B, cin, *hw, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        fn = lambda: fast._conv(x, conv, relu=True)
        for _ in range(3): fn()                      # the first un-captured call times the candidates
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(20): fn()
        g.replay(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5): g.replay()
        b.record(); torch.cuda.synchronize()
        res[key] = (a.elapsed_time(b) / 100 * 1e3, count)
    return res
for B in (16, 48):
    base = table(B, 1)
    for tune in (8, 24):
        t = table(B, tune)
        tot0 = sum(v[0] * v[1] for v in base.values()); tot1 = sum(t[k][0] * t[k][1] for k in base)
        print(f"B={B} tune={tune}: sum over the forward's 1x1 convolutions {tot0:.0f} -> {tot1:.0f} us", flush=True)
        for k in sorted(base, key=lambda k: -base[k][0] * base[k][1])[:8]:
            print(f"    {k}: {base[k][0]:.1f} -> {t[k][0]:.1f} us x{base[k][1]}", flush=True)
