#!/usr/bin/env python3
"""MIOpen's fused convolution + bias + ReLU (torch.miopen_convolution_relu) against what FastEncoder does for a k x k
convolution today (F.conv2d without bias, then dmm_bias_act_bf16 in place): us per call, 20 calls per HIP-graph replay,
the encoder's 3x3 / 7x7 shapes at 16 and 48 images of 255 x 448; max |difference| of the two results."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from dmm_net_amd.encoder import _bias_act_
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
def t_us(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(20): fn()
    gr.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): gr.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / 100 * 1e3
shapes = [(256, 32, 3, 1, 64, 112), (512, 64, 3, 1, 32, 56), (1024, 128, 3, 1, 16, 28), (2048, 128, 3, 1, 8, 14),
          (64, 64, 3, 1, 64, 112), (128, 128, 3, 1, 32, 56), (256, 256, 3, 1, 16, 28), (512, 512, 3, 1, 8, 14),
          (128, 128, 3, 2, 64, 112), (256, 256, 3, 2, 32, 56), (512, 512, 3, 2, 16, 28), (3, 64, 7, 2, 255, 448)]
for images in (16, 48):
    tot_a = tot_b = 0.0
    for cin, cout, k, st, h, w in shapes:
        x = torch.randn((images, cin, h, w), generator=g, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        wt = (torch.randn((cout, cin, k, k), generator=g, device=dev) / (cin * k * k) ** 0.5).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        b32 = torch.randn((cout,), generator=g, device=dev)
        bb = b32.to(torch.bfloat16)
        pad = k // 2
        cur = lambda: _bias_act_(F.conv2d(x, wt, None, (st, st), (pad, pad)), b32, None, True)
        try:
            fus = lambda: torch.miopen_convolution_relu(x, wt, bb, (st, st), (pad, pad), (1, 1), 1)
            ya, yb = cur(), fus()
            torch.cuda.synchronize()
            d = (ya.float() - yb.float()).abs().max().item()
            ta, tb = t_us(cur), t_us(fus)
        except Exception as e:                                   # noqa: BLE001
            print(f"{images} img {cin}->{cout} {k}x{k}/{st} @{h}x{w}: fused op failed: {str(e)[:120]}", flush=True)
            continue
        tot_a += ta; tot_b += tb
        print(f"{images} img {cin:4d}->{cout:3d} {k}x{k}/{st} @{h}x{w}: conv2d + bias_act {ta:7.1f} us | miopen_convolution_relu {tb:7.1f} us | "
              f"max diff {d:.3g} (layout out: {'NHWC' if yb.is_contiguous(memory_format=torch.channels_last) else 'other'})", flush=True)
    print(f"{images} img: sum {tot_a:.0f} -> {tot_b:.0f} us", flush=True)
