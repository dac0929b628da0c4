#!/usr/bin/env python3
"""Per-layer roofline table of the inference encoder (ResNet-50 + heads, bf16 channels-last, BatchNorm folded), VERDICT r3
item 6: every distinct convolution of FastEncoder timed alone (20 launches captured in a HIP graph, device-side time per
call, epilogue launch included) at SEVERAL batches, with BOTH ceilings per row:
  * MFMA: 2 x MACs / time vs the 2.5 PFLOP/s dense bf16 peak,
  * HBM: (input + output (+ residual) activations + weights) bytes / time vs 8 TB/s -- what "activation bound" would mean.
A row far below both is bound by neither: launch + fill + drain of a kernel that runs 5-25 us on a 256-CU device.

    python tools/encoder_layer_table2.py OUT.md [B,H,W ...]      default: 8,255,255  16,255,448  48,255,448
"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn
from dmm_net_amd.encoder import FastEncoder, FeatureEncoder, fold_batchnorm

dev = "cuda:0"
PEAK_TF, PEAK_GB = 2500.0, 8000.0
torch.manual_seed(0)
torch.backends.cudnn.benchmark = True
folded = fold_batchnorm(FeatureEncoder("resnet50").to(dev).eval())
fast = FastEncoder(folded)
cases = [tuple(int(v) for v in a.split(",")) for a in sys.argv[2:]] or [(8, 255, 255), (16, 255, 448), (48, 255, 448)]
out = ["# Encoder convolutions one by one: time, MFMA fraction AND achieved HBM GB/s (FastEncoder building blocks, bf16 channels-last)",
       "", "Each row: one distinct convolution shape, 20 launches captured in a HIP graph and replayed (device-side time per call, "
       "epilogue launch included).  bytes = input + output activations + weights (bf16); `HBM frac` = bytes / time / 8 TB/s, "
       "`MFMA frac` = 2 x MACs / time / 2.5 PFLOP/s.  A residual input (+1 output-sized read) is not counted: the rows are lower bounds."]
summary = []
for (B, H, W) in cases:
    img = torch.randn(B, 3, H, W, device=dev)
    shapes, hooks = {}, []

    def hook(m, inp, o):
        key = (m.in_channels, m.out_channels, m.kernel_size[0], m.stride[0], tuple(inp[0].shape[2:]))
        shapes.setdefault(key, [m, 0])[1] += 1
    for m in folded.modules():
        if isinstance(m, nn.Conv2d):
            hooks.append(m.register_forward_hook(hook))
    with torch.no_grad():
        folded(img)
    for h in hooks:
        h.remove()
    rows = []
    for (cin, cout, k, st, hw), (conv, count) in shapes.items():
        x = torch.randn(B, cin, *hw, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        fn = lambda: fast._conv(x, conv, relu=True)
        for _ in range(3):
            y = fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(20):
                y = fn()
        g.replay()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            g.replay()
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) / 100 * 1e3
        flops = 2.0 * y.numel() * cin * k * k
        nbytes = 2.0 * (x.numel() + y.numel() + cin * cout * k * k)
        rows.append((us * count, count, cin, cout, k, st, hw, us, flops, nbytes))
        del g
    rows.sort(reverse=True)
    tot_us = sum(r[0] for r in rows)
    tot_fl = sum(r[8] * r[1] for r in rows)
    tot_by = sum(r[9] * r[1] for r in rows)
    out += ["", f"## {B} images of {H} x {W}", "",
            "| Cin -> Cout | k / stride | input HxW | calls | us per call | GFLOP | TFLOP/s | MFMA frac | MB moved | GB/s | HBM frac | bound |",
            "|---|---|---|---|---|---|---|---|---|---|---|---|"]
    below = 0.0
    for tot, count, cin, cout, k, st, hw, us, fl, nb in rows:
        tf, gb = fl / (us * 1e-6) / 1e12, nb / (us * 1e-6) / 1e9
        fm, fh = tf / PEAK_TF, gb / PEAK_GB
        bound = "HBM" if fh >= 0.45 else ("MFMA" if fm >= 0.3 else "neither (launch / fill / drain)")
        if bound.startswith("neither"):
            below += tot
        out.append(f"| {cin} -> {cout} | {k}x{k} / {st} | {hw[0]}x{hw[1]} | {count} | {us:.1f} | {fl / 1e9:.2f} | {tf:.0f} | {fm:.3f} | "
                   f"{nb / 1e6:.1f} | {gb:.0f} | {fh:.3f} | {bound} |")
    line = (f"sum over the forward: {tot_us:.0f} us, {tot_fl / 1e9:.1f} GFLOP = {tot_fl / tot_us / 1e6:.0f} TFLOP/s "
            f"({tot_fl / tot_us / 1e6 / PEAK_TF:.3f} of the MFMA peak), {tot_by / 1e6:.0f} MB = {tot_by / tot_us / 1e3:.0f} GB/s "
            f"({tot_by / tot_us / 1e3 / PEAK_GB:.3f} of the HBM peak); {below / tot_us:.0%} of the time is in rows bound by neither "
            f"(HBM frac < 0.45 and MFMA frac < 0.3).")
    out += ["", line]
    summary.append(f"{B} x {H} x {W}: " + line)
    print(summary[-1], flush=True)
open(sys.argv[1], "w").write("\n".join(out) + "\n")
