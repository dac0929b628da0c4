#!/usr/bin/env python3
"""Encoder (ResNet-50 + heads, 8 frames of 255x255) on MI355X: eager vs BatchNorm-folded vs folded + HIP graph,
fp32 / bf16, NCHW / channels_last."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dmm_net_amd.encoder import FeatureEncoder, GraphedEncoder, fold_batchnorm

dev = "cuda:0"


def timed(f, n=20, warm=5):
    for _ in range(warm):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


for cl in (False, True):
    for dt in (torch.float32, torch.bfloat16):
        torch.manual_seed(0)
        enc = FeatureEncoder("resnet50").to(dev).eval()
        img = torch.randn(8, 3, 255, 255, device=dev)
        if cl:
            enc = enc.to(memory_format=torch.channels_last)
            img = img.contiguous(memory_format=torch.channels_last)
        folded = fold_batchnorm(enc)
        if cl:
            folded = folded.to(memory_format=torch.channels_last)

        def run(m):
            with torch.no_grad(), torch.autocast("cuda", dtype=dt, enabled=dt != torch.float32):
                return m(img)
        t_e = timed(lambda: run(enc))
        t_f = timed(lambda: run(folded))
        g = GraphedEncoder(folded, autocast_dtype=None if dt == torch.float32 else dt)
        t_g = timed(lambda: g(img))
        msg = f"channels_last={cl!s:5} {str(dt):15}: eager {t_e:.3f} ms | BN folded {t_f:.3f} ms | folded + HIP graph {t_g:.3f} ms"
        if dt != torch.float32:
            import copy
            g2 = GraphedEncoder(copy.deepcopy(folded), weights_dtype=dt)
            t_w = timed(lambda: g2(img))
            msg += f" | folded + {str(dt).split('.')[-1]} weights + HIP graph {t_w:.3f} ms"
        print(msg)
