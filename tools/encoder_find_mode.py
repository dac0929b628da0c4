#!/usr/bin/env python3
"""MIOpen find mode vs default kernel selection for the graphed inference encoder (BM=1 enables cudnn.benchmark;
run each setting in a fresh process)."""
import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
from dmm_net_amd.encoder import FeatureEncoder, GraphedEncoder, fold_batchnorm
dev = "cuda:0"
def timed(f, n=20, warm=5):
    for _ in range(warm): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
for bm in ((os.environ.get('BM', '0') == '1'),):
    torch.backends.cudnn.benchmark = bm
    for cl in (False,):
        torch.manual_seed(0)
        enc = fold_batchnorm(FeatureEncoder("resnet50").to(dev).eval())
        img = torch.randn(8, 3, 255, 255, device=dev)
        if cl:
            enc = enc.to(memory_format=torch.channels_last); img = img.contiguous(memory_format=torch.channels_last)
        t0 = time.time()
        g = GraphedEncoder(enc, weights_dtype=torch.bfloat16)
        t = timed(lambda: g(img))
        print(f"cudnn.benchmark={bm} channels_last={cl}: folded + bf16 weights + HIP graph {t:.3f} ms (setup {time.time()-t0:.1f} s)", flush=True)
