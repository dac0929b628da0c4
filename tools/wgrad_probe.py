#!/usr/bin/env python3
"""Weight-gradient kernels alone on the shapes of config 4's ResNet-101 (12 frames of 255 x 448): time per call of
dmm_wgrad_bf16 / dmm_wgrad3x3_bf16 (partial tables + ordered reduce), and the sum over one training step.

    python tools/wgrad_probe.py [--reps 50]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dmm_net_amd import _lib  # noqa: E402

# (name, rows (B, H, W of the OUTPUT for 1x1; input for 3x3), co, ci, kind, stride, count per step)
B = 12
SHAPES = [
    ("l1.conv1 64<-256", (B, 64, 112), 64, 256, 1, 1, 2), ("l1.conv2 3x3 64", (B, 64, 112), 64, 64, 3, 1, 3),
    ("l1.conv3 256<-64", (B, 64, 112), 256, 64, 1, 1, 4),
    ("l2.conv1 128<-512", (B, 32, 56), 128, 512, 1, 1, 3), ("l2.conv2 3x3 128", (B, 32, 56), 128, 128, 3, 1, 3),
    ("l2.conv3 512<-128", (B, 32, 56), 512, 128, 1, 1, 4),
    ("l3.conv1 256<-1024", (B, 16, 28), 256, 1024, 1, 1, 22), ("l3.conv2 3x3 256", (B, 16, 28), 256, 256, 3, 1, 22),
    ("l3.conv3 1024<-256", (B, 16, 28), 1024, 256, 1, 1, 23),
    ("l4.conv1 512<-2048", (B, 8, 14), 512, 2048, 1, 1, 2), ("l4.conv2 3x3 512", (B, 8, 14), 512, 512, 3, 1, 2),
    ("l4.conv3 2048<-512", (B, 8, 14), 2048, 512, 1, 1, 3),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=50)
    a = ap.parse_args()
    L = _lib.load()
    dev = torch.device("cuda:0")
    total = 0.0
    out = []
    for name, (b, h, w), co, ci, k, stride, count in SHAPES:
        rows = b * h * w
        dy = torch.randn(rows, co, device=dev).bfloat16()
        x = torch.randn(rows, ci, device=dev).bfloat16()
        cv = ci * (9 if k == 3 else 1)
        dw = torch.empty(co * cv, device=dev)
        nb = L.dmm_wgrad_workspace_bytes(rows, co, cv)
        ws = torch.empty(max(nb, 4) // 4, device=dev)

        def call():
            st = torch.cuda.current_stream().cuda_stream
            if k == 1:
                rc = L.dmm_wgrad_bf16(dy.data_ptr(), x.data_ptr(), rows, co, ci, co, ci, dw.data_ptr(), ws.data_ptr(), nb, st)
            else:
                rc = L.dmm_wgrad3x3_bf16(dy.data_ptr(), x.data_ptr(), b, h, w, ci, co, stride, dw.data_ptr(), ws.data_ptr(), nb,
                                         st)
            assert rc == 0, rc
        for _ in range(5):
            call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(a.reps):
                call()
        g.replay()
        torch.cuda.synchronize()
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / a.reps
        total += us * count
        out.append({"shape": name, "us": round(us, 2), "partial_MB": round(nb / 2 ** 20, 1), "count": count})
        print(json.dumps(out[-1]), flush=True)
    print(json.dumps({"step_total_ms": round(total / 1e3, 3)}), flush=True)


if __name__ == "__main__":
    main()
