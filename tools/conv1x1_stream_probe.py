#!/usr/bin/env python3
"""dmm_conv1x1_bf16: the streaming MFMA kernel (option CONV1X1_STREAM) against the library GEMM, shape by shape: max |diff|
against an fp32 reference of the same bf16 inputs, and us per call (20 launches per HIP-graph replay)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dmm_net_amd import _lib
dev = "cuda:0"
L = _lib.load()
g = torch.Generator(device=dev).manual_seed(0)
ws = torch.empty((64 << 20,), dtype=torch.uint8, device=dev)
def call(x, w, b, res, relu, y):
    rc = L.dmm_conv1x1_bf16(x.data_ptr(), w.data_ptr(), b.data_ptr(), None if res is None else res.data_ptr(), x.shape[0],
                            w.shape[0], w.shape[1], int(relu), y.data_ptr(), ws.data_ptr(), ws.numel(),
                            torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc
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
for images, hw in ((16, (64, 112)), (48, (64, 112)), (8, (64, 64))):
    rows = images * hw[0] * hw[1]
    for cin, cout, hwdiv in ((64, 64, 1), (64, 256, 1), (256, 64, 1), (256, 128, 1)):
        for res_on, relu in ((False, True), (True, True), (False, False)):
            x = (torch.randn((rows, cin), generator=g, device=dev)).to(torch.bfloat16)
            w = (torch.randn((cin, cout), generator=g, device=dev) / cin ** 0.5).to(torch.bfloat16)
            b = torch.randn((cout,), generator=g, device=dev)
            res = torch.randn((rows, cout), generator=g, device=dev).to(torch.bfloat16) if res_on else None
            ref = x[:65536].float() @ w.float() + b
            if res is not None: ref = ref + res[:65536].float()
            if relu: ref = ref.clamp_min(0)
            out = {}
            for v in (0, 1):
                with _lib.options(CONV1X1_STREAM=v):
                    y = torch.full((rows, cout), float("nan"), dtype=torch.bfloat16, device=dev)
                    call(x, w, b, res, relu, y)
                    torch.cuda.synchronize()
                    err = (y[:65536].float() - ref).abs().max().item()
                    tail_ok = bool(torch.isfinite(y.float()).all())
                    us = t_us(lambda: call(x, w, b, res, relu, y))
                    out[v] = (err, us, tail_ok, y)
            same = (out[0][3].float() - out[1][3].float()).abs().max().item()
            mb = (rows * (cin + cout * (2 if res_on else 1)) * 2) / 1e6
            print(f"{images:2d} img {cin:3d}->{cout:3d} res={int(res_on)} relu={int(relu)}: library {out[0][1]:6.1f} us (err {out[0][0]:.3g}) | "
                  f"stream {out[1][1]:6.1f} us (err {out[1][0]:.3g}, finite {out[1][2]}) | stream vs library max diff {same:.3g} | "
                  f"{mb:.0f} MB -> {mb / out[1][1]:.2f} TB/s", flush=True)
