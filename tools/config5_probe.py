#!/usr/bin/env python3
"""BASELINE configs[4] (200 proposals x 20 templates, fp16 planes at a line-aligned stride, 512 frames): ms per step of the
2-lane schedule (fp16-state solver beside the streaming kernels) and of the single-stream fp32-state schedule, against the
workgroup target of the template-lane count kernel (option COST_TL_WGS: few long-lived workgroups stream best ALONE; under the
2-lane schedule the solver's waves slow the CUs they land on and a statically partitioned launch waits for its slowest CU).

    python tools/config5_probe.py [wgs ...]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from dmm_net_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda", 0)
B, N, M, H, W, D = 512, 200, 20, 255, 255, 512
g = torch.Generator(device=dev).manual_seed(5)


def planes(k):
    t = ops.alloc_planes(B, k, H, W, torch.float16, dev, 128, fill=0)
    for b0 in range(0, B, 64):
        t[b0:b0 + 64].copy_(torch.rand((min(64, B - b0), k, H, W), generator=g, device=dev))
    return t


inputs = (planes(N), planes(M), torch.randn((B, N, D), generator=g, device=dev), torch.randn((B, M, D), generator=g, device=dev),
          torch.rand((B, N), generator=g, device=dev))
kw = dict(score_weight=0.3, max_iter=20, proj_iter=5, lr=0.1, is_test=1)


def ms(plan, n=20):
    for _ in range(3):
        plan.run(*inputs, **kw)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        plan.run(*inputs, **kw)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


out = {}
for wgs in [int(v) for v in sys.argv[1:]] or [512, 1024, 2048, 4096]:
    with _lib.options(COST_TL_WGS=wgs):
        row = {}
        for name, kwp in (("two_lane_f16_solver", dict(pipeline=True, solver_state="f16")),
                          ("single_stream_f32_solver", dict(pipeline=False, solver_state="f32")),
                          ("single_stream_f16_solver", dict(pipeline=False, solver_state="f16"))):
            plan = ops.ForwardPlan(B, N, M, H, W, D, dev, mask_dtype=torch.float16, out_dtype=torch.float16,
                                   out_plane_align=128, **kwp)
            t = ms(plan)
            row[name] = {"ms_per_step": round(t, 4), "frames_per_s": round(B / t * 1e3, 1),
                         "b_cost_frac": round((N + M) * H * W * 2 * B / (t * 1e-3) / 8e12, 4)}
            del plan
        out[str(wgs)] = row
print(json.dumps(out, indent=1))
