#!/usr/bin/env python3
"""The solver on B small frames: the dense launch (every frame the same template count, exact-row kernels) beside the
ragged launch (m_valid given, per-frame switch into the exact-row bodies), forward and backward, HIP-event us per launch.

    python tools/solver_ragged_probe.py [B ...]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from dmm_net_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
N, M, H, W, D = 50, 5, 255, 448, 512
g = torch.Generator(device=dev).manual_seed(11)


def us(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return round(a.elapsed_time(b) / n * 1e3, 1)


out = {}
for B in [int(v) for v in sys.argv[1:]] or [1, 4]:
    cos = torch.randn((B, M, N), generator=g, device=dev) * 0.05
    inter = torch.randint(20000, 40000, (B, M, N), generator=g, device=dev, dtype=torch.int32)
    ap = torch.randint(50000, 60000, (B, N), generator=g, device=dev, dtype=torch.int32)
    at = torch.randint(50000, 60000, (B, M), generator=g, device=dev, dtype=torch.int32)
    sc = torch.rand((B, N), generator=g, device=dev)
    kw = dict(score_weight=0.3, max_iter=10, proj_iter=5, lr=0.1, is_test=0)
    mv = torch.full((B,), M, dtype=torch.int32, device=dev)
    nv = torch.full((B,), N, dtype=torch.int32, device=dev)
    r = ops.relax_match(cos, inter, ap, at, sc, **kw)
    r2 = ops.relax_match(cos, inter, ap, at, sc, n_valid=nv, m_valid=mv, **kw)
    assert torch.equal(r["Rb"], r2["Rb"]) and torch.equal(r["iters"], r2["iters"])
    dRb = torch.rand(r["Rb"].shape, generator=g, device=dev)
    dms = torch.rand((B, M), generator=g, device=dev)
    kb = dict(max_iter=10, proj_iter=5, lr=0.1, is_test=0)
    d1 = ops.relax_match_bwd(r["sim"], sc, dRb, dms, None, **kb)
    d2 = ops.relax_match_bwd(r["sim"], sc, dRb, dms, None, n_valid=nv, m_valid=mv, **kb)
    assert torch.equal(d1, d2)
    out[B] = {"iters": r["iters"].flatten().tolist()[:8],
              "fwd_dense_us": us(lambda: ops.relax_match(cos, inter, ap, at, sc, **kw)),
              "fwd_ragged_us": us(lambda: ops.relax_match(cos, inter, ap, at, sc, n_valid=nv, m_valid=mv, **kw)),
              "fwd_mvalid_only_us": us(lambda: ops.relax_match(cos, inter, ap, at, sc, m_valid=mv, **kw)),
              "bwd_dense_us": us(lambda: ops.relax_match_bwd(r["sim"], sc, dRb, dms, None, **kb)),
              "bwd_ragged_us": us(lambda: ops.relax_match_bwd(r["sim"], sc, dRb, dms, None, n_valid=nv, m_valid=mv, **kb)),
              "bwd_nvalid_only_us": us(lambda: ops.relax_match_bwd(r["sim"], sc, dRb, dms, None, n_valid=nv, **kb))}
print(json.dumps(out, indent=1))
