#!/usr/bin/env python3
"""Where a wide-table forward spends its time: the stages of dmm_match_forward one by one (HIP events, B = 1 and 64)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dmm_net_amd import ops
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for (N, M) in [(300, 40), (50, 40)]:
    for B in (1, 64):
        pm = torch.rand((B, N, 96, 96), generator=g, device=dev); tm = torch.rand((B, M, 96, 96), generator=g, device=dev)
        pf = torch.randn((B, N, 512), generator=g, device=dev); tf = torch.randn((B, M, 512), generator=g, device=dev)
        sc = torch.rand((B, N), generator=g, device=dev)
        inter, ap, at = ops.iou_counts(pm, tm)
        pn, tn = ops.feature_normalize(pf), ops.feature_normalize(tf)
        cos = ops.cosine(tn, pn)
        kw = dict(score_weight=0.3, max_iter=20, proj_iter=5, lr=0.1, is_test=1)
        r = ops.relax_match(cos, inter, ap, at, sc, **kw)
        print(f"N={N} M={M} B={B}: counts {t(lambda: ops.iou_counts(pm, tm)):7.1f}  normalise x2 {t(lambda: (ops.feature_normalize(pf), ops.feature_normalize(tf))):7.1f}  "
              f"cosine {t(lambda: ops.cosine(tn, pn)):7.1f}  solver {t(lambda: ops.relax_match(cos, inter, ap, at, sc, **kw)):7.1f}  "
              f"mix {t(lambda: ops.mask_mix(r['Rb'], pm)):7.1f} us   whole {t(lambda: ops.match_forward(pm, tm, pf, tf, sc, **kw)):7.1f}", flush=True)
