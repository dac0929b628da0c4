#!/usr/bin/env python3
"""Go / no-go probe for serving the mix's re-read of the selected proposal planes from the Infinity Cache (VERDICT r2
item 7).  Config-2 frames (50 x 10, 255x255 fp32), slices of G frames: cost(slice) -> solver(slice) -> mix(slice), the
mix timed (HIP events) HOT (right behind its slice's cost + solver) and COLD (a 2 GB read in between evicts everything).
Run once with the product library (non-temporal plane loads) and once with --lib libdmm_cached.so (plain loads;
_lib.use_library() before the first load).
Under rocprofv3 --pmc FETCH_SIZE the per-dispatch bytes of mask_mix_rows_kernel are the second witness (even dispatches
hot, odd cold)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dmm_net_amd import _lib, ops

if "--lib" in sys.argv:                                       # an A/B build of the library, chosen explicitly
    _lib.use_library(sys.argv[sys.argv.index("--lib") + 1])
dev = "cuda:0"
B, N, M, H, W, D = 64, 50, 10, 255, 255, 512
G = int(os.environ.get("G", "8"))
g = torch.Generator(device=dev).manual_seed(0)
pm = torch.rand((B, N, H, W), generator=g, device=dev)
tm = torch.rand((B, M, H, W), generator=g, device=dev)
pf, tf = torch.randn((B, N, D), generator=g, device=dev), torch.randn((B, M, D), generator=g, device=dev)
sc = torch.rand((B, N), generator=g, device=dev)
flush = torch.empty(512 << 20, dtype=torch.float32, device=dev)       # 2 GB
kw = dict(score_weight=0.3, max_iter=20, proj_iter=5, lr=0.1, is_test=1)
cos = ops.cosine_features(tf, pf)
res = {"hot": [], "cold": []}
for rep in range(6):
    for s0 in range(0, B, G):
        for mode in ("hot", "cold"):
            sl = slice(s0, s0 + G)
            inter, ap, at = ops.iou_counts(pm[sl], tm[sl])
            r = ops.relax_match(cos[sl].contiguous(), inter, ap, at, sc[sl].contiguous(), **kw)
            if mode == "cold":
                flush.add_(1.0)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            ops.mask_mix(r["Rb"], pm[sl])
            b.record()
            torch.cuda.synchronize()
            if rep > 0:
                res[mode].append(a.elapsed_time(b) * 1e3)
alg = G * 2 * M * H * W * 4
for mode, v in res.items():
    v.sort()
    med = v[len(v) // 2]
    print(f"G={G} mix {mode:4s}: median {med:7.1f} us  min {v[0]:7.1f} us  ({alg / med / 1e3:6.0f} GB/s algorithmic r+w, "
          f"lib={_lib.LIB_PATH[-24:]})")
