#!/usr/bin/env python3
"""Per-stage timing of the matching layer on the GPU (HIP events on torch's current stream).
usage: python tools/stage_timing.py [B ...]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dmm_net_amd import ops, synth

dev = "cuda:0"
CFG = int(os.environ.get("CFG", "2"))                  # CFG=5: N=200, M=20, fp16 masks
c = synth.CONFIGS[CFG]
N, M, H, W, D = c["P"], c["O"], c["H"], c["W"], c["D"]
MDT = torch.float16 if CFG == 5 else torch.float32
ES = 2 if CFG == 5 else 4


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3  # us


for B in [int(x) for x in sys.argv[1:]] or [1, 64, 256, 1024]:
    g = torch.Generator(device=dev).manual_seed(1)
    pm = torch.rand((B, N, H, W), generator=g, device=dev).to(MDT)
    tm = torch.rand((B, M, H, W), generator=g, device=dev).to(MDT)
    pf = torch.randn((B, N, D), generator=g, device=dev)
    tf = torch.randn((B, M, D), generator=g, device=dev)
    sc = torch.rand((B, N), generator=g, device=dev)
    inter, ap, at = ops.iou_counts(pm, tm)
    pn, tn = ops.feature_normalize(pf), ops.feature_normalize(tf)
    cos = ops.cosine(tn, pn)
    r = ops.relax_match(cos, inter, ap, at, sc, score_weight=0.3, max_iter=20, proj_iter=5, lr=0.1, is_test=1)
    C = -r["sim"]
    t_cost = timeit(lambda: ops.iou_counts(pm, tm))
    t_norm = timeit(lambda: (ops.feature_normalize(pf), ops.feature_normalize(tf)))
    t_cos = timeit(lambda: ops.cosine(tn, pn))
    t_relax = timeit(lambda: ops.relax_match(cos, inter, ap, at, sc, score_weight=0.3, max_iter=20, proj_iter=5,
                                             lr=0.1, is_test=1))
    t_relax0 = timeit(lambda: ops.relax_match(cos, inter, ap, at, sc, score_weight=0.3, max_iter=0, proj_iter=0,
                                              lr=0.1, is_test=1))
    t_solve = timeit(lambda: ops.relax_solve(C, 20, 5, 0.1))
    t_solve0 = timeit(lambda: ops.relax_solve(C, 0, 0, 0.1))
    t_mix = timeit(lambda: ops.mask_mix(r["Rb"], pm))
    src = pm[:, :M].contiguous()
    dst = torch.empty_like(src)
    t_copy = timeit(lambda: dst.copy_(src))
    print(f"      torch copy of {src.numel() * 4 / 1e9:.2f} GB: {t_copy:8.1f}us ({2 * src.numel() * 4 / t_copy / 1e3:6.0f} GB/s r+w)")
    del src, dst
    if os.environ.get("PACKED"):
        pp, pt = ops.pack_masks(pm), ops.pack_masks(tm)
        t_pack_t = timeit(lambda: ops.pack_masks(tm))
        t_pack_p = timeit(lambda: ops.pack_masks(pm))
        t_pc = timeit(lambda: ops.iou_counts_packed(pp, pt, H * W))
        print(f"      packed: pack templates {t_pack_t:7.1f}us, pack proposals {t_pack_p:7.1f}us "
              f"({B * N * H * W * ES / t_pack_p / 1e3:6.0f} GB/s), counts on packed planes {t_pc:7.1f}us")
        del pp, pt
    gb = B * (N + M) * H * W * ES / 1e9
    print(f"B={B:5d} cost {t_cost:8.1f}us ({gb / t_cost * 1e6:7.0f} GB/s)  norm {t_norm:6.1f} cos {t_cos:6.1f}  relax_match {t_relax:7.1f} "
          f"(iters=0: {t_relax0:6.1f})  solve-only {t_solve:7.1f} (init only {t_solve0:6.1f})  "
          f"mix {t_mix:7.1f} ({B * M * H * W * (4 + ES) / t_mix / 1e3:6.0f} GB/s)", flush=True)
    del pm, tm
    torch.cuda.empty_cache()
