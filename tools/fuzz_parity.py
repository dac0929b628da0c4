#!/usr/bin/env python3
"""Randomised differential run: the HIP layer (through ops.ForwardPlan and the ragged autograd path) against the C
oracle on many random shapes / iteration settings.  Not part of the test suite (needs minutes); prints a summary.

    python tools/fuzz_parity.py [cases] [seed]
"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import oracle
from dmm_net_amd import ops, synth
from dmm_net_amd.autograd import match_layer_batched

dev = "cuda:0"
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for k in range(cases):
    N, M = int(rng.integers(1, 90)), int(rng.integers(1, 14))
    if rng.random() < 0.15:
        N, M = int(rng.integers(100, 257)), int(rng.integers(10, 33))
    H, W = int(rng.integers(2, 72)), int(rng.integers(2, 72))
    D = int(rng.choice([8, 33, 64, 512]))
    is_test = int(rng.integers(0, 2))
    mi, pi = int(rng.choice([0, 1, 5, 20, 40])), int(rng.choice([1, 2, 5]))
    kind = "structured" if rng.random() < 0.5 else "uniform"
    B = int(rng.integers(1, 4))
    frs = [synth.make_frame(N, M, H, W, D, seed=int(rng.integers(1 << 30)), kind=kind) for _ in range(B)]
    nv = [int(rng.integers(1, N + 1)) for _ in range(B)]
    mv = [int(rng.integers(1, M + 1)) for _ in range(B)]
    if rng.random() < 0.5:
        nv, mv = [N] * B, [M] * B
    t = lambda name: torch.from_numpy(np.stack([getattr(f, name) for f in frs])).to(dev)
    pm, tm, pf, tf, sc = t("proposed_mask"), t("mask_last_occurence"), t("proposed_feature"), t("template_feature"), t("proposal_score")
    ragged = nv != [N] * B or mv != [M] * B
    nvt = torch.tensor(nv, dtype=torch.int32, device=dev) if ragged else None
    mvt = torch.tensor(mv, dtype=torch.int32, device=dev) if ragged else None
    with torch.no_grad():
        full, ms, ds, _, iters = match_layer_batched(pf, pm, tf, tm, sc, None, nvt, mvt, score_weight=0.3, max_iter=mi,
                                                     proj_iter=pi, lr=0.1, is_test=is_test)
    ok = True
    for b in range(B):
        f = frs[b]
        o = oracle.match_forward(f.proposed_mask[:nv[b]], f.mask_last_occurence[:mv[b]], f.proposed_feature[:nv[b]],
                                 f.template_feature[:mv[b]], f.proposal_score[:nv[b]], max_iter=mi, proj_iter=pi,
                                 is_test=is_test)
        g_full = full[b, :mv[b]].cpu().numpy()
        ok &= int(iters[b]) == o["iters"]
        ok &= np.array_equal(ms[b, :mv[b]].cpu().numpy(), o["match_score"])
        ok &= np.array_equal(ds[b, :mv[b]].cpu().numpy(), o["det_score"])
        if is_test:
            ok &= np.array_equal(g_full, o["full_outmask"])
        else:
            ok &= float(np.abs(g_full - o["full_outmask"]).max(initial=0.0)) <= 1e-5
        ok &= float(full[b, mv[b]:].abs().sum()) == 0.0
    if not ok:
        bad += 1
        print(f"MISMATCH case {k}: N={N} M={M} HxW={H}x{W} D={D} is_test={is_test} iters=({mi},{pi}) B={B} nv={nv} mv={mv} {kind}")
print(f"{cases} cases, {bad} mismatches")
