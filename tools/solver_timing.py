#!/usr/bin/env python3
"""Solver mappings side by side (dmm_solve.hip thread-per-column vs dmm_solve_rs.hip row-split): us per launch of
dmm_relax_solve_f32 on random costs, (max_iter, proj_iter) = (20, 5), for the shapes of configs 2 / 5 and the
product's 5-template frames."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dmm_net_amd import _lib, ops

dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)


def t_us(fn, n=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for (n, m) in [(10, 50), (5, 50), (20, 200), (32, 256), (16, 64)]:
    for B in (1, 4, 64, 256, 1024):
        C = -torch.rand((B, n, m), generator=g, device=dev)
        row = []
        res = []
        for k in (0, 1):
            with _lib.options(SOLVER_KERNEL=k):
                row.append(t_us(lambda: ops.relax_solve(C, 20, 5, 0.1)))
                res.append(ops.relax_solve(C, 20, 5, 0.1))
        same = all(torch.equal(res[0][key], res[1][key]) for key in ("X", "R", "iters", "cost"))
        print(f"solver {n:2d} x {m:3d}  B={B:5d}:  thread-per-column {row[0]:8.1f} us   row-split {row[1]:8.1f} us   "
              f"identical={same}", flush=True)
