#!/usr/bin/env python3
"""Timing experiments on the one-wave solver: us per solve with parts of the sweep compiled out (libdmm_dbg<mask>.so,
-DDMM_DBG=<mask>; results are wrong by construction).  mask bits: 1 no row sums, 2 no column-sum chains, 4 no 8-lane
sequential combine, 8 no sweep exit test, 16 no cost exit test, 32 no cost norm."""
import glob
import os
import subprocess
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import torch
    from dmm_net_amd import ops
    g = torch.Generator(device="cuda:0").manual_seed(0)
    out = []
    for (n, m, it, pj) in [(10, 50, 20, 5), (5, 50, 40, 5), (5, 50, 20, 5), (16, 64, 20, 5)]:
        C = -torch.rand((1, n, m), generator=g, device="cuda:0")
        for _ in range(5):
            r = ops.relax_solve(C, it, pj, 0.1)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(50):
            ops.relax_solve(C, it, pj, 0.1)
        b.record()
        torch.cuda.synchronize()
        out.append(f"{n}x{m} {it}x{pj}: {a.elapsed_time(b) / 50 * 1e3:6.1f} us (iters {int(r['iters'][0])})")
    print(" | ".join(out))
    sys.exit(0)
libs = [None] + sorted(glob.glob(os.path.join(ROOT, "dmm_net_amd", "libdmm_dbg*.so")), key=lambda p: int(p.split("dbg")[-1][:-3]))
for lib in libs:
    env = dict(os.environ)
    if lib:
        env["DMM_LIB_PATH"] = lib
    r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
    tag = "product" if lib is None else "DBG=" + lib.split("dbg")[-1][:-3]
    print(f"{tag:10s} {r.stdout.strip() or r.stderr.strip()[-300:]}", flush=True)
