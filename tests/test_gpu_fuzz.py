"""Random-shape parity fuzz of the forward layer against the C oracle: the other GPU tests hold the fixed cases, this one
walks ragged batches of arbitrary small shapes -- solver widths 2..64 and 1..16 rows (every shape of ATen's inner sum),
the small-batch count kernel on fp32 / fp16 / bf16 planes, the lanes / tile / three-launch feature similarity, the per-frame exact solver bodies of
ragged batches.  8 s by default; DMM_FUZZ_SECONDS / DMM_FUZZ_SEED for the long run."""
import os
import time

import numpy as np
import pytest
import torch

import oracle
from dmm_net_amd import ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_random_ragged_batches_match_the_oracle_frame_by_frame():
    budget = float(os.environ.get("DMM_FUZZ_SECONDS", "8"))
    rng = np.random.Generator(np.random.PCG64(int(os.environ.get("DMM_FUZZ_SEED", "1"))))
    g = lambda a: torch.from_numpy(a).to(DEV)
    t0, cases, frames = time.time(), 0, 0
    while time.time() - t0 < budget or cases < 12:
        B = int(rng.integers(1, 10))
        N, M = int(rng.integers(1, 65)), int(rng.integers(1, 17))
        H, W = int(rng.integers(1, 70)), int(rng.integers(1, 70))
        D = int(rng.choice([64, 128, 256, 512]))
        mi, pi = int(rng.integers(0, 25)), int(rng.integers(1, 6))
        is_test = int(rng.integers(0, 2))
        ragged = bool(rng.integers(0, 2))
        mdt = [torch.float32, torch.float32, torch.float16, torch.bfloat16][int(rng.integers(0, 4))]
        # the oracle sees the values the planes hold after rounding to their storage type (16-bit -> fp32 is exact)
        pm = torch.from_numpy(rng.random((B, N, H, W), dtype=np.float32)).to(mdt).float().numpy()
        tm = torch.from_numpy(rng.random((B, M, H, W), dtype=np.float32)).to(mdt).float().numpy()
        pf = rng.standard_normal((B, N, D), dtype=np.float32)
        tf = rng.standard_normal((B, M, D), dtype=np.float32)
        sc = rng.random((B, N), dtype=np.float32)
        nv = rng.integers(0, N + 1, B) if ragged else np.full(B, N)
        mv = rng.integers(0, M + 1, B) if ragged else np.full(B, M)
        plan = ops.ForwardPlan(B, N, M, H, W, D, DEV, mask_dtype=mdt, want_tables=True, pipeline=False)
        kw = dict(score_weight=0.3, max_iter=mi, proj_iter=pi, lr=0.1, is_test=is_test)
        if ragged:
            kw.update(n_valid=g(nv.astype(np.int32)), m_valid=g(mv.astype(np.int32)))
        full, ms, ds = plan.run(g(pm).to(mdt), g(tm).to(mdt), g(pf), g(tf), g(sc), **kw)
        full, ms, ds, iters = full.cpu().numpy(), ms.cpu().numpy(), ds.cpu().numpy(), plan.iters.cpu().numpy()
        for b in range(B):
            n, m = int(nv[b]), int(mv[b])
            tag = (B, N, M, H, W, D, mi, pi, is_test, ragged, str(mdt), b, n, m)
            if n == 0 or m == 0:
                assert not full[b].any() and not ms[b].any() and not ds[b].any(), ("dead frame", tag)
                continue
            o = oracle.match_forward(pm[b, :n], tm[b, :m], pf[b, :n], tf[b, :m], sc[b, :n], score_weight=0.3,
                                     max_iter=mi, proj_iter=pi, lr=0.1, is_test=is_test)
            assert int(iters[b]) == o["iters"], ("iters", tag)
            assert np.array_equal(ms[b, :m], o["match_score"]), ("match_score", tag)
            assert np.array_equal(ds[b, :m], o["det_score"]), ("det_score", tag)
            if is_test:
                assert np.array_equal(full[b, :m], o["full_outmask"]), ("full_outmask", tag)
            else:
                assert np.abs(full[b, :m] - o["full_outmask"]).max() <= 1e-5, ("full_outmask", tag)
            assert not full[b, m:].any(), ("rows beyond the live templates", tag)
            frames += 1
        cases += 1
    assert frames > 20
    print(f"fuzz: {cases} random batches, {frames} live frames checked against the oracle in {time.time() - t0:.0f} s")
