"""Seeded synthetic frames for the matching layer (build-owned generator).

Shapes follow the reference's call contract for ``MatchModel.forward``
(reference ``dmm/modules/match_model.py:24-47``): per frame

    proposed_mask        [P, H, W]  fp32 soft probabilities in [0, 1]
    mask_last_occurence  [O, H, W]  fp32 soft probabilities in [0, 1]
    proposed_feature     [P, D]     fp32
    template_feature     [O, D]     fp32
    proposal_score       [P]        fp32
    targets (optional)   [O, H, W]  fp32 binary

Two distributions:

* ``uniform``    -- masks ~ U[0,1): IoU ~ 1/3 everywhere; throughput-neutral worst
                    case for argmax stability.  Used by ``bench.py``.
* ``structured`` -- each proposal is a soft-edged random rectangle / ellipse
                    covering 2-30 % of the frame; each template is a distinct
                    proposal shifted by <= 8 px plus noise, so the assignment is
                    meaningful.  Used by the parity / argmax tests.

numpy's ``Generator(PCG64(seed))`` stream is stable across platforms, so the
golden fixtures under ``tests/golden`` store only the seed + a checksum of
the inputs for the large configurations.
"""
from __future__ import annotations

import hashlib
from dataclasses import dataclass
from typing import Optional

import numpy as np

# BASELINE.json "configs", index -> (P, O, H, W, D, max_iter, proj_iter, lr)
CONFIGS = {
    1: dict(P=8, O=3, H=64, W=64, D=512, max_iter=20, proj_iter=5, lr=0.1),
    2: dict(P=50, O=10, H=255, W=255, D=512, max_iter=20, proj_iter=5, lr=0.1),
    5: dict(P=200, O=20, H=255, W=255, D=512, max_iter=20, proj_iter=5, lr=0.1),
}
BASE_SEED = 1234


@dataclass
class Frame:
    proposed_mask: np.ndarray        # [P,H,W] f32
    mask_last_occurence: np.ndarray  # [O,H,W] f32
    proposed_feature: np.ndarray     # [P,D]   f32
    template_feature: np.ndarray     # [O,D]   f32
    proposal_score: np.ndarray       # [P]     f32
    targets: Optional[np.ndarray]    # [O,H,W] f32 (0/1) or None
    perm: Optional[np.ndarray]       # [O] planted proposal index per template (structured only)

    def checksum(self) -> str:
        h = hashlib.sha256()
        for a in (self.proposed_mask, self.mask_last_occurence, self.proposed_feature,
                  self.template_feature, self.proposal_score):
            h.update(np.ascontiguousarray(a).tobytes())
        if self.targets is not None:
            h.update(np.ascontiguousarray(self.targets).tobytes())
        return h.hexdigest()


def _soft_shape(rng: np.random.Generator, H: int, W: int) -> np.ndarray:
    """One soft-edged rectangle or ellipse covering 2-30 % of the frame."""
    frac = rng.uniform(0.02, 0.30)
    aspect = rng.uniform(0.5, 2.0)
    area = frac * H * W
    h = float(np.clip(np.sqrt(area / aspect), 2.0, H))
    w = float(np.clip(area / h, 2.0, W))
    cy = rng.uniform(h / 2, H - h / 2) if H > h else H / 2
    cx = rng.uniform(w / 2, W - w / 2) if W > w else W / 2
    yy = np.arange(H, dtype=np.float32)[:, None]
    xx = np.arange(W, dtype=np.float32)[None, :]
    edge = rng.uniform(1.0, 3.0)
    if rng.random() < 0.5:  # rectangle: signed distance to the border
        d = np.minimum(h / 2 - np.abs(yy - cy), w / 2 - np.abs(xx - cx))
    else:                   # ellipse
        r = np.sqrt(((yy - cy) / (h / 2)) ** 2 + ((xx - cx) / (w / 2)) ** 2)
        d = (1.0 - r) * min(h, w) / 2
    with np.errstate(over="ignore"):  # exp overflow -> inf -> mask value 0, intended
        m = 1.0 / (1.0 + np.exp(-d / edge))
    return m.astype(np.float32)


def make_frame(P: int, O: int, H: int, W: int, D: int = 512, *, seed: int = BASE_SEED,
               kind: str = "structured", with_targets: bool = False) -> Frame:
    rng = np.random.Generator(np.random.PCG64(seed))
    if kind == "uniform":
        pm = rng.random((P, H, W), dtype=np.float32)
        tm = rng.random((O, H, W), dtype=np.float32)
        pf = rng.standard_normal((P, D), dtype=np.float32)
        tf = rng.standard_normal((O, D), dtype=np.float32)
        perm = None
    elif kind == "structured":
        pm = np.stack([_soft_shape(rng, H, W) for _ in range(P)], 0)
        pf = rng.standard_normal((P, D), dtype=np.float32)
        # templates = distinct proposals when possible (else with repetition), shifted + noise
        perm = (rng.permutation(P)[:O] if O <= P else rng.integers(0, P, size=O)).astype(np.int64)
        tm = np.empty((O, H, W), np.float32)
        for o, p in enumerate(perm):
            dy, dx = rng.integers(-8, 9, size=2)
            sh = np.roll(np.roll(pm[p], int(dy), 0), int(dx), 1)
            sh = sh + 0.05 * rng.standard_normal((H, W), dtype=np.float32)
            tm[o] = np.clip(sh, 0.0, 1.0)
        tf = pf[perm] + np.float32(0.3) * rng.standard_normal((O, D), dtype=np.float32)
    else:
        raise ValueError(f"unknown kind {kind!r}")
    sc = rng.random(P, dtype=np.float32)
    tg = None
    if with_targets:
        if kind == "structured":
            tg = np.stack([(np.roll(pm[p], 2, 1) > 0.5) for p in perm], 0).astype(np.float32)
        else:
            tg = (rng.random((O, H, W), dtype=np.float32) > 0.5).astype(np.float32)
    return Frame(pm, tm, pf.astype(np.float32), tf.astype(np.float32), sc, tg, perm)


def make_config_frame(cfg_index: int, *, kind: str = "structured", with_targets: bool = False,
                      seed_offset: int = 0) -> Frame:
    """Frame for BASELINE.json config ``cfg_index`` (seed = 1234 + index [+ offset])."""
    c = CONFIGS[cfg_index]
    return make_frame(c["P"], c["O"], c["H"], c["W"], c["D"], seed=BASE_SEED + cfg_index + seed_offset,
                      kind=kind, with_targets=with_targets)


# ---- frame-loop reductions (tests/golden g11): inputs by seed ---------------------------------------------------
def template_planes(seed: int, O: int, H: int, W: int) -> np.ndarray:
    """[O,H,W] fp32 template planes for the box / valid reduction: plane o is empty, a binary rectangle, a soft
    blob with zero holes, or a single pixel (kinds cycle with o + seed)."""
    rng = np.random.Generator(np.random.PCG64(7000 + seed))
    m = np.zeros((O, H, W), np.float32)
    for o in range(O):
        kind = (o + seed) % 4
        if kind == 0:
            continue
        y0, x0 = int(rng.integers(0, H)), int(rng.integers(0, W))
        y1, x1 = int(rng.integers(y0, H)), int(rng.integers(x0, W))
        if kind == 1:
            m[o, y0:y1 + 1, x0:x1 + 1] = 1.0
        elif kind == 2:
            blk = rng.random((y1 - y0 + 1, x1 - x0 + 1)).astype(np.float32)
            blk[blk < 0.3] = 0.0
            m[o, y0:y1 + 1, x0:x1 + 1] = blk
        else:
            m[o, y0, x0] = 0.25
    return m


def refined_planes(seed: int, O: int, H: int, W: int) -> np.ndarray:
    """[O,H,W] fp32 'refined masks' for the label merge: uniform values, the upper half quantised to quarters (so
    background/foreground and foreground/foreground ties occur), the left third damped (background wins)."""
    rng = np.random.Generator(np.random.PCG64(8000 + seed))
    outs = rng.random((O, H, W)).astype(np.float32)
    outs[:, : H // 2] = np.round(outs[:, : H // 2] * 4) / 4
    outs[:, :, : W // 3] *= np.float32(0.4)
    return outs


# ---- seeded inputs of the matching-loss fixture G21 (tests/golden/gen_golden.py g21 and the tests regenerate them) ----------
MATCH_LOSS_CASES = [(8, 3, 16, 16, "random"), (50, 10, 24, 24, "random"), (3, 5, 8, 8, "random"), (6, 4, 8, 8, "zeros"),
                    (6, 4, 8, 8, "duplicates"), (1, 1, 4, 4, "random"), (40, 20, 12, 12, "random"), (5, 5, 6, 6, "one_target")]


def match_loss_case(k: int):
    """-> (proposals [N,H,W] fp32, targets [M,H,W] 0/1 fp32, similarity table [M,N]) of MATCH_LOSS_CASES[k]: inputs of
    compute_matching_loss chosen for their ties (empty masks: every IoU 0; duplicate planes: tied columns and rows; one live
    target plane)."""
    N, M, H, W, kind = MATCH_LOSS_CASES[k]
    rng = np.random.default_rng(2100 + k)
    P = rng.random((N, H, W)).astype(np.float32)
    Tg = (rng.random((M, H, W)) > 0.5).astype(np.float32)
    if kind == "zeros":
        P[:] = 0
        Tg[:] = 0
    elif kind == "duplicates":
        base = (rng.random((H, W)) > 0.5).astype(np.float32)
        P[:] = base
        Tg[:] = base
        Tg[2] = 0
    elif kind == "one_target":
        Tg[1:] = 0
    sim = rng.standard_normal((M, N)).astype(np.float32)
    return P, Tg, sim
