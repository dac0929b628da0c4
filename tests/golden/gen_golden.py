#!/usr/bin/env python3
"""Capture golden vectors from the REFERENCE's own hot-path modules.

Runs ONLY in the build container (needs /root/reference; the GPU box has none):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden.py

It imports ``dmm.modules.match_model``, ``dmm.utils.match_helper`` and
``dmm.modules.submodules.relax_match`` from /root/reference (torch CPU, fp32), feeds them the
build-owned seeded inputs of ``dmm_net_amd.synth`` (plus hand-made edge cases) and stores
inputs-by-seed + expected outputs as small ``.npz`` fixtures next to this script.  The fixtures
are data only -- no reference source text is stored.  Groups follow SURVEY.md section 8c:

  G1 solver known-answer test of the reference's own self-test (relax_match.py:108-119)
  G2 config 1 (P=8, O=3, 64x64): every intermediate, is_test 0/1, four iteration settings,
     + targets -> gt_iou / gt_matched / cost_loss
  G3 pad path P <= O
  G4 config 2 / config 5 shapes at 255x255: [M,N]-sized tables + checksums of the big output
  G5 edge cases (empty masks, 0.5 pixels, ties, no-column-minimum rows, zero sim, zero features)
  G6 backward of the layer wrt the features
  G7 solver-only and cosine-only known answers on many [n, m] / (O, P, D) shapes (pins the reduction order)
  G8 the DMM_Model per-video harness steps (dmm_model.py:115-141) re-executed around the imported
     MatchModel (DMM_Model itself needs maskrcnn_benchmark and cannot be imported)
"""
import os
import sys

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from dmm.modules.match_model import MatchModel  # noqa: E402  (reference)
from dmm.modules.submodules.relax_match import relax_matching  # noqa: E402  (reference)
from dmm.utils import match_helper  # noqa: E402  (reference)
from scipy.optimize import linear_sum_assignment  # noqa: E402

from dmm_net_amd import synth  # noqa: E402

torch.set_num_threads(8)
torch.manual_seed(0)


def cfg(max_iter, proj_iter, lr=0.1, w=0.3, algo="relax"):
    return {"matching": {"algo": algo}, "relax_max_iter": max_iter, "relax_proj_iter": proj_iter,
            "relax_learning_rate": lr, "score_weight": w}


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def int_tables(pm, tm):
    a = T(pm).flatten(1) > 0.5
    b = T(tm).flatten(1) > 0.5
    inter = (b[:, None, :] & a[None, :, :]).sum(-1).to(torch.int32)
    return inter.numpy(), a.sum(1).to(torch.int32).numpy(), b.sum(1).to(torch.int32).numpy()


def run_layer(fr, max_iter, proj_iter, is_test, lr=0.1, w=0.3, with_targets=False, full=True):
    """Run the reference layer piecewise so every intermediate is captured."""
    model = MatchModel(cfg(max_iter, proj_iter, lr, w), is_test)
    pf, tf = T(fr.proposed_feature), T(fr.template_feature)
    pm, tm, sc = T(fr.proposed_mask), T(fr.mask_last_occurence), T(fr.proposal_score)
    tg = T(fr.targets) if with_targets else None
    out = {}
    with torch.no_grad():
        feats = {"proposed": pf, "template": [tf]}
        masks = {"proposed": pm, "template": tm}
        sim, n_prop, n_tplt, mloss = model.compute_cost_matrix(feats, masks, {"proposal_score": sc}, tg)
        cosv = match_helper.get_cosine_score(tf, pf)
        P, O = pm.shape[0], tm.shape[0]
        iou = match_helper.compute_iou_binary_mask_2D(
            pm.view(P, -1).expand(O, -1, -1).contiguous().view(O * P, -1),
            tm.contiguous().view(O, 1, -1).expand(-1, P, -1).contiguous().view(O * P, -1)).view(O, P)
        out.update(cos=cosv.numpy(), iou=iou.numpy(), sim=sim.numpy())
        if with_targets:
            out["cost_loss"] = np.float32(mloss["cost_loss"].item())
            bp = pm > 0.5
            gt_iou = match_helper.compute_iou_binary_mask_2D(
                bp.view(P, -1).expand(O, -1, -1).contiguous().view(O * P, -1),
                tg.contiguous().view(O, 1, -1).expand(-1, P, -1).contiguous().view(O * P, -1)).view(O, P)
            gt_matched = relax_matching(-gt_iou, max_iter=0, proj_iter=0, lr=0)[0]
            out.update(gt_iou=gt_iou.numpy(), gt_matched=gt_matched.numpy())
        # pad + solver exactly as match_with_first_frame does (match_model.py:107-121)
        if sim.shape[1] <= sim.shape[0]:
            simp = sim.new_zeros((sim.shape[0], sim.shape[0] + 1))
            simp[:, :sim.shape[1]] = sim
        else:
            simp = sim
        X, cost, X_list, _ = relax_matching(-simp, max_iter=max_iter, proj_iter=proj_iter, lr=lr)
        R = sum(X_list) / len(X_list)
        out.update(X_final=X.numpy(), cost=np.asarray(cost, np.float32), n_xlist=np.int32(len(X_list)),
                   R=R.numpy(), X0=X_list[0].numpy())
        if full:
            out["xlist"] = torch.stack(X_list).numpy()
        fo, ms, ds, logic, Rb = model.match_with_first_frame(sim, n_prop, n_tplt, pm.float(), sc, tm)
        out.update(match_score=ms.numpy(), det_score=ds.numpy(), logic=logic.numpy(), Rb=Rb.numpy())
        # the public forward (5-tuple) must agree with the piecewise run
        f5 = model(pf, pm, [tf], tm, sc, tg)
        assert torch.equal(f5[0], fo) and torch.equal(f5[1], ms) and torch.equal(f5[2], ds)
        assert f5[3] is f5[0]
        if full:
            out["full_outmask"] = fo.numpy()
        else:
            fo64 = fo.double()
            out["outmask_sum"] = fo64.flatten(1).sum(1).numpy()
            out["outmask_sample"] = fo.flatten(1)[:, ::997].numpy()
        out["argmax"] = R.argmax(1).to(torch.int32).numpy()
    inter, ap, at = int_tables(fr.proposed_mask, fr.mask_last_occurence)
    iou_chk = torch.from_numpy(inter).float() / ((torch.from_numpy(ap)[None, :] + torch.from_numpy(at)[:, None]
                                                   - torch.from_numpy(inter)).float() + 1e-6)
    assert torch.equal(iou_chk, T(out["iou"])), "integer tables disagree with the reference's iou"
    out.update(inter=inter, area_p=ap, area_t=at)
    return out


def save(name, d):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **d)
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.1f} KiB  ({len(d)} arrays)")


def flat(prefix, d):
    return {f"{prefix}/{k}": v for k, v in d.items()}


# ------------------------------------------------------------------------------------------ G1
def g1():
    cost = np.array([[4, 1, 3], [2, 0, 5], [3, 2, 2]])
    r, c = linear_sum_assignment(cost)
    C = torch.from_numpy(cost).float()
    X, costs, X_list, inner = relax_matching(C, max_iter=100, proj_iter=100, lr=0.1)
    d = dict(C=C.numpy(), hungarian_cols=c.astype(np.int32), X_final=X.numpy(),
             n_xlist=np.int32(len(X_list)), R=(sum(X_list) / len(X_list)).numpy(),
             cost=np.asarray(costs, np.float32), xlist=torch.stack(X_list).numpy(),
             max_iter=np.int32(100), proj_iter=np.int32(100), lr=np.float32(0.1))
    # more solver-only KATs on random costs, with and without early exits
    rng = np.random.Generator(np.random.PCG64(77))
    k = 0
    for (n, m, mi, pi) in [(3, 8, 20, 5), (10, 50, 20, 5), (10, 50, 40, 5), (5, 50, 10, 5), (20, 200, 20, 5),
                           (4, 5, 400, 50), (10, 50, 400, 50), (2, 3, 100, 100), (6, 7, 60, 1)]:
        Cn = -rng.random((n, m), dtype=np.float32)
        X, costs, X_list, inner = relax_matching(T(Cn), max_iter=mi, proj_iter=pi, lr=0.1)
        d.update(flat(f"rand{k}", dict(C=Cn, X_final=X.numpy(), n_xlist=np.int32(len(X_list)),
                                      R=(sum(X_list) / len(X_list)).numpy(), cost=np.asarray(costs, np.float32),
                                      max_iter=np.int32(mi), proj_iter=np.int32(pi), lr=np.float32(0.1),
                                      n_inner_last=np.int32(len(inner)))))
        k += 1
    d["n_rand"] = np.int32(k)
    save("g1_solver_kat", d)


def g7():
    """Solver-only KATs on many [n, m] shapes: pins the reduction order of the torch CPU build the
    goldens come from (AVX2 cascade sums) for every kernel envelope, incl. early exits."""
    rng = np.random.Generator(np.random.PCG64(707))
    d = {}
    shapes = [(1, 2), (1, 9), (2, 3), (3, 4), (4, 5), (5, 6), (7, 8), (3, 7), (8, 9), (9, 10), (10, 11), (5, 16),
              (5, 31), (5, 32), (5, 33), (10, 40), (10, 48), (10, 63), (10, 64), (12, 65), (15, 100), (16, 17),
              (16, 128), (17, 129), (20, 21), (20, 200), (24, 255), (31, 256), (32, 33), (32, 256), (10, 50)]
    k = 0
    for (n, m) in shapes:
        for (mi, pi) in ((12, 4), (60, 8)):
            if n * m > 4000 and mi > 12:
                continue
            Cn = -rng.random((n, m), dtype=np.float32)
            if k % 3 == 1:
                Cn = (Cn * np.float32(0.2)).astype(np.float32)
            X, costs, X_list, inner = relax_matching(T(Cn), max_iter=mi, proj_iter=pi, lr=0.1)
            d.update(flat(f"k{k}", dict(C=Cn, X_final=X.numpy(), n_xlist=np.int32(len(X_list)),
                                       R=(sum(X_list) / len(X_list)).numpy(), cost=np.asarray(costs, np.float32),
                                       max_iter=np.int32(mi), proj_iter=np.int32(pi), lr=np.float32(0.1))))
            k += 1
    d["n"] = np.int32(k)
    # cosine tables on several (O, P, D) incl. the degenerate P == 1 layout and ragged D
    j = 0
    for (O, P, D) in [(3, 8, 512), (10, 50, 512), (5, 3, 64), (2, 7, 33), (1, 1, 16), (4, 1, 512), (1, 5, 40),
                      (20, 200, 512), (6, 33, 100), (7, 64, 256), (3, 65, 1024), (2, 40, 8), (2, 9, 7)]:
        q = rng.standard_normal((O, D), dtype=np.float32)
        kf = rng.standard_normal((P, D), dtype=np.float32)
        if j == 2:
            q[0] = 0
            kf[1] = np.float32(1e-12)
        c = match_helper.get_cosine_score(T(q), T(kf)).numpy()
        d.update(flat(f"cos{j}", dict(q=q, k=kf, cos=c)))
        j += 1
    d["n_cos"] = np.int32(j)
    save("g7_shapes", d)


# ------------------------------------------------------------------------------------------ G2
ITER_SETTINGS = [(10, 5), (40, 5), (20, 5), (0, 0)]


def g2():
    d = {}
    for kind in ("structured", "uniform"):
        fr = synth.make_config_frame(1, kind=kind, with_targets=True)
        d[f"{kind}/checksum"] = np.array(fr.checksum())
        for is_test in (0, 1):
            for (mi, pi) in ITER_SETTINGS:
                o = run_layer(fr, mi, pi, is_test, with_targets=True)
                d.update(flat(f"{kind}/t{is_test}/i{mi}_{pi}", o))
    save("g2_config1", d)


# ------------------------------------------------------------------------------------------ G3
def g3():
    d = {}
    for (P, O) in [(3, 5), (1, 1), (5, 5), (2, 1), (1, 4)]:
        fr = synth.make_frame(P, O, 64, 64, 512, seed=synth.BASE_SEED + 100 + 10 * P + O, kind="structured",
                              with_targets=True)
        d[f"p{P}o{O}/checksum"] = np.array(fr.checksum())
        for is_test in (0, 1):
            o = run_layer(fr, 20, 5, is_test, with_targets=True)
            d.update(flat(f"p{P}o{O}/t{is_test}", o))
    save("g3_pad", d)


# ------------------------------------------------------------------------------------------ G4
def g4():
    d = {}
    for ci in (2, 5):
        for kind in ("structured", "uniform"):
            fr = synth.make_config_frame(ci, kind=kind)
            d[f"c{ci}/{kind}/checksum"] = np.array(fr.checksum())
            for is_test in (1, 0):
                if ci == 5 and is_test == 0:
                    continue
                o = run_layer(fr, 20, 5, is_test, full=False)
                d.update(flat(f"c{ci}/{kind}/t{is_test}", o))
            if ci == 2:
                o = run_layer(fr, 40, 5, 1, full=False)
                d.update(flat(f"c{ci}/{kind}/eval40", o))
    save("g4_big", d)


# ------------------------------------------------------------------------------------------ G5
def g5():
    d = {}
    rng = np.random.Generator(np.random.PCG64(55))
    H = W = 16
    D = 32

    def base(P, O):
        fr = synth.make_frame(P, O, H, W, D, seed=5000 + 10 * P + O, kind="uniform")
        return fr

    cases = {}
    # (a) all-zero masks -> union 0 -> iou 0
    fr = base(6, 3)
    fr.proposed_mask[:] = 0
    fr.mask_last_occurence[:] = 0
    cases["zero_masks"] = fr
    # (b) pixels exactly 0.5 are NOT set (strict >)
    fr = base(6, 3)
    fr.proposed_mask[:] = np.where(rng.random(fr.proposed_mask.shape) < 0.5, 0.5, 0.75).astype(np.float32)
    fr.mask_last_occurence[:] = np.where(rng.random(fr.mask_last_occurence.shape) < 0.5, 0.5,
                                         np.nextafter(np.float32(0.5), np.float32(1))).astype(np.float32)
    cases["half_pixels"] = fr
    # (c) duplicate proposals (argmin ties) and duplicate templates
    fr = base(6, 3)
    fr.proposed_mask[3] = fr.proposed_mask[1]
    fr.proposed_feature[3] = fr.proposed_feature[1]
    fr.mask_last_occurence[2] = fr.mask_last_occurence[0]
    fr.template_feature[2] = fr.template_feature[0]
    cases["duplicates"] = fr
    # (d) a template that owns no column minimum (its sim is the lowest everywhere) -> picks col 0
    fr = base(6, 3)
    fr.mask_last_occurence[1] = 0
    fr.template_feature[1] = -fr.proposed_feature.mean(0)
    cases["no_col_min"] = fr
    # (e) all-zero sim: zero features and empty masks -> outer exit at it=0 (cost[0]==cost[1]==0)
    fr = base(6, 3)
    fr.proposed_mask[:] = 0
    fr.proposed_feature[:] = 0
    cases["zero_sim"] = fr
    # (f) zero-norm feature rows (eps clamp), rest generic
    fr = base(6, 3)
    fr.proposed_feature[2] = 0
    fr.template_feature[0] = 0
    cases["zero_feature_rows"] = fr
    # (g) tiny-norm features (norm below eps = 1e-8)
    fr = base(6, 3)
    fr.proposed_feature[4] = np.float32(1e-12)
    cases["tiny_feature_rows"] = fr
    # (h) full masks (everything set)
    fr = base(6, 3)
    fr.proposed_mask[:] = 1
    fr.mask_last_occurence[:] = 1
    cases["full_masks"] = fr
    # (i) odd plane size, single pixel row
    fr = synth.make_frame(5, 2, 1, 7, D, seed=5999, kind="uniform")
    cases["tiny_plane"] = fr
    for name, fr in cases.items():
        d.update(flat(f"{name}/in", dict(pm=fr.proposed_mask, tm=fr.mask_last_occurence, pf=fr.proposed_feature,
                                         tf=fr.template_feature, sc=fr.proposal_score)))
        for is_test in (0, 1):
            o = run_layer(fr, 20, 5, is_test)
            d.update(flat(f"{name}/t{is_test}", o))
    d["names"] = np.array(sorted(cases.keys()))
    save("g5_edge", d)


# ------------------------------------------------------------------------------------------ G6
def g6():
    d = {}
    for name, (P, O, mi, pi, is_test) in {"c1_train": (8, 3, 10, 5, 0), "c1_test": (8, 3, 20, 5, 1),
                                           "pad": (3, 5, 10, 5, 0), "mid": (20, 5, 10, 5, 0)}.items():
        fr = synth.make_frame(P, O, 64, 64, 512, seed=synth.BASE_SEED + 600 + P + O, kind="structured",
                              with_targets=True)
        d[f"{name}/checksum"] = np.array(fr.checksum())
        d[f"{name}/shape"] = np.array([P, O, 64, 64, 512, mi, pi, is_test], np.int32)
        model = MatchModel(cfg(mi, pi), is_test)
        pf = T(fr.proposed_feature).requires_grad_(True)
        tf = T(fr.template_feature).requires_grad_(True)
        gen = torch.Generator().manual_seed(7)
        wmask = torch.rand((O, 64, 64), generator=gen)
        wms, wds = torch.rand(O, generator=gen), torch.rand(O, generator=gen)
        fo, ms, ds, _, loss = model(pf, T(fr.proposed_mask), [tf], T(fr.mask_last_occurence),
                                    T(fr.proposal_score), T(fr.targets))
        total = (fo * wmask).sum() + (ms * wms).sum() + (ds * wds).sum() + 3.0 * loss["cost_loss"]
        total.backward()
        d.update(flat(name, dict(wmask=wmask.numpy(), wms=wms.numpy(), wds=wds.numpy(),
                                 total=np.float32(total.item()), cost_loss=np.float32(loss["cost_loss"].item()),
                                 grad_pf=pf.grad.numpy(), grad_tf=tf.grad.numpy(),
                                 full_outmask=fo.detach().numpy())))
    save("g6_backward", d)


# ------------------------------------------------------------------------------------------ G8
def g8():
    """Steps of DMM_Model.forward / .inference (dmm_model.py:48-158) around the imported layer."""
    d = {}
    B, F, P, H, W, D = 3, 5, 8, 32, 32, 64
    n_valid = [0, 2, 5]
    rng = np.random.Generator(np.random.PCG64(88))
    mask_last = np.zeros((B, F, H, W), np.float32)
    tplt_feat = np.zeros((B, F, D), np.float32)
    targets = np.zeros((B, F, H, W), np.float32)
    valid = np.zeros((B, F), np.float32)
    frames = []
    for b in range(B):
        fr = synth.make_frame(P, F, H, W, D, seed=8800 + b, kind="structured", with_targets=True)
        frames.append(fr)
        O = n_valid[b]
        valid[b, :O] = 1
        mask_last[b, :O] = fr.mask_last_occurence[:O]
        mask_last[b, O:] = rng.random((F - O, H, W), dtype=np.float32) * 0.1   # stale junk in invalid slots
        tplt_feat[b] = fr.template_feature
        targets[b, :O] = fr.targets[:O]
    for mode in ("train", "test"):
        is_test = int(mode == "test")
        layer = MatchModel(cfg(10, 5), is_test)
        out_mask, out_last, losses = [], [], []
        with torch.no_grad():
            for b in range(B):
                tv = T(valid[b])
                O = int(tv.sum().item())
                ml = T(mask_last[b])
                if O == 0:
                    out_mask.append(ml.new_zeros(F, H, W))
                    out_last.append(ml)
                    losses.append(0.0)
                    continue
                FF = torch.diag(tv).float()
                OF = FF[:O, :]
                tfv = [torch.mm(OF, T(tplt_feat[b]).view(F, -1)).view(O, D)]
                fo, ms, ds, newm, loss = layer(T(frames[b].proposed_feature), T(frames[b].proposed_mask), tfv,
                                               ml[:O].view(O, H, W), T(frames[b].proposal_score),
                                               targets=None if is_test else T(targets[b, :O]))
                FO = OF.t()
                out_mask.append(torch.mm(FO, fo.view(O, -1)).view(F, H, W))
                out_last.append(torch.mm(FO, newm.view(O, -1)).view(F, H, W))
                losses.append(float(loss["cost_loss"]) if len(loss) > 0 else 0.0)
        d[f"{mode}/output_mask"] = torch.stack(out_mask).numpy()
        d[f"{mode}/out_mask_last"] = torch.stack(out_last).numpy()
        d[f"{mode}/losses"] = np.asarray(losses, np.float32)
    d.update(mask_last=mask_last, tplt_feat=tplt_feat, targets=targets, valid=valid,
             shape=np.array([B, F, P, H, W, D], np.int32), n_valid=np.array(n_valid, np.int32))
    for b, fr in enumerate(frames):
        d[f"frame{b}/checksum"] = np.array(fr.checksum())
    save("g8_harness", d)


# ------------------------------------------------------------------------------------------ G9
def g9():
    """Proposal preprocessing: the steps of paste_mask_in_image / binmask_to_box (dmm/utils/masker.py:110-173)
    re-executed with torch (masker.py itself needs maskrcnn_benchmark; its `interpolate` is torch's
    F.interpolate for non-empty inputs).  Pins the bilinear paste + tight-box arithmetic."""
    import torch.nn.functional as F
    rng = np.random.Generator(np.random.PCG64(909))
    d = {}
    k = 0
    for (im_h, im_w, P, M) in [(64, 96, 6, 28), (255, 255, 12, 28), (255, 448, 10, 28), (33, 47, 5, 14)]:
        prob = rng.random((P, 1, M, M), dtype=np.float32)
        x1 = rng.uniform(-10, im_w * 0.8, P)
        y1 = rng.uniform(-10, im_h * 0.8, P)
        bw = rng.uniform(0.3, im_w * 0.7, P)
        bh = rng.uniform(0.3, im_h * 0.7, P)
        boxes = np.stack([x1, y1, x1 + bw, y1 + bh], 1).astype(np.float32)
        boxes[0] = [3.2, 4.7, 3.9, 5.1]                     # sub-pixel box
        boxes[1] = [im_w - 5.0, im_h - 6.0, im_w + 30.0, im_h + 12.0]   # sticks out bottom-right
        prob[2] = 0.1                                       # nothing above the threshold -> fallback box
        masks, nboxes = [], []
        padding, thresh = 1, 0.4
        for p in range(P):
            mask = T(prob[p, 0])
            box = T(boxes[p])
            # expand_masks (masker.py:110-117)
            pad2 = 2 * padding
            scale = float(M + pad2) / M
            padded = mask.new_zeros((1, 1, M + pad2, M + pad2))
            padded[:, :, padding:-padding, padding:-padding] = mask
            # expand_boxes (masker.py:93-107)
            w_half = (box[2] - box[0]) * .5
            h_half = (box[3] - box[1]) * .5
            x_c = (box[2] + box[0]) * .5
            y_c = (box[3] + box[1]) * .5
            w_half = w_half * scale
            h_half = h_half * scale
            eb = torch.stack([x_c - w_half, y_c - h_half, x_c + w_half, y_c + h_half]).to(dtype=torch.int32)
            w = max(int(eb[2] - eb[0] + 1), 1)
            h = max(int(eb[3] - eb[1] + 1), 1)
            m = F.interpolate(padded.to(torch.float32), size=(h, w), mode='bilinear', align_corners=False)[0][0]
            im_mask = m.new_zeros((im_h, im_w))
            x_0 = max(int(eb[0]), 0)
            x_1 = min(int(eb[2]) + 1, im_w)
            y_0 = max(int(eb[1]), 0)
            y_1 = min(int(eb[3]) + 1, im_h)
            if y_1 > y_0 and x_1 > x_0:
                im_mask[y_0:y_1, x_0:x_1] = m[(y_0 - int(eb[1])):(y_1 - int(eb[1])), (x_0 - int(eb[0])):(x_1 - int(eb[0]))]
            # binmask_to_box (masker.py:157-173)
            inds = (im_mask > thresh).nonzero()
            if inds.shape[0] < 1:
                nb = [0, 0, im_h, im_w]
            else:
                nb = [int(inds[:, 1].min()), int(inds[:, 0].min()), int(inds[:, 1].max()), int(inds[:, 0].max())]
            masks.append(im_mask.numpy())
            nboxes.append(nb)
        d.update(flat(f"c{k}", dict(prob=prob, boxes=boxes, size=np.array([im_h, im_w], np.int32),
                                   masks=np.stack(masks), new_boxes=np.asarray(nboxes, np.float32),
                                   thresh=np.float32(thresh), padding=np.int32(padding))))
        k += 1
    d["n"] = np.int32(k)
    save("g9_paste", d)


def g10():
    """template_feature lists with more than one entry: feature_sim = mean of the per-entry cosines
    (match_model.py:71-76).  DMM-Net itself always passes one entry; the layer's API allows more."""
    d = {}
    P, O, H, W, D = 9, 4, 32, 32, 96
    fr = synth.make_frame(P, O, H, W, D, seed=synth.BASE_SEED + 1010, kind="structured", with_targets=True)
    rng = np.random.Generator(np.random.PCG64(1010))
    tf2 = (fr.template_feature + 0.5 * rng.standard_normal((O, D), dtype=np.float32)).astype(np.float32)
    tf3 = rng.standard_normal((O, D), dtype=np.float32)
    d["checksum"] = np.array(fr.checksum())
    d["tf2"], d["tf3"] = tf2, tf3
    for is_test in (0, 1):
        model = MatchModel(cfg(10, 5), is_test)
        pf = T(fr.proposed_feature).requires_grad_(True)
        tfs = [T(fr.template_feature).requires_grad_(True), T(tf2).requires_grad_(True), T(tf3).requires_grad_(True)]
        fo, ms, ds, _, loss = model(pf, T(fr.proposed_mask), tfs, T(fr.mask_last_occurence), T(fr.proposal_score),
                                    T(fr.targets))
        gen = torch.Generator().manual_seed(3)
        wmask = torch.rand((O, H, W), generator=gen)
        total = (fo * wmask).sum() + ms.sum() + 2.0 * loss["cost_loss"]
        total.backward()
        d.update(flat(f"t{is_test}", dict(full_outmask=fo.detach().numpy(), match_score=ms.detach().numpy(),
                                          det_score=ds.detach().numpy(), cost_loss=np.float32(loss["cost_loss"].item()),
                                          wmask=wmask.numpy(), grad_pf=pf.grad.numpy(),
                                          grad_tf0=tfs[0].grad.numpy(), grad_tf1=tfs[1].grad.numpy(),
                                          grad_tf2=tfs[2].grad.numpy())))
    save("g10_multi_template", d)


def verify_sweep():
    """Not a fixture: cross-checks the ORACLE against the imported reference on the broad shape sweep that
    tests/test_gpu_parity.py runs GPU-vs-oracle (all row counts 1..32 x 17 width classes, 12 cosine shapes).
    Run in the build container: `python tests/golden/gen_golden.py verify_sweep` -> "0 mismatches"."""
    import oracle
    rng = np.random.Generator(np.random.PCG64(2024))
    widths = [2, 3, 7, 8, 9, 31, 32, 33, 63, 64, 65, 100, 128, 129, 200, 255, 256]
    bad = tot = 0
    for n in range(1, 33):
        for m in widths:
            C = (-rng.random((2, n, m), dtype=np.float32) * np.float32(0.7)).astype(np.float32)
            for b in range(2):
                X, cost, xl, _ = relax_matching(T(C[b]), max_iter=4, proj_iter=3, lr=0.1)
                R = (sum(xl) / len(xl)).numpy()
                o = oracle.relax(C[b], 4, 3, 0.1)
                tot += 1
                bad += not (np.array_equal(o["X"], X.numpy()) and np.array_equal(o["R"], R)
                            and o["iters"] + 1 == len(xl))
    for (O, P, D) in [(1, 1, 16), (2, 3, 33), (8, 9, 100), (5, 33, 64), (16, 64, 512), (17, 65, 40), (20, 130, 256),
                      (32, 7, 8), (31, 255, 48), (3, 2, 9), (4, 8, 5), (2, 12, 1030)]:
        q = rng.standard_normal((O, D), dtype=np.float32)
        k = rng.standard_normal((P, D), dtype=np.float32)
        ref = match_helper.get_cosine_score(T(q), T(k)).numpy()
        tot += 1
        bad += not np.array_equal(oracle.cosine(q, k), ref)
    print(f"verify_sweep: {tot} cases, {bad} mismatches")
    assert bad == 0


def g11():
    """Frame-loop reductions: the torch steps of ohw_mask2boxlist / binmask_to_bbox_xyxy_pt
    (dmm/utils/utils.py:114-143, :179-210) and of the evaluator's label merge (dmm/modules/evaluator.py:134-139)
    re-executed here (utils.py needs torchvision + maskrcnn_benchmark, evaluator.py the whole model stack)."""
    d = {}

    def ref_boxes(ohw):                                     # utils.py:188-199 + :120-143
        O, H, W = ohw.shape
        valid = (ohw.sum(2).sum(1) > 0).long()
        boxes = []
        for o in range(O):
            inds = torch.nonzero((ohw[o] > 0).float())
            if inds.shape[0] == 0:
                boxes.append([0, 0, W - 1, H - 1])
                continue
            x_min = max(inds[:, 1].min() - 0, 0)
            y_min = max(inds[:, 0].min() - 0, 0)
            x_max = min(inds[:, 1].max() + 0, W - 1)
            y_max = min(inds[:, 0].max() + 0, H - 1)
            boxes.append([int(x_min), int(y_min), int(x_max), int(y_max)])
        return np.asarray(boxes, np.float32), valid.numpy().astype(np.int32)

    def ref_merge(outs, n_obj, H, W):                       # evaluator.py:134-139 for one video
        refine_mask = outs[:n_obj].view(-1, H * W)
        refine_bg = 1 - refine_mask.max(0)[0]
        refine_fbg = torch.cat([refine_bg.view(1, H, W), refine_mask.view(-1, H, W)], dim=0)
        _, max_i = refine_fbg.max(0)
        return max_i.float().numpy().astype(np.uint8)       # plot_scores_map: astype(np.uint8) (eval_helper.py:36)

    shapes = [(3, 17, 23), (5, 64, 64), (10, 255, 255), (1, 9, 1), (4, 33, 130), (8, 255, 448)]
    for k, (O, H, W) in enumerate(shapes):                  # inputs by seed: synth.template_planes(k, O, H, W)
        boxes, valid = ref_boxes(T(synth.template_planes(k, O, H, W)))
        d[f"box{k}_shape"] = np.asarray([O, H, W], np.int64)
        d[f"box{k}_boxes"] = boxes
        d[f"box{k}_valid"] = valid
    d["n_box"] = np.int64(len(shapes))
    cases = [(5, 3, 20, 31), (5, 5, 64, 64), (3, 1, 7, 9), (10, 7, 255, 255), (6, 2, 40, 40), (5, 4, 255, 448)]
    for k, (O, n_obj, H, W) in enumerate(cases):            # inputs by seed: synth.refined_planes(k, O, H, W)
        outs = synth.refined_planes(k, O, H, W)
        d[f"mrg{k}_shape"] = np.asarray([O, n_obj, H, W], np.int64)
        d[f"mrg{k}_labels"] = ref_merge(T(outs.reshape(O, H * W)), n_obj, H, W)
    d["n_mrg"] = np.int64(len(cases))
    save("g11_frame_loop", d)


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g6", "g7", "g8", "g9", "g10", "g11"]
    for w in which:
        globals()[w]()
