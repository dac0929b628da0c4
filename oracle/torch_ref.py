"""Op-for-op torch (CPU) restatement of the reference's matching layer -- TEST / MEASUREMENT INFRASTRUCTURE ONLY.

What north_star / SURVEY.md 8(d) call "the reference PyTorch-CPU path timed on the host cores": the same sequence of
torch tensor ops the reference issues per frame, written from SURVEY.md 8(a)'s description of a2-a7 (the reference
itself -- /root/reference -- does not exist on the GPU box):

  cosine   match_helper.py:51-64     F.cosine_similarity over [O,D,P] expands
  IoU      match_helper.py:9-28      both mask sets expanded to contiguous [O*P, HW] (match_model.py:83-87), > 0.5, sums
  sim      match_model.py:90         (1 - w) * feature_sim + w * iou
  solver   relax_match.py:36-105     greedy init with Python loops, PGD + Dykstra projections as rank-1 mm updates,
                                     .item() on the norms, both exact-equality exits
  output   match_model.py:116-147    R = mean(X_list), logic, Rb, Rb @ masks, scores

Only bench.py's ``cpu_baseline`` leg and tests import it; nothing under dmm_net_amd/ does.  tests/test_oracle_golden.py
holds it to the C oracle (scores / iteration count bit for bit on the fixture shapes).
"""
import torch
import torch.nn.functional as F


def cosine_score(tplt, prop):
    O, D = tplt.shape
    P = prop.shape[0]
    q = tplt.unsqueeze(2).expand(O, D, P)
    k = prop.t().unsqueeze(0).expand(O, D, P)
    return F.cosine_similarity(q, k, dim=1)


def iou_binary_2d(x, y):
    a, b = (x > 0.5), (y > 0.5)
    inter = (a & b).float().sum(1)
    union = (a | b).float().sum(1) + 1e-6
    return inter / union


def _project_row(X):
    """relax_match.py:9-19: the row sums by ``sum(dim=1, keepdim=True)``, broadcast back as a rank-1 ``mm`` with a ones row."""
    m = X.shape[1]
    row_sum = X.sum(dim=1, keepdim=True)                 # [n, 1]
    ones_row = torch.ones(1, m).to(X.device)             # [1, m]
    return X - (row_sum - 1).mm(ones_row) / m


def _project_col(X):
    """relax_match.py:21-34: column sums by ``sum(dim=0, keepdim=True)``, the correction as ``ones[n,1].mm(.)``, applied only
    to the columns whose sum exceeds 1 (a 0/1 float mask)."""
    n = X.shape[0]
    col_sum = X.sum(dim=0, keepdim=True)                 # [1, m]
    ones_col = torch.ones(n, 1).to(X.device)             # [n, 1]
    mask = (col_sum <= 1).float()
    Y = X - ones_col.mm(col_sum - 1) / n
    return X * mask + (1 - mask) * Y


def relax_matching(C, max_iter, proj_iter, lr):
    n, m = C.shape
    X = torch.zeros_like(C)
    Crm = C.clone()
    for j in range(m):                                   # column minima keep their value, everything else the maximum
        i_star = torch.argmin(Crm[:, j])
        for i in range(n):
            if i != i_star:
                Crm[i, j] = C.max()                      # (a full reduction per element, as the reference issues it: :46-51)
    _, idx = torch.min(Crm, dim=1)
    X[torch.arange(n).long(), idx.long()] = 1.0
    X_list, cost = [X], [0.0]
    P0, P1, P2 = torch.zeros_like(C), torch.zeros_like(C), torch.zeros_like(C)
    for _ in range(max_iter):
        X = X - lr * C
        cost.append((X * C).norm().item())
        X_list.append(X)
        for _ in range(proj_iter):
            Xs = X.clone()                               # (:74)
            X = X + P0
            Y = F.relu(X)
            P0 = X - Y
            X = Y + P1
            Y = _project_col(X)
            P1 = X - Y
            X = Y + P2
            Y = _project_row(X)
            P2 = X - Y
            X = Y
            if (X - Xs).norm().item() == 0:
                break
            _ = (X - Xs).norm().item()                   # the reference reads the norm a second time for its log (:91)
        if cost[-2] == cost[-1]:
            break
    return X, cost, X_list


def match_forward(prop_feat, prop_mask, tplt_feat, tplt_mask, prop_score, *, score_weight=0.3, max_iter=20, proj_iter=5,
                  lr=0.1, is_test=1):
    """One frame, algo 'relax', no targets.  Returns dict(full_outmask, match_score, det_score, iters)."""
    P, H, W = prop_mask.shape
    O = tplt_mask.shape[0]
    feature_sim = cosine_score(tplt_feat, prop_feat)
    tm = tplt_mask.reshape(O, 1, H * W).expand(O, P, H * W).contiguous().view(O * P, H * W)
    pm = prop_mask.reshape(1, P, H * W).expand(O, P, H * W).contiguous().view(O * P, H * W)
    iou = iou_binary_2d(pm, tm).view(O, P)
    sim = (1 - score_weight) * feature_sim + score_weight * iou
    mask2d, score = prop_mask.reshape(P, H * W), prop_score
    if P <= O:                                           # pad to O + 1 columns with zeros
        pad = O + 1 - P
        sim = torch.cat([sim, sim.new_zeros(O, pad)], 1)
        mask2d = torch.cat([mask2d, mask2d.new_zeros(pad, H * W)], 0)
        score = torch.cat([score, score.new_zeros(pad)], 0)
    X, cost, X_list = relax_matching(-sim, max_iter, proj_iter, lr)
    R = torch.stack(X_list, 0).mean(0)
    maxv = R.max(1, keepdim=True)[0]
    logic = (R == maxv).float() if is_test else (R > 0.01).float()
    Rb = R * logic
    full = torch.mm(Rb, mask2d).view(O, H, W)
    match_score = (R.clamp(0, 1) * sim).max(1)[0]
    det_score = (score.view(1, -1) * Rb).sum(1)
    return dict(full_outmask=full, match_score=match_score, det_score=det_score, iters=len(X_list) - 1)
