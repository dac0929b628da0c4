"""GPU parity of the GENERAL kernels (dmm_wide.hip, the general mask mix): tables outside the envelope the fast kernels are
compiled for -- more than 32 template rows, solver width max(N, M + 1) above 256.  The reference is unbounded
(relax_match.py:36-105); bars as everywhere: integer and fp32 tables, scores and executed iterations BIT exact against the
oracle and against the reference's own output (G19), test-mode masks bit exact, train-mode masks within 1e-5.
With option FORCE_WIDE (dmm_set_option) the same kernels run INSIDE the envelope, where every shape also has a fast kernel to agree with."""
import numpy as np
import pytest
import torch

import oracle
from conftest import golden
from dmm_net_amd import _lib, ops, synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
WIDE_CASES = [(300, 40, 1), (20, 50, 1), (20, 50, 0), (257, 33, 0), (400, 1, 1)]     # gen_golden.WIDE_CASES


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def forward(frames, max_iter, proj_iter, is_test, n_valid=None, m_valid=None):
    """frames: list of synth frames of one shape -> batched ops.match_forward with tables, as numpy."""
    st = lambda key: dev(np.stack([getattr(f, key) for f in frames]))
    full, ms, ds, it, tab = ops.match_forward(st("proposed_mask"), st("mask_last_occurence"), st("proposed_feature"),
                                              st("template_feature"), st("proposal_score"), score_weight=0.3,
                                              max_iter=max_iter, proj_iter=proj_iter, lr=0.1, is_test=is_test,
                                              n_valid=n_valid, m_valid=m_valid, return_tables=True)
    torch.cuda.synchronize()
    out = dict(full_outmask=full, match_score=ms, det_score=ds, iters=it, **tab)
    return {k: v.cpu().numpy() for k, v in out.items()}


def check_frame(g, b, o, is_test, N, M):
    """frame b of the batched result g against the oracle's dict o for its live [M', N'] block."""
    Mo, No = o["sim"].shape
    Ppo = o["R"].shape[1]
    assert int(g["iters"][b]) == o["iters"]
    assert np.array_equal(g["sim"][b, :Mo, :No], o["sim"]), float(np.abs(g["sim"][b, :Mo, :No] - o["sim"]).max())
    assert np.array_equal(g["R"][b, :Mo, :Ppo], o["R"]), float(np.abs(g["R"][b, :Mo, :Ppo] - o["R"]).max())
    assert np.array_equal(g["Rb"][b, :Mo, :Ppo], o["Rb"])
    assert np.array_equal(g["match_score"][b, :Mo], o["match_score"])
    assert np.array_equal(g["det_score"][b, :Mo], o["det_score"])
    full = g["full_outmask"][b, :Mo].reshape(Mo, -1)
    ref = o["full_outmask"].reshape(Mo, -1)
    if is_test:
        assert np.array_equal(full, ref)
    else:
        assert float(np.abs(full.astype(np.float64) - ref).max()) <= 1e-5
    # everything outside the live block is zero
    for key, lim in (("sim", (Mo, No)), ("R", (Mo, Ppo)), ("Rb", (Mo, Ppo))):
        t = g[key][b].copy()
        t[:lim[0], :lim[1]] = 0
        assert not t.any(), key
    assert not g["match_score"][b, Mo:].any() and not g["det_score"][b, Mo:].any() and not g["full_outmask"][b, Mo:].any()


@pytest.mark.parametrize("P,O", [(300, 40), (257, 33), (64, 64), (400, 1), (20, 50), (130, 129), (513, 70), (1, 40)])
@pytest.mark.parametrize("is_test", [1, 0])
def test_wide_tables_bit_exact_vs_oracle(P, O, is_test):
    fr = synth.make_frame(P, O, 24, 24, 64, seed=9000 + P + 7 * O, kind="uniform")
    o = oracle.match_forward(fr.proposed_mask, fr.mask_last_occurence, fr.proposed_feature, fr.template_feature,
                             fr.proposal_score, max_iter=12, proj_iter=4, is_test=is_test)
    g = forward([fr], 12, 4, is_test)
    check_frame(g, 0, o, is_test, P, O)


@pytest.mark.parametrize("k", range(len(WIDE_CASES)))
def test_g19_reference_output_at_wide_shapes(k):
    """First hand: the reference's own MatchModel output (tests/golden/g19_wide_tables.npz)."""
    gold = golden("g19_wide_tables").group(f"c{k}")
    P, O, is_test = WIDE_CASES[k]
    fr = synth.make_frame(P, O, 24, 24, 64, seed=1900 + k, kind="uniform")
    g = forward([fr], 12, 4, is_test)
    assert int(g["iters"][0]) + 1 == int(gold["n_xlist"])
    for key in ("sim", "R", "Rb", "match_score", "det_score"):
        assert np.array_equal(g[key][0], gold[key]), key
    assert np.array_equal(g["R"][0].argmax(1), gold["argmax"])
    s = g["full_outmask"][0].astype(np.float64).reshape(O, -1).sum(1)
    assert np.allclose(s, gold["outmask_sum"], rtol=1e-6, atol=1e-3)


def test_wide_ragged_batch_dead_frames_and_single_proposal():
    """One launch over frames with different live counts inside a wide table: a full frame, ONE live proposal (the inner-sum
    form of the similarity), a frame without proposals, one without templates, a narrow live block inside the wide table."""
    P, O, H, W, D = 300, 40, 16, 16, 64
    frames = [synth.make_frame(P, O, H, W, D, seed=7700 + b, kind="uniform") for b in range(5)]
    nv, mv = [300, 1, 0, 257, 9], [40, 35, 40, 0, 3]
    for is_test in (1, 0):
        g = forward(frames, 10, 3, is_test, n_valid=torch.tensor(nv, dtype=torch.int32, device=DEV),
                    m_valid=torch.tensor(mv, dtype=torch.int32, device=DEV))
        for b, fr in enumerate(frames):
            if nv[b] == 0 or mv[b] == 0:
                assert int(g["iters"][b]) == 0
                for key in ("sim", "R", "Rb", "match_score", "det_score", "full_outmask"):
                    assert not g[key][b].any(), (b, key)
                continue
            o = oracle.match_forward(fr.proposed_mask[:nv[b]], fr.mask_last_occurence[:mv[b]], fr.proposed_feature[:nv[b]],
                                     fr.template_feature[:mv[b]], fr.proposal_score[:nv[b]], max_iter=10, proj_iter=3,
                                     is_test=is_test)
            check_frame(g, b, o, is_test, P, O)


def test_general_kernels_inside_the_envelope_agree_with_the_fast_ones():
    """Option FORCE_WIDE sends dmm_match_forward through the general kernels at shapes the fast kernels cover: same tables, scores,
    iteration counts (data-dependent exits included: structured frames converge early) and -- same arithmetic in the mix --
    the same masks bit for bit in both modes; and both equal the oracle."""
    cases = [(1, 1, 5, 7, 16, "uniform", 6, 3), (9, 8, 17, 13, 100, "uniform", 6, 3), (33, 5, 20, 20, 64, "uniform", 6, 3),
             (50, 10, 32, 32, 512, "structured", 80, 5), (7, 32, 8, 8, 8, "uniform", 6, 3), (255, 31, 6, 6, 48, "uniform", 6, 3),
             (12, 6, 24, 24, 64, "structured", 100, 5), (200, 20, 16, 16, 512, "uniform", 20, 5)]
    exits = 0
    for k, (P, O, H, W, D, kind, it, pj) in enumerate(cases):
        fr = synth.make_frame(P, O, H, W, D, seed=9300 + k, kind=kind)
        for is_test in (1, 0):
            fast = forward([fr], it, pj, is_test)
            with _lib.options(FORCE_WIDE=1):
                wide = forward([fr], it, pj, is_test)
            for key in fast:
                assert np.array_equal(fast[key], wide[key]), (P, O, is_test, key)
            o = oracle.match_forward(fr.proposed_mask, fr.mask_last_occurence, fr.proposed_feature, fr.template_feature,
                                     fr.proposal_score, max_iter=it, proj_iter=pj, is_test=is_test)
            check_frame(wide, 0, o, is_test, P, O)
            exits += int(wide["iters"][0]) < it
    assert exits >= 2, "no case exercised the data-dependent exits"
    # 16-bit planes through the general mix: same result as the fast mix
    fr = synth.make_frame(40, 6, 20, 20, 64, seed=9400, kind="uniform")
    for dt in (torch.float16, torch.bfloat16):
        res = []
        for wide in (False, True):
            with _lib.options(FORCE_WIDE=int(wide)):
                out = ops.match_forward(dev(fr.proposed_mask)[None].to(dt), dev(fr.mask_last_occurence)[None].to(dt),
                                        dev(fr.proposed_feature)[None], dev(fr.template_feature)[None],
                                        dev(fr.proposal_score)[None], score_weight=0.3, max_iter=8, proj_iter=3, lr=0.1,
                                        is_test=0)
            res.append([t.clone() for t in out])
        assert all(torch.equal(a, c) for a, c in zip(*res)), dt


def test_granular_calls_compose_to_the_same_result_at_wide_tables():
    """counts -> normalise -> cosine -> relax_match (dmm_relax_match_any_f32, scratch allocated by the wrapper) -> mix, the
    way the per-video driver chains them: the same tables as the fused forward and the oracle."""
    for (P, O) in [(300, 40), (20, 50), (1, 40)]:
        fr = synth.make_frame(P, O, 24, 24, 64, seed=9600 + P, kind="uniform")
        for is_test in (1, 0):
            pm, tm = dev(fr.proposed_mask)[None], dev(fr.mask_last_occurence)[None]
            inter, ap, at = ops.iou_counts(pm, tm)
            cos = ops.cosine(ops.feature_normalize(dev(fr.template_feature)[None]),
                             ops.feature_normalize(dev(fr.proposed_feature)[None]))
            r = ops.relax_match(cos, inter, ap, at, dev(fr.proposal_score)[None], score_weight=0.3, max_iter=12, proj_iter=4,
                                lr=0.1, is_test=is_test, want_x=True)
            full = ops.mask_mix(r["Rb"], pm)
            g = forward([fr], 12, 4, is_test)
            for key in ("sim", "R", "Rb", "match_score", "det_score", "iters"):
                assert np.array_equal(r[key].cpu().numpy(), g[key]), (P, O, key)
            assert np.array_equal(full.cpu().numpy(), g["full_outmask"])
            o = oracle.relax(-np.pad(g["sim"][0], ((0, 0), (0, max(P, O + 1) - P))), 12, 4, 0.1)
            assert np.array_equal(r["X"][0].cpu().numpy(), o["X"]) and int(r["iters"][0]) == o["iters"]


def test_dmm_model_inference_with_wide_tables():
    """The per-video driver (dmm_model.py:48-158 counterpart) on 2 videos with 300 / 270 proposals and 40 template slots:
    every video against the oracle's single-frame forward."""
    from dmm_net_amd.dmm_model import DMM_Model
    from test_gpu_parity import _Props
    O, H, W, D = 40, 20, 20, 64
    counts = [300, 270]
    frames = [synth.make_frame(P, O, H, W, D, seed=9700 + b, kind="uniform") for b, P in enumerate(counts)]
    feats = torch.cat([dev(fr.proposed_feature) for fr in frames], 0)
    model = DMM_Model({"matching": {"algo": "relax"}, "relax_max_iter": 8, "relax_proj_iter": 3, "relax_learning_rate": 0.1,
                       "score_weight": 0.3}, is_test=1, feature_extractor=lambda bf, props: feats)
    props = [_Props(dev(fr.proposed_mask).unsqueeze(1), dev(fr.proposal_score)) for fr in frames]
    tplt = {b: {"feat": [dev(fr.template_feature)]} for b, fr in enumerate(frames)}
    valid = torch.ones((2, O), device=DEV)
    ml = dev(np.stack([fr.mask_last_occurence for fr in frames]))
    with torch.no_grad():
        out, _, losses, last = model.inference({"args": None, "shape": None, "extra_frame": [0, 0], "valid": valid}, props,
                                               None, ml, tplt)
    for b, fr in enumerate(frames):
        o = oracle.match_forward(fr.proposed_mask, fr.mask_last_occurence, fr.proposed_feature, fr.template_feature,
                                 fr.proposal_score, max_iter=8, proj_iter=3, is_test=1)
        assert np.array_equal(out[b].cpu().numpy().reshape(O, -1), o["full_outmask"].reshape(O, -1)), b


def test_matchmodel_dropin_takes_wide_tables_at_inference():
    """The reference's call -- MatchModel(cfgs, is_test=1)(features, masks, [template features], template masks, scores) --
    with 300 proposals x 40 templates: the fused forward's general kernels behind the same nn.Module signature."""
    from dmm_net_amd.match_model import MatchModel
    P, O = 300, 40
    fr = synth.make_frame(P, O, 24, 24, 64, seed=9500, kind="uniform")
    model = MatchModel({"matching": {"algo": "relax"}, "relax_max_iter": 12, "relax_proj_iter": 4,
                        "relax_learning_rate": 0.1, "score_weight": 0.3}, 1)
    with torch.no_grad():
        fo, ms, ds, fo2, loss = model(dev(fr.proposed_feature), dev(fr.proposed_mask), [dev(fr.template_feature)],
                                      dev(fr.mask_last_occurence), dev(fr.proposal_score))
    o = oracle.match_forward(fr.proposed_mask, fr.mask_last_occurence, fr.proposed_feature, fr.template_feature,
                             fr.proposal_score, max_iter=12, proj_iter=4, is_test=1)
    assert fo2 is fo and loss == {}
    assert np.array_equal(fo.cpu().numpy(), o["full_outmask"].reshape(O, 24, 24))
    assert np.array_equal(ms.cpu().numpy(), o["match_score"]) and np.array_equal(ds.cpu().numpy(), o["det_score"])


def _cfg(mi, pi):
    return {"matching": {"algo": "relax"}, "relax_max_iter": mi, "relax_proj_iter": pi, "relax_learning_rate": 0.1,
            "score_weight": 0.3}


@pytest.mark.parametrize("k", range(4))
def test_g20_training_at_wide_tables_matches_the_reference_autograd(k):
    """VERDICT r3 missing #2: the backward used to answer DMM_ERR_UNSUPPORTED beyond 32 templates / 256 solver columns
    while the reference's autograd is unbounded.  G20 = the reference's OWN gradients (imported MatchModel, CPU autograd)
    at 300 x 40, 20 x 50 (pad path), 257 x 33 (test mode) and 40 x 10 with 1100 outer iterations (beyond the register
    kernel's tape index): the drop-in module trains there now -- general solver backward (dmm_wide.hip), general mix
    backward, greedy one-hot of the matching loss at any size -- within the bound the inside-envelope gradients are held
    to (2e-5 of the largest entry)."""
    from dmm_net_amd.match_model import MatchModel
    g = golden("g20_wide_gradients")
    assert int(g["n"]) == 4
    c = g.group(f"c{k}")
    P, O, H, W, D, it, pj, is_test, seed = [int(v) for v in c["shape"]]
    fr = synth.make_frame(P, O, H, W, D, seed=seed, kind="structured", with_targets=True)
    assert fr.checksum() == str(c["checksum"])
    model = MatchModel(_cfg(it, pj), is_test)
    pf = dev(fr.proposed_feature).requires_grad_(True)
    tf = dev(fr.template_feature).requires_grad_(True)
    fo, ms, ds, _, loss = model(pf, dev(fr.proposed_mask), [tf], dev(fr.mask_last_occurence), dev(fr.proposal_score),
                                dev(fr.targets))
    assert np.abs(ms.detach().cpu().numpy() - c["match_score"]).max() <= 1e-6
    assert np.abs(ds.detach().cpu().numpy() - c["det_score"]).max() <= 1e-6
    assert abs(float(loss["cost_loss"]) - float(c["cost_loss"])) <= 1e-6
    total = (fo * dev(c["wmask"])).sum() + (ms * dev(c["wms"])).sum() + (ds * dev(c["wds"])).sum() + 3.0 * loss["cost_loss"]
    assert abs(float(total.detach()) - float(c["total"])) < 1e-3 * max(1.0, abs(float(c["total"])))
    total.backward()
    from conftest import record_achieved
    for name, mine, ref in (("pf", pf.grad.cpu().numpy(), c["grad_pf"]), ("tf", tf.grad.cpu().numpy(), c["grad_tf"])):
        scale = max(float(np.abs(ref).max()), 1e-12)
        err = float(np.abs(mine - ref).max())
        record_achieved(f"g20_wide_backward/c{k}/{name}_rel_err", err / scale)
        assert err <= 2e-5 * scale + 1e-7, (k, name, err, scale)


def test_general_backward_inside_the_envelope_agrees_with_the_register_kernel():
    """Option FORCE_WIDE sends the solver backward, the mix backward and the forward through the general kernels at shapes
    the fast kernels cover too: same forward bit for bit, gradients within the accumulation-order bound (the two backward
    kernels reduce in different orders)."""
    from dmm_net_amd.match_model import MatchModel
    for (P, O, it, pj, is_test, seed) in [(8, 3, 10, 5, 0, 5), (50, 10, 20, 5, 0, 6), (3, 5, 10, 5, 0, 7), (64, 16, 12, 3, 1, 8),
                                          (12, 6, 100, 5, 0, 9)]:
        fr = synth.make_frame(P, O, 32, 32, 64, seed=2100 + seed, kind="structured", with_targets=True)
        res = []
        for wide in (0, 1):
            with _lib.options(FORCE_WIDE=wide):
                model = MatchModel(_cfg(it, pj), is_test)
                pf = dev(fr.proposed_feature).requires_grad_(True)
                tf = dev(fr.template_feature).requires_grad_(True)
                fo, ms, ds, _, loss = model(pf, dev(fr.proposed_mask), [tf], dev(fr.mask_last_occurence),
                                            dev(fr.proposal_score), dev(fr.targets))
                w = torch.rand(fo.shape, generator=torch.Generator().manual_seed(3)).to(DEV)
                ((fo * w).sum() + ms.sum() + 0.5 * ds.sum() + 3.0 * loss["cost_loss"]).backward()
                res.append((fo.detach().clone(), ms.detach().clone(), pf.grad.clone(), tf.grad.clone()))
        assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]), (P, O)
        for a, b_ in ((res[0][2], res[1][2]), (res[0][3], res[1][3])):
            scale = max(float(a.abs().max()), 1e-12)
            assert float((a - b_).abs().max()) <= 2e-5 * scale + 1e-7, (P, O, float((a - b_).abs().max()), scale)
