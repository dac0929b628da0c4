"""The training call as ONE library call each way (include/dmm_match.h (5d) / (5e) / (1e)).

``dmm_match_train_forward`` / ``_backward`` chain the same kernels as the granular entries, so everything they return must
equal the granular chain BIT FOR BIT (only the loss tail is new arithmetic: checked against the oracle, which is pinned on
the reference by G2 / G3 / G17 / G20, and against the granular tensor-op form).  The reference's own autograd numbers for
the same path are the goldens G6 / G10 / G17 / G20 in test_gpu_parity.py / test_gpu_features.py, which run THROUGH these
entries since they are the default.
"""
import numpy as np
import pytest
import torch

import oracle
from dmm_net_amd import _lib, autograd, ops, synth
from dmm_net_amd.match_model import MatchModel

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(DEV)


def cfg(max_iter, proj_iter, lr=0.1, w=0.3):
    return {"matching": {"algo": "relax"}, "relax_max_iter": max_iter, "relax_proj_iter": proj_iter,
            "relax_learning_rate": lr, "score_weight": w}


class granular:
    """Pin the pre-fusion chain (separate library calls + tensor ops for the loss tail)."""

    def __enter__(self):
        self.old = autograd._FUSED_TRAIN
        autograd._FUSED_TRAIN = False

    def __exit__(self, *exc):
        autograd._FUSED_TRAIN = self.old
        return False


def batch(B, N, M, H, W, D, seed, dtype=torch.float32, kind="structured"):
    frs = [synth.make_frame(N, M, H, W, D, seed=seed + 17 * b, kind=kind, with_targets=True) for b in range(B)]
    st = lambda k, dt=None: dev(np.stack([getattr(f, k) for f in frs], 0), dt)
    return dict(pf=st("proposed_feature"), tf=st("template_feature"), pm=st("proposed_mask", dtype),
                tm=st("mask_last_occurence", dtype), sc=st("proposal_score"), tg=st("targets", dtype), frames=frs)


def run_layer(d, *, n_valid=None, m_valid=None, with_targets=True, is_test=0, max_iter=10, proj_iter=5, pm=None, seed=0):
    pf = d["pf"].clone().requires_grad_(True)
    tf = d["tf"].clone().requires_grad_(True)
    full, ms, ds, loss, iters = autograd.match_layer_batched(
        pf, d["pm"] if pm is None else pm, tf, d["tm"], d["sc"], d["tg"] if with_targets else None, n_valid, m_valid,
        score_weight=0.3, max_iter=max_iter, proj_iter=proj_iter, lr=0.1, is_test=is_test)
    g = torch.Generator(device=DEV).manual_seed(1234 + seed)
    w_full = torch.rand(full.shape, generator=g, device=DEV)
    w_ms = torch.rand(ms.shape, generator=g, device=DEV)
    w_ds = torch.rand(ds.shape, generator=g, device=DEV)
    obj = (full * w_full).sum() + (ms * w_ms).sum() + (ds * w_ds).sum() + 2.0 * loss.sum()
    obj.backward()
    torch.cuda.synchronize()
    return [t.detach().cpu().numpy() for t in (full, ms, ds, loss, iters, pf.grad, tf.grad)]


NAMES = ("full_outmask", "match_score", "det_score", "cost_loss", "iters", "d proposed_feature", "d template_feature")


def assert_same(a, b):
    """Bit for bit -- except cost_loss: the mse tail is the one piece of new arithmetic (a workgroup tree sum instead of
    torch's mean kernel): last-ulp agreement.  Its GRADIENT does not depend on the sum's order and stays bit exact."""
    for name, x, y in zip(NAMES, a, b):
        assert x.shape == y.shape, name
        if name == "cost_loss":
            assert np.all(np.abs(x.astype(np.float64) - y.astype(np.float64)) <= 2e-7 * np.maximum(1.0, np.abs(y))), (x, y)
            continue
        if name.startswith("d "):
            # gradients: dmm_mask_mix_bwd adds the workgroups' partial sums of a frame into dRb with float atomics, in
            # arrival order (both chains run the SAME kernels; two runs of either may differ in the last bits)
            scale = float(np.abs(y).max()) or 1.0
            assert float(np.abs(x.astype(np.float64) - y.astype(np.float64)).max()) <= 1e-5 * scale, name
            continue
        assert np.array_equal(x, y), (name, float(np.abs(x.astype(np.float64) - y.astype(np.float64)).max()))


@pytest.mark.parametrize("B,N,M,H,W,D", [(1, 50, 5, 64, 72, 512),      # the trainer's call: front kernel with the targets
                                         (4, 50, 5, 33, 47, 512),      # ... for a handful of frames
                                         (1, 10, 8, 40, 40, 512),      # 2 x 8 rows: the front kernel's widest dual tile
                                         (3, 50, 10, 40, 56, 512),     # 2 x 10 rows: lanes similarity + dual count pass
                                         (16, 50, 10, 31, 29, 512),    # more frames than the front kernel takes
                                         (2, 40, 20, 24, 40, 512),     # > 16 rows per set: the targets in a pass of their own
                                         (2, 30, 4, 20, 20, 256),      # D the lanes kernel takes, the front kernel does not
                                         (2, 30, 4, 20, 20, 96),       # D neither takes: normalise + cosine
                                         (2, 3, 5, 16, 16, 512),       # P <= O: the padded solver width
                                         (1, 1, 1, 9, 9, 64),          # one proposal, one template
                                         (2, 130, 6, 20, 24, 512),     # more than one wave of columns
                                         (2, 50, 10, 255, 255, 512),   # BASELINE configs[1]'s frame at full size
                                         (1, 50, 5, 255, 448, 512)])   # the product's frame (bench.py --config dropin)
def test_fused_training_call_equals_the_granular_chain_bit_for_bit(B, N, M, H, W, D):
    d = batch(B, N, M, H, W, D, seed=300 + N + M)
    with granular():
        ref = run_layer(d)
    got = run_layer(d)
    assert_same(got, ref)
    assert float(np.abs(got[5]).sum()) > 0 and float(np.abs(got[6]).sum()) > 0


@pytest.mark.parametrize("is_test,with_targets", [(1, True), (0, False), (1, False)])
def test_fused_call_without_targets_and_in_test_mode(is_test, with_targets):
    d = batch(3, 50, 5, 40, 40, 512, seed=77)
    with granular():
        ref = run_layer(d, is_test=is_test, with_targets=with_targets)
    assert_same(run_layer(d, is_test=is_test, with_targets=with_targets), ref)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_fused_call_on_16_bit_planes(dtype):
    d = batch(2, 50, 5, 40, 44, 512, seed=5, dtype=dtype)
    with granular():
        ref = run_layer(d)
    assert_same(run_layer(d), ref)


def test_fused_call_on_ragged_batches_and_per_frame_plane_tables():
    B, N, M, H, W, D = 5, 40, 6, 32, 36, 512
    d = batch(B, N, M, H, W, D, seed=11)
    counts_n, counts_m = [40, 17, 1, 33, 40], [6, 3, 0, 1, 5]
    nv = torch.tensor(counts_n, dtype=torch.int32, device=DEV)
    mv = torch.tensor(counts_m, dtype=torch.int32, device=DEV)
    with granular():
        ref = run_layer(d, n_valid=nv, m_valid=mv)
    got = run_layer(d, n_valid=nv, m_valid=mv)
    assert_same(got, ref)
    assert got[3][2] == 0.0                                           # a frame without live templates has no loss
    # one tensor per frame (DMM_Model's per-video proposal planes): the pointer-table forms inside the same entries
    planes = [d["pm"][b, :counts_n[b]].clone() for b in range(B)]
    with granular():
        ref2 = run_layer(d, m_valid=mv, pm=planes)
    assert_same(run_layer(d, m_valid=mv, pm=planes), ref2)
    assert_same(ref2[:5], ref[:5])


def test_matching_loss_kernel_against_the_oracle():
    """(1e) alone: gt IoU -> greedy one-hot -> mse, against oracle.matching_loss (pinned on the reference by G2 / G3)."""
    L = _lib.load()
    for (N, M, H, W, seed, kind) in [(8, 3, 64, 64, 1, "structured"), (50, 10, 40, 40, 2, "uniform"), (3, 5, 16, 16, 3, "uniform"),
                                     (1, 1, 8, 8, 4, "uniform"), (200, 20, 24, 24, 5, "structured"), (300, 40, 12, 12, 6, "uniform")]:
        fr = synth.make_frame(N, M, H, W, 16, seed=seed, kind=kind, with_targets=True)
        rng = np.random.default_rng(seed)
        cos = rng.standard_normal((M, N)).astype(np.float32)
        (gi, ap, at), (gi2, at2) = ops.iou_counts_dual(dev(fr.proposed_mask)[None], dev(fr.mask_last_occurence)[None],
                                                       dev(fr.targets)[None])
        gt = torch.empty((1, M, N), dtype=torch.float32, device=DEV)
        loss = torch.empty((1,), dtype=torch.float32, device=DEV)
        cos_d = dev(cos)
        rc = L.dmm_matching_loss_f32(gi2.data_ptr(), ap.data_ptr(), at2.data_ptr(), cos_d.data_ptr(), 1, N, M, None, None,
                                     gt.data_ptr(), loss.data_ptr(), torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        o_loss, _, o_gt = oracle.matching_loss(fr.proposed_mask, fr.targets, cos)
        assert np.array_equal(gt[0].cpu().numpy(), o_gt), (N, M)
        assert abs(float(loss[0]) - o_loss) <= 1e-6 * max(1.0, abs(o_loss)), (float(loss[0]), o_loss)


def test_matching_loss_kernel_ties_and_empty_masks():
    """All-zero masks (every IoU 0: every argmin is a tie -> row i takes ... the reference's first-index rule) and duplicate
    proposals (tied columns), against the oracle's greedy init."""
    L = _lib.load()
    N, M, HW = 6, 4, 64
    rng = np.random.default_rng(0)
    for case in ("zeros", "duplicates"):
        P = np.zeros((N, 8, 8), np.float32)
        T = np.zeros((M, 8, 8), np.float32)
        if case == "duplicates":
            base = (rng.random((8, 8)) > 0.5).astype(np.float32)
            P[:] = base
            T[:] = base
            T[2] = 0
        cos = rng.standard_normal((M, N)).astype(np.float32)
        (gi, ap, at), (gi2, at2) = ops.iou_counts_dual(dev(P)[None], dev(T)[None], dev(T)[None])
        gt = torch.empty((1, M, N), dtype=torch.float32, device=DEV)
        loss = torch.empty((1,), dtype=torch.float32, device=DEV)
        cos_d = dev(cos)
        assert L.dmm_matching_loss_f32(gi2.data_ptr(), ap.data_ptr(), at2.data_ptr(), cos_d.data_ptr(), 1, N, M, None, None,
                                       gt.data_ptr(), loss.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
        o_loss, _, o_gt = oracle.matching_loss(P, T, cos)
        assert np.array_equal(gt[0].cpu().numpy(), o_gt), case
        assert abs(float(loss[0]) - o_loss) <= 1e-6


def test_matchmodel_training_call_takes_the_one_frame_function():
    """The drop-in's training call (dmm_model.py:130-132) goes through ``_MatchFrameFn`` (no batch axis around the fused
    entries): outputs and gradients equal the batched granular chain bit for bit, and the forward the oracle's."""
    fr = synth.make_config_frame(1, kind="structured", with_targets=True)
    t = lambda a: dev(a)

    def call(fused):
        pf = t(fr.proposed_feature).requires_grad_(True)
        tf = t(fr.template_feature).requires_grad_(True)
        model = MatchModel(cfg(10, 5), is_test=0)
        if fused:
            fo, ms, ds, fo2, loss = model(pf, t(fr.proposed_mask), [tf], t(fr.mask_last_occurence), t(fr.proposal_score),
                                          t(fr.targets))
        else:
            with granular():
                fo, ms, ds, fo2, loss = model(pf, t(fr.proposed_mask), [tf], t(fr.mask_last_occurence),
                                              t(fr.proposal_score), t(fr.targets))
        assert fo2 is fo and set(loss) == {"cost_loss"}
        (fo.sum() + 3.0 * loss["cost_loss"] + ms.sum()).backward()
        return [x.detach().cpu().numpy() for x in (fo, ms, ds, loss["cost_loss"], pf.grad, tf.grad)]
    a, b = call(True), call(False)
    for k, (x, y) in enumerate(zip(a, b)):
        assert x.shape == y.shape
        if k == 3:                                                   # cost_loss: last-ulp agreement (see assert_same)
            assert abs(float(x) - float(y)) <= 2e-7 * max(1.0, abs(float(y)))
        elif k >= 4:                                                 # gradients: float atomics across workgroups
            assert float(np.abs(x - y).max()) <= 1e-5 * (float(np.abs(y).max()) or 1.0)
        else:
            assert np.array_equal(x, y)
    o = oracle.match_forward(fr.proposed_mask, fr.mask_last_occurence, fr.proposed_feature, fr.template_feature,
                             fr.proposal_score, max_iter=10, proj_iter=5, is_test=0)
    assert np.array_equal(a[1], o["match_score"]) and np.array_equal(a[2], o["det_score"])
    assert float(np.abs(a[0] - o["full_outmask"]).max()) <= 1e-5
    o_loss = oracle.matching_loss(fr.proposed_mask, fr.targets, o["cos"])[0]
    assert abs(float(a[3]) - o_loss) <= 1e-6


def test_unused_outputs_send_no_gradient_tensors():
    """Only full_outmask feeds the objective: the score / loss gradients arrive as None (no zero-filled stand-ins) and the
    feature gradients equal the granular chain's."""
    d = batch(1, 50, 5, 40, 40, 512, seed=3)

    def grads():
        pf = d["pf"].clone().requires_grad_(True)
        tf = d["tf"].clone().requires_grad_(True)
        full = autograd.match_layer_batched(pf, d["pm"], tf, d["tm"], d["sc"], d["tg"], score_weight=0.3, max_iter=10,
                                            proj_iter=5, lr=0.1, is_test=0)[0]
        full.square().sum().backward()
        return pf.grad.cpu().numpy(), tf.grad.cpu().numpy()
    with granular():
        ref = grads()
    got = grads()
    for a, b in zip(got, ref):
        assert float(np.abs(a - b).max()) <= 1e-5 * (float(np.abs(b).max()) or 1.0)


# ------------------------------------------------------------------------------------ mix backward against float64
def _mix_bwd_case(B, N, M, H, W, seed, dtype=torch.float32, density=0.3, ragged=False):
    g = torch.Generator(device=DEV).manual_seed(seed)
    pm = torch.rand((B, N, H, W), generator=g, device=DEV).to(dtype)
    dout = torch.randn((B, M, H, W), generator=g, device=DEV)
    Pp = ops.padded_width(N, M)
    Rb = torch.rand((B, M, Pp), generator=g, device=DEV)
    Rb = torch.where(torch.rand((B, M, Pp), generator=g, device=DEV) < density, Rb, torch.zeros_like(Rb))
    Rb[:, :, N:] = 0
    nv = mv = None
    if ragged:
        nv = torch.randint(1, N + 1, (B,), generator=g, device=DEV).int()
        mv = torch.randint(0, M + 1, (B,), generator=g, device=DEV).int()
        live = (torch.arange(M, device=DEV)[None, :, None] < mv[:, None, None]) & \
               (torch.arange(Pp, device=DEV)[None, None, :] < nv[:, None, None])
        Rb = torch.where(live, Rb, torch.zeros_like(Rb))
    ref = torch.einsum("bmx,bnx->bmn", dout.double().flatten(2), pm.double().flatten(2))
    want = torch.zeros((B, M, Pp), dtype=torch.float64, device=DEV)
    want[:, :, :N] = ref
    want = torch.where(Rb != 0, want, torch.zeros_like(want))
    return pm, dout, Rb, nv, mv, want


@pytest.mark.parametrize("B,N,M,H,W,dtype,ragged", [
    (2, 50, 10, 64, 64, torch.float32, False),
    (1, 64, 16, 255, 255, torch.float32, False),       # the product's plane size (odd planes, tail)
    (3, 17, 3, 33, 31, torch.float32, False),
    (2, 9, 1, 5, 7, torch.float32, False),             # 35 pixels: the tail step alone
    (4, 50, 5, 40, 44, torch.float32, True),           # ragged frames, dead frames
    (2, 50, 10, 48, 52, torch.float16, False),
    (2, 33, 7, 48, 52, torch.bfloat16, True),
    (2, 65, 4, 24, 24, torch.float32, False), (2, 20, 17, 24, 24, torch.float32, False),
    (2, 120, 32, 24, 24, torch.float32, False), (1, 240, 16, 16, 16, torch.float32, False),    # just past the LDS gate: row kernel
    (2, 112, 32, 24, 24, torch.float32, False), (1, 224, 16, 16, 16, torch.float32, False),    # ON the gate's edge --
    #   56 KB of dense per-wave tables in the union kernel's dynamic block (pairs > 256 at this density) + 3.1 KB static
    (2, 112, 32, 24, 24, torch.float16, True),
    (1, 50, 5, 256, 448, torch.float32, False), (1, 50, 10, 480, 854, torch.float32, False),    # the product's plane sizes
    (1, 20, 4, 1080, 1920, torch.float32, False),
    (19, 50, 10, 40, 52, torch.float32, True)])        # two complete groups of 8 frames (XCD mapping) + 3, ragged
def test_mix_backward_against_the_float64_product(B, N, M, H, W, dtype, ragged):
    """dmm_mask_mix_bwd (union kernel with the scalar-branch slot parking of round 5, and the row kernel it falls back to:
    option MIX_SHARED) against the float64 product on the support of Rb: inside the backward's bound (2e-5 of the largest
    entry); entries outside the support are exactly zero; per-frame plane tables take the same kernels."""
    from conftest import record_achieved
    pm, dout, Rb, nv, mv, want = _mix_bwd_case(B, N, M, H, W, seed=N * 31 + M, dtype=dtype, ragged=ragged)
    scale = float(want.abs().max()) or 1.0
    # union kernel with the waves in lock step / free running, plain frame-major dispatch / one frame per XCD for the union
    # kernels (MIX_XCD bits 2 and 4); the row kernel
    for mode, lock, xcd in ((-1, 1, 3), (-1, 0, 7), (-1, 1, 1), (0, 1, 3)):
        with _lib.options(MIX_SHARED=mode, MIX_SHARED_LOCKSTEP=lock, MIX_XCD=xcd):
            got = ops.mask_mix_bwd(Rb, pm, dout, nv, mv).double()
            if mode == -1 and M <= 32 and N <= 256:
                # the forward union kernel under the same setting: bit for bit the row kernel's result
                fwd = ops.mask_mix(Rb, pm, nv, mv, shared=True)
                with _lib.options(MIX_SHARED=0):
                    fwd_rows = ops.mask_mix(Rb, pm, nv, mv, shared=False)
                assert torch.equal(fwd, fwd_rows), (mode, lock, xcd)
        assert bool((got[Rb == 0] == 0).all())
        err = float((got - want).abs().max()) / scale
        assert err <= 2e-5, (mode, lock, xcd, err)
        if mode == -1 and lock == 1 and xcd == 3:
            record_achieved(f"mix_bwd/{B}x{N}x{M}x{H}x{W}_{str(dtype)[6:]}/rel_err", err)
    fp = ops.FramePlanes([pm[b, :(int(nv[b]) if nv is not None else N)] for b in range(B)])
    t = ops.mask_mix_bwd(Rb, fp, dout, nv if nv is not None else None, mv).double()
    assert float((t - want).abs().max()) / scale <= 2e-5


def test_ragged_frames_get_the_similarity_in_the_order_of_their_own_proposal_count():
    """The one-launch similarity kernel on a RAGGED batch: ATen reduces the [D, P] slab of products in an order that
    depends on P (dmm_torch_order.h: columns below 32 * (P / 32) -- 4 * (P / 4) for P < 8 -- by one cascade chain, the rest by
    the ILP-4 row sum; P = 1 as a contiguous inner sum), and the reference is called per frame with ITS proposals.  Every
    frame's table must equal the oracle's for that frame's own count, bit for bit -- counts chosen so that the class of
    many columns differs from the slot count's (50 slots: columns 0-31 class A; 20 live: none; 9 live: 0-7; 5 live: 0-3)."""
    B, N, M, H, W, D = 6, 50, 5, 12, 12, 512
    d = batch(B, N, M, H, W, D, seed=91)
    counts_n, counts_m = [50, 20, 9, 1, 5, 33], [5, 3, 5, 2, 1, 4]
    nv = torch.tensor(counts_n, dtype=torch.int32, device=DEV)
    mv = torch.tensor(counts_m, dtype=torch.int32, device=DEV)
    got = ops.match_train_forward(d["pm"], d["tm"], None, d["pf"], d["tf"], d["sc"], nv, mv, score_weight=0.3, max_iter=5,
                                  proj_iter=5, lr=0.1, is_test=1)
    assert got is not None
    cos = got[5][:B * M * N].view(B, M, N).cpu().numpy()
    for b in range(B):
        fr = d["frames"][b]
        want = oracle.cosine(fr.template_feature[:counts_m[b]], fr.proposed_feature[:counts_n[b]])
        assert np.array_equal(cos[b, :counts_m[b], :counts_n[b]], want), (b, counts_n[b])
        assert not cos[b, :, counts_n[b]:].any()


def test_matching_loss_kernel_against_the_reference_fixture_g21():
    """dmm_matching_loss_f32 against the reference's own compute_matching_loss tail (fixture G21, first hand): the one-hot
    bit for bit -- empty masks (every argmin a tie), duplicate planes, one live target -- and the loss to the last ulps."""
    from conftest import golden
    g = golden("g21_matching_loss")
    L = _lib.load()
    for k in range(int(g["n"])):
        P, Tg, sim = synth.match_loss_case(k)
        N, M = P.shape[0], Tg.shape[0]
        (gi, ap, at), (gi2, at2) = ops.iou_counts_dual(dev(P)[None], dev(Tg)[None], dev(Tg)[None])
        gt = torch.empty((1, M, N), dtype=torch.float32, device=DEV)
        loss = torch.empty((1,), dtype=torch.float32, device=DEV)
        sim_d = dev(sim)
        assert L.dmm_matching_loss_f32(gi2.data_ptr(), ap.data_ptr(), at2.data_ptr(), sim_d.data_ptr(), 1, N, M, None, None,
                                       gt.data_ptr(), loss.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
        assert np.array_equal(gt[0].cpu().numpy(), g[f"c{k}/gt_matched"]), k
        want = float(g[f"c{k}/loss"])
        assert abs(float(loss[0]) - want) <= 2e-7 * max(1.0, abs(want)), (k, float(loss[0]), want)


def test_ragged_solver_backward_runs_every_frame_in_its_own_row_count():
    """``dmm_relax_match_bwd_f32`` with ``m_valid`` (DMM_Model's batches): each frame goes through the exact-row body of ITS
    template count (``relax_match_bwd_ragged_kernel``) -- the gradient of a frame must equal, bit for bit, what the dense
    launch gives for that frame alone with exactly its live rows and columns."""
    B, N, M = 7, 50, 8
    g = torch.Generator(device=DEV).manual_seed(77)
    sim = torch.randn((B, M, N), generator=g, device=DEV) * 0.2
    sc = torch.rand((B, N), generator=g, device=DEV)
    Pp = ops.padded_width(N, M)
    dRb = torch.rand((B, M, Pp), generator=g, device=DEV)
    dms = torch.rand((B, M), generator=g, device=DEV)
    dds = torch.rand((B, M), generator=g, device=DEV)
    mv = [8, 5, 1, 0, 3, 2, 7]
    nv = [50, 50, 37, 50, 2, 50, 9]
    for is_test in (0, 1):
        kb = dict(max_iter=10, proj_iter=5, lr=0.1, is_test=is_test)
        got = ops.relax_match_bwd(sim, sc, dRb, dms, dds, n_valid=torch.tensor(nv, dtype=torch.int32, device=DEV),
                                  m_valid=torch.tensor(mv, dtype=torch.int32, device=DEV), **kb)
        for b in range(B):
            m, n = mv[b], nv[b]
            if m == 0:
                assert float(got[b].abs().max()) == 0.0
                continue
            pp = ops.padded_width(n, m)
            one = ops.relax_match_bwd(sim[b:b + 1, :m, :n].contiguous(), sc[b:b + 1, :n].contiguous(),
                                      dRb[b:b + 1, :m, :pp].contiguous(), dms[b:b + 1, :m].contiguous(),
                                      dds[b:b + 1, :m].contiguous(), **kb)
            assert torch.equal(got[b, :m, :n], one[0]), (is_test, b, float((got[b, :m, :n] - one[0]).abs().max()))
            assert float(got[b, m:].abs().max() if m < M else 0.0) == 0.0
            assert float(got[b, :, n:].abs().max() if n < N else 0.0) == 0.0


def test_ragged_pad_with_a_gradient_path():
    """``autograd.ragged_pad``: the per-video blocks in one launch, zeros behind every block, and the gradient of a block =
    its rows of the batch's gradient (what the zero-fill + per-video copy loop it replaces gives)."""
    g = torch.Generator(device=DEV).manual_seed(5)
    rows = [50, 0, 13, 50]
    blocks = [torch.randn((r, 512), generator=g, device=DEV, requires_grad=True) for r in rows]
    P = max(rows)
    counts = torch.tensor(rows, dtype=torch.int32, device=DEV)
    out = autograd.ragged_pad(blocks, P, counts)
    assert out.requires_grad and out.shape == (4, P, 512)
    ref = torch.zeros((4, P, 512), device=DEV)
    for b, blk in enumerate(blocks):
        ref[b, :rows[b]] = blk.detach()
    assert torch.equal(out.detach(), ref)
    w = torch.rand(out.shape, generator=g, device=DEV)
    (out * w).sum().backward()
    for b, blk in enumerate(blocks):
        assert torch.equal(blk.grad, w[b, :rows[b]])
    with torch.no_grad():
        assert not autograd.ragged_pad(blocks, P, counts).requires_grad
    # the table handed over by the caller (one upload for all of a call's small tables)
    blks, addr = ops.ragged_blocks([b.detach() for b in blocks])
    cnt, tab = _lib.small_to_device_many([(rows, torch.int32), (addr, torch.int64)], torch.device(DEV))
    assert torch.equal(ops.ragged_pad(blks, P, cnt, tab), ref)


@pytest.mark.parametrize("O", [5, 3])
def test_dmm_model_one_video_takes_the_frame_call_and_equals_the_batched_path(O):
    """``DMM_Model`` with ONE video (the evaluator's batch size) calls ``MatchModel`` on the first O rows like the reference;
    outputs, loss and gradients must equal the ragged batch path's (which G17 pins on the reference's own ``DMM_Model``)."""
    from dmm_net_amd.dmm_model import DMM_Model
    from dmm_net_amd.proposals import SimpleBoxList
    P, F, H, W, D = 50, 5, 64, 72, 512
    fr = synth.make_frame(P, F, H, W, D, seed=900 + O, kind="structured", with_targets=True)
    bl = SimpleBoxList(torch.tensor([[0.0, 0.0, 10.0, 10.0]] * P, device=DEV), (W, H))
    bl.add_field("mask", dev(fr.proposed_mask).unsqueeze(1))
    bl.add_field("scores", dev(fr.proposal_score))
    valid = torch.tensor([[1.0] * O + [0.0] * (F - O)], device=DEV)
    mask_last = dev(fr.mask_last_occurence).unsqueeze(0)
    targets = dev(fr.targets).unsqueeze(0)
    tfeat = dev(fr.template_feature)

    def run(single, is_test):
        pf = dev(fr.proposed_feature).requires_grad_(not is_test)
        dm = DMM_Model(cfg(10, 5), is_test=is_test, feature_extractor=lambda f, pr: pf)
        tplt = {0: {"feat": [tfeat], "refine_input_feat": [()]}}
        if is_test:
            infos = {"extra_frame": [False], "valid": valid}
            if single:
                full, _, _, last = dm.inference(infos, [bl], None, mask_last, tplt)
            else:
                full, _ = dm._match_batch([pf], [bl.get_field("mask").squeeze(1)], [bl.get_field("scores")], [tfeat],
                                          mask_last, [O], None, [False])
                last = full
            return full, last, None, None
        if single:
            full, _, ml, last = dm(None, [bl], None, mask_last, tplt, valid, targets)
        else:
            full, loss = dm._match_batch([pf], [bl.get_field("mask").squeeze(1)], [bl.get_field("scores")], [tfeat],
                                         mask_last, [O], targets, [False])
            ml, last = [loss[0]], full
        (full.sum() * 0.01 + ml[0]).backward()
        return full, last, ml[0], pf.grad

    for is_test in (1, 0):
        a, b_ = run(True, is_test), run(False, is_test)
        assert a[0].shape == (1, F, H, W) and torch.equal(a[0], b_[0]) and torch.equal(a[1], b_[1]), is_test
        assert float(a[0][0, O:].abs().max() if O < F else 0.0) == 0.0
        if not is_test:
            la, lb = float(a[2].detach()), float(b_[2].detach())
            assert abs(la - lb) <= 2e-7 * max(1.0, abs(lb))
            scale = float(b_[3].abs().max())
            assert float((a[3] - b_[3]).abs().max()) <= 1e-5 * scale


@pytest.mark.parametrize("B,N,M,ragged", [(1, 50, 5, False), (3, 50, 10, False), (5, 40, 8, True), (2, 3, 5, False),
                                          (2, 64, 16, False)])
def test_backward_from_the_forwards_tape_equals_the_rerun_bit_for_bit(B, N, M, ragged):
    """The training forward keeps the solver's tape (gate bits per sweep, sweep counts, R); the backward that walks it must
    give exactly what the backward that re-runs the solver gives -- same gates, same arithmetic.  Tables wider than one
    wave are not taped (``taped`` = 0) and always re-run."""
    H, W, D = 24, 28, 512
    d = batch(B, N, M, H, W, D, seed=300 + N + M)
    nv = mv = None
    if ragged:
        nv = torch.tensor([N, N - 7, 1, N, 9][:B], dtype=torch.int32, device=DEV)
        mv = torch.tensor([M, 3, 1, 0, M - 1][:B], dtype=torch.int32, device=DEV)
    g = torch.Generator(device=DEV).manual_seed(9)
    for is_test in (0, 1):
        kw = dict(score_weight=0.3, max_iter=10, proj_iter=5, lr=0.1, is_test=is_test)
        got = ops.match_train_forward(d["pm"], d["tm"], d["tg"], d["pf"], d["tf"], d["sc"], nv, mv, **kw)
        full, ms, ds, loss, iters, saved, taped = got
        assert taped == 1
        d_full = torch.rand(full.shape, generator=g, device=DEV)
        d_ms = torch.rand(ms.shape, generator=g, device=DEV)
        d_ds = torch.rand(ds.shape, generator=g, device=DEV)
        d_loss = torch.rand(loss.shape, generator=g, device=DEV)
        args = (d["pm"], d["pf"], d["tf"], d["sc"], saved, True, d_full, d_ms, d_ds, d_loss, nv, mv, M)
        walked = ops.match_train_backward(*args, iters=iters, taped=taped, **kw)
        rerun = ops.match_train_backward(*args, **kw)
        assert torch.equal(walked[0], rerun[0]) and torch.equal(walked[1], rerun[1]), (is_test, B, N, M)
        assert float(rerun[1].abs().max()) > 0
    wide = batch(1, 100, 5, 16, 16, D, seed=5)
    assert ops.match_train_forward(wide["pm"], wide["tm"], wide["tg"], wide["pf"], wide["tf"], wide["sc"], None, None,
                                   score_weight=0.3, max_iter=10, proj_iter=5, lr=0.1, is_test=0)[6] == 0
