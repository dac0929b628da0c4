"""Frame-loop reductions and the frame loop itself on the GPU."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import oracle
from conftest import golden
from dmm_net_amd import proposals as prop
from dmm_net_amd import _lib, ops, synth, video
from dmm_net_amd.dmm_model import DMM_Model
from dmm_net_amd.roi_features import FeatureExtractor

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_g11_mask_boxes_match_reference_steps():
    g = golden("g11_frame_loop")
    for k in range(int(g["n_box"])):
        O, H, W = [int(v) for v in g[f"box{k}_shape"]]
        m = torch.from_numpy(synth.template_planes(k, O, H, W)).to(DEV)
        boxes, valid = video.mask_boxes(m)
        assert np.array_equal(boxes.cpu().numpy(), g[f"box{k}_boxes"]), k
        assert np.array_equal(valid.cpu().numpy(), g[f"box{k}_valid"]), k
        bl, tv = video.ohw_mask2boxlist(m)                               # utils.py:179-210 return values
        assert bl.size == (W, H) and bl.mode == "xyxy" and tv.dtype == torch.long
        assert torch.equal(bl.get_field("scores"), torch.ones(O, device=DEV)) and bl.get_field("mask") is m
        assert torch.equal(tv, valid.long()) and torch.equal(bl.bbox, boxes)


def test_g11_merge_labels_match_reference_steps():
    g = golden("g11_frame_loop")
    for k in range(int(g["n_mrg"])):
        O, n_obj, H, W = [int(v) for v in g[f"mrg{k}_shape"]]
        outs = torch.from_numpy(synth.refined_planes(k, O, H, W)).to(DEV)[None]
        valid = torch.zeros(1, O, device=DEV)
        valid[0, :n_obj] = 1
        lab = video.merge_labels(outs, valid)
        assert lab.shape == (1, H, W) and lab.dtype == torch.uint8
        assert np.array_equal(lab[0].cpu().numpy(), g[f"mrg{k}_labels"]), k


def test_reductions_vs_oracle_ragged_strided():
    rng = np.random.default_rng(5)
    for (B, O, H, W) in [(3, 5, 31, 17), (2, 1, 1, 1), (4, 7, 64, 65), (1, 12, 255, 255)]:
        big = (rng.random((B, O + 2, H, W)) ** 2).astype(np.float32)
        big[:, :, :, : W // 4] = 0.0
        big[rng.random((B, O + 2)) < 0.3] = 0.0                          # some empty planes
        t = torch.from_numpy(big).to(DEV)
        view = t[:, 1:O + 1]                                             # strided batch / plane view
        ov = rng.integers(0, O + 1, B).astype(np.int32)
        lab = video.merge_labels(view, torch.from_numpy(ov).to(DEV))
        exp = oracle.merge_labels(big[:, 1:O + 1].reshape(B, O, H * W), ov).reshape(B, H, W)
        assert np.array_equal(lab.cpu().numpy(), exp)
        for thresh in (0.0, 0.5):
            boxes, valid = video.mask_boxes(view.reshape(B * O, H, W), thresh)
            eb, ev = oracle.mask_boxes(big[:, 1:O + 1].reshape(B * O, H, W), thresh)
            assert np.array_equal(boxes.cpu().numpy(), eb) and np.array_equal(valid.cpu().numpy(), ev)
    # no live template -> all background; empty batch
    z = video.merge_labels(torch.rand(2, 3, 50, device=DEV), torch.zeros(2, dtype=torch.int32, device=DEV))
    assert int(z.sum()) == 0
    assert video.merge_labels(torch.rand(0, 3, 50, device=DEV)).shape == (0, 50)
    assert video.mask_boxes(torch.rand(0, 4, 4, device=DEV))[0].shape == (0, 4)


class _PoolEncoder:
    """Batch-independent toy encoder (pooling only): 4 levels, C channels, strides 4..32."""

    def __init__(self, C=8):
        self.mul = torch.linspace(0.5, 1.5, C, device=DEV).view(1, C, 1, 1)

    def __call__(self, x):
        g = x.mean(1, keepdim=True)
        lv = tuple(F.avg_pool2d(g, s, ceil_mode=True) * self.mul for s in (4, 8, 16, 32))
        return {"backbone_feature": lv, "refine_input_feat": lv}


def _raw_proposals(rng, n, H, W):
    x1, y1 = rng.uniform(0, W - 24, n), rng.uniform(0, H - 24, n)
    boxes = np.stack([x1, y1, np.minimum(x1 + rng.uniform(10, 60, n), W - 1), np.minimum(y1 + rng.uniform(10, 50, n), H - 1)], 1)
    bl = prop.SimpleBoxList(torch.from_numpy(boxes.astype(np.float32)), (W, H))      # on the host, like a loaded file
    bl.add_field("scores", torch.from_numpy(rng.random(n).astype(np.float32)))
    bl.add_field("mask", torch.from_numpy((rng.random((n, 1, 28, 28)) * 0.6 + 0.4).astype(np.float32)))
    return bl


def test_frame_loop_batched_equals_per_video_and_writes_labels(tmp_path):
    rng = np.random.default_rng(11)
    B, T, O, H, W = 3, 4, 5, 96, 128
    cfgs = {"matching": {"algo": "relax"}, "relax_max_iter": 40, "relax_proj_iter": 5, "relax_learning_rate": 0.1,
            "score_weight": 0.3}
    frames = torch.randn(B, T, 3, H, W, device=DEV)
    n_frames = [4, 2, 3]                                                 # videos 1, 2 have 'extra' frames in the clip
    n_obj = [2, 0, 4]
    props = [[_raw_proposals(rng, 30 + 5 * b + t, H, W) for t in range(n_frames[b])] for b in range(B)]
    first = torch.zeros(B, O, H, W, device=DEV)
    for b in range(B):
        for o in range(n_obj[b]):
            y0, x0 = int(rng.integers(0, H - 30)), int(rng.integers(0, W - 30))
            first[b, o, y0:y0 + 25, x0:x0 + 28] = 1.0
    first = first.view(B, O, H * W)

    def make_loop():
        return video.FrameLoop(_PoolEncoder(), DMM_Model(cfgs, is_test=1, feature_extractor=FeatureExtractor()),
                               nms_thresh=0.4, max_proposals=20)

    got = {}
    hist = make_loop().run(frames, first, props, n_frames, on_labels=lambda b, t, lab: got.__setitem__((b, t), lab.clone()))
    assert len(hist) == T and hist[0].shape == (B, O, H * W)
    assert sorted(got) == sorted((b, t) for b in range(B) for t in range(n_frames[b]))      # extra frames are not written
    assert torch.equal(hist[0], first)                                   # frame 0 reports the annotation (:119-121)
    exp0 = oracle.merge_labels(first.cpu().numpy(), n_obj).reshape(B, H, W)
    for b in range(B):
        assert np.array_equal(got[(b, 0)].cpu().numpy(), exp0[b])
    assert all(float(h[1].abs().sum()) == 0.0 for h in hist[1:])         # no template -> zeros (dmm_model.py:66-69)
    assert float(hist[1][0, :2].sum()) > 0 and float(hist[1][0, 2:].abs().sum()) == 0.0
    # video 1 runs out of frames after t = 1: its slots are skipped ('extra_frame') from t = 2 on
    assert float(hist[3][2].abs().sum()) == 0.0 or n_frames[2] > 3
    for b in range(B):                                                   # one launch per frame for all videos == per video
        solo = {}
        h1 = make_loop().run(frames[b:b + 1], first[b:b + 1], [props[b]], [n_frames[b]],
                             on_labels=lambda _b, t, lab: solo.__setitem__(t, lab.clone()))
        for t in range(T):
            assert torch.equal(h1[t][0], hist[t][b]), (b, t)
        for t in range(n_frames[b]):
            assert torch.equal(solo[t], got[(b, t)])
    # the reorderings of the loop -- proposal look-ahead on a side stream, several frames per encoder batch, the next
    # chunk's encoder on its own stream -- change no result (the reference's strictly sequential order: all off)
    plain = make_loop()
    plain.lookahead, plain.encode_ahead, plain.encode_overlap = False, 1, False
    for rep in range(3):                                                 # a race would not show every time
        h0 = plain.run(frames, first, props, n_frames)
        h2 = make_loop().run(frames, first, props, n_frames)
        assert all(torch.equal(a, c) and torch.equal(a, d) for a, c, d in zip(hist, h0, h2)), rep
        for (la, ea, eo) in [(True, 1, True), (False, 3, False), (True, 2, True), (False, 4, True)]:
            lp = make_loop()
            lp.lookahead, lp.encode_ahead, lp.encode_overlap = la, ea, eo
            assert all(torch.equal(a, c) for a, c in zip(hist, lp.run(frames, first, props, n_frames))), (rep, la, ea, eo)
    # output format: one palette PNG per frame, read back identically
    Image = pytest.importorskip("PIL.Image")
    f = tmp_path / "merged" / "v0" / "00001.png"
    video.save_label_png(got[(0, 1)], str(f))
    assert np.array_equal(np.array(Image.open(str(f))), got[(0, 1)].cpu().numpy())


def test_ragged_pad_stacks_per_video_blocks_in_one_launch():
    """dmm_ragged_pad: out[b, i] = block_b[i] for i < counts[b], zeros up to P_max -- feature rows (fp32), scores (one
    float per row), packed planes (int64 words), empty videos, and the status codes of bad arguments."""
    rng = np.random.default_rng(3)
    for (tail, dtype) in [((512,), torch.float32), ((1,), torch.float32), ((1017,), torch.int64), ((3, 5), torch.float32),
                          ((2,), torch.bfloat16)]:
        counts = [7, 0, 50, 1, 23]
        blocks = []
        for c in counts:
            a = rng.standard_normal((c,) + tail).astype(np.float32) * 100
            blocks.append(torch.from_numpy(a).to(DEV).to(dtype))
        cd = torch.tensor(counts, dtype=torch.int32, device=DEV)
        for P_max in (50, 64):
            out = ops.ragged_pad(blocks, P_max, cd)
            assert out.shape == (len(counts), P_max) + tail and out.dtype == dtype
            for b, c in enumerate(counts):
                assert torch.equal(out[b, :c], blocks[b]) and not out[b, c:].to(torch.float32).abs().sum().item(), (tail, b)
    L = _lib.load()
    t = torch.zeros(8, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    assert L.dmm_ragged_pad(t.data_ptr(), t.data_ptr(), 2, 4, 6, t.data_ptr(), st) == 1        # row_bytes % 4 != 0
    assert L.dmm_ragged_pad(None, t.data_ptr(), 2, 4, 8, t.data_ptr(), st) == 1                # null table
    assert L.dmm_ragged_pad(t.data_ptr(), t.data_ptr(), 0, 4, 8, t.data_ptr(), st) == 0         # empty batch
    with pytest.raises(ValueError):
        ops.ragged_pad([torch.zeros(2, 3, device=DEV, dtype=torch.float16)], 4, torch.tensor([2], dtype=torch.int32, device=DEV))


def _count_host_syncs(fn):
    """Run fn() counting the tensor methods that round-trip to the host (tolist / item / cpu / numpy on CUDA tensors)."""
    calls = []
    orig = {k: getattr(torch.Tensor, k) for k in ("tolist", "item", "cpu", "numpy")}

    def wrap(name):
        def f(self, *a, **kw):
            if self.is_cuda:
                calls.append(name)
            return orig[name](self, *a, **kw)
        return f
    try:
        for k in orig:
            setattr(torch.Tensor, k, wrap(k))
        out = fn()
    finally:
        for k, v in orig.items():
            setattr(torch.Tensor, k, v)
    return out, calls


def test_two_phase_slots_equal_paste_nms_gather():
    """dmm_proposal_boxes_f32 -> dmm_nms_slots_f32 -> dmm_paste_kept_f32 == paste every raw proposal, NMS + top-k, index the
    kept ones (model_encoder.py:115-134 + boxlist_ops.py:15-29): planes, 1-bit planes, boxes, scores, counts bit for bit,
    in score order; dead slots are marked; frames of a clip are selected by the device scalar."""
    rng = np.random.default_rng(21)
    T, B, H, W, K = 3, 3, 57, 83, 12
    raw = [[_raw_proposals(rng, int(rng.integers(1, 30)), H, W) for _ in range(T)] for _ in range(B)]
    raw[1][2] = _raw_proposals(rng, 1, H, W)
    step = torch.zeros(1, dtype=torch.int32, device=DEV)
    base = torch.tensor([0, 7, 14], dtype=torch.int32, device=DEV)
    for t, Rcap in [(0, 0), (1, 0), (2, 0), (0, 80), (2, 80)]:              # R <= 64: one-wave NMS; above: the block kernel
        clip = prop.ClipProposals.from_boxlists(raw, T, H, W, DEV, R=Rcap)
        slots = prop.ProposalSlots(B, K, H, W, clip.R, DEV)
        step.fill_(t)
        prop.prepare_slots(clip, slots, 0.4, 0.4, 1, step=step, img_base=base)
        ref = prop.forward_mask_prop([raw[b][t].get_field("mask").to(DEV) for b in range(B)],
                                     [raw[b][t].to(DEV) for b in range(B)], 0.4, 1, want_packed=True)
        ref = prop.filter_results(list(ref), 0.4, K, "scores")
        cnt = slots.count.cpu().tolist()
        for b in range(B):
            n = len(ref[b])
            assert cnt[b] == n and n <= K, (t, b)
            assert torch.equal(slots.planes[b, :n], ref[b].get_field("mask").squeeze(1)), (t, b)
            assert torch.equal(slots.packed[b, :n], ref[b].get_field("mask_packed")), (t, b)
            assert torch.equal(slots.boxes[b, :n], ref[b].bbox) and torch.equal(slots.scores[b, :n], ref[b].get_field("scores"))
            rois = slots.rois.view(B, K, 5)[b]
            assert torch.equal(rois[:n, 1:], ref[b].bbox) and bool((rois[:n, 0] == 7 * t + b).all())
            assert bool((rois[n:, 0] == -1).all()) and float(slots.scores[b, n:].abs().sum()) == 0.0


def test_frame_loop_fixed_slots_and_graph_equal_boxlist_path_without_host_syncs():
    """The fixed-slot frame step (two-phase paste, device-side counts, whole step in one HIP graph) reproduces the BoxList
    path bit for bit -- histories and label maps, ragged proposal counts, videos without templates, 'extra' frames, a
    non-prefix template layout, a decoder -- and does not touch the host per frame."""
    rng = np.random.default_rng(12)
    B, T, O, H, W = 3, 6, 5, 96, 128
    cfgs = {"matching": {"algo": "relax"}, "relax_max_iter": 40, "relax_proj_iter": 5, "relax_learning_rate": 0.1,
            "score_weight": 0.3}
    frames = torch.randn(B, T, 3, H, W, device=DEV)
    n_frames = [6, 3, 5]
    props = [[_raw_proposals(rng, 22 + 6 * b + t, H, W) for t in range(n_frames[b])] for b in range(B)]
    first = torch.zeros(B, O, H, W, device=DEV)
    for b, objs in enumerate([(0, 1), (), (0, 2, 3)]):                     # video 2: object 1 empty in frame 0 (non-prefix)
        for o in objs:
            y0, x0 = int(rng.integers(0, H - 30)), int(rng.integers(0, W - 30))
            first[b, o, y0:y0 + 25, x0:x0 + 28] = 1.0
    first = first.view(B, O, H * W)

    def make(slots, graph, refine=None, **kn):
        lp = video.FrameLoop(_PoolEncoder(), DMM_Model(cfgs, is_test=1, feature_extractor=FeatureExtractor()),
                             refine=refine, nms_thresh=0.4, max_proposals=20)
        lp.slots, lp.graph = slots, graph
        for k, v in kn.items():
            setattr(lp, k, v)
        return lp

    def run(lp, fr=frames, fi=first, pr=props, nf=n_frames):
        labs = {}
        h = lp.run(fr, fi, pr, nf, on_labels=lambda b, t, lab: labs.__setitem__((b, t), lab.clone()))
        return [x.clone() for x in h], labs

    ref_h, ref_l = run(make(False, False))
    for (slots, graph, kn) in [(True, False, {}), (True, True, {}), (True, True, dict(encode_ahead=1, encode_overlap=False)),
                               (True, True, dict(encode_ahead=3)), (True, True, dict(fuse_epilogue=False)),
                               (True, False, dict(fuse_epilogue=False, encode_ahead=2))]:
        lp = make(slots, graph, **kn)
        for rep in range(3):                                             # replays of the captured step; races
            h, l = run(lp)
            assert sorted(l) == sorted(ref_l)
            assert all(torch.equal(a, c) for a, c in zip(ref_h, h)), (slots, graph, kn, rep)
            assert all(torch.equal(ref_l[k], l[k]) for k in ref_l), (slots, graph, kn, rep)
    # the same plan on another clip of the same shape (graph reuse), and on a shorter one
    lp = make(True, True)
    run(lp)
    props2 = [[_raw_proposals(rng, 25 + t, H, W) for t in range(T)] for b in range(B)]
    fr2 = torch.randn(B, T, 3, H, W, device=DEV)
    h2, l2 = run(lp, fr2, first, props2, [T] * B)
    r2, rl2 = run(make(False, False), fr2, first, props2, [T] * B)
    assert all(torch.equal(a, c) for a, c in zip(r2, h2)) and all(torch.equal(rl2[k], l2[k]) for k in rl2)
    h3, _ = run(lp, fr2[:, :4], first, props2, [4] * B)
    assert all(torch.equal(a, c) for a, c in zip(r2[:4], h3))
    # a decoder (stand-in: blends the matched masks with the previous prediction and keeps a running state)
    def refine(features, prev_mask, y_mask, init_pred, hist_new, valid, state):
        Bq, Oq = init_pred.shape[:2]
        outs = 0.75 * init_pred.reshape(Bq, Oq, -1) + 0.25 * prev_mask + 0.0 * y_mask
        state = outs.mean() if state is None else state + outs.mean()
        return outs, hist_new * 0.5 + 0.5 * outs.view_as(hist_new), state
    rh, rl = run(make(False, False, refine=refine))
    for fe_ in (True, False):
        gh, gl = run(make(True, True, refine=refine, fuse_epilogue=fe_))
        assert all(torch.equal(a, c) for a, c in zip(rh, gh)) and all(torch.equal(rl[k], gl[k]) for k in rl), fe_
    # no host round trip per frame: the count does not grow with the clip length
    lp = make(True, True)
    run(lp)                                                              # capture
    quiet = lambda b, t, lab: None
    _, c6 = _count_host_syncs(lambda: lp.run(frames, first, props, n_frames, on_labels=quiet))
    _, c3 = _count_host_syncs(lambda: lp.run(frames[:, :3], first, props, [3, 3, 3], on_labels=quiet))
    assert len(c6) == len(c3) <= 2, (c6, c3)
    _, old = _count_host_syncs(lambda: make(False, False).run(frames, first, props, n_frames, on_labels=quiet))
    assert len(old) >= T


def test_frame_loop_prefetch_of_the_next_clip_changes_no_result():
    """``run(next_frames=...)`` issues the next clip's first encoder chunk under this clip's last steps.  A walk over three
    clips (different lengths, one with a decoder-free static-output encoder = a captured graph) gives the same histories
    and labels with and without it; a prefetch for frames that are then NOT the next clip is ignored."""
    from dmm_net_amd.encoder import GraphedEncoder
    rng = np.random.default_rng(21)
    B, O, H, W = 2, 4, 96, 128
    cfgs = {"matching": {"algo": "relax"}, "relax_max_iter": 40, "relax_proj_iter": 5, "relax_learning_rate": 0.1,
            "score_weight": 0.3}
    first = torch.zeros(B, O, H, W, device=DEV)
    for b in range(B):
        for o in range(2 + b):
            y0, x0 = int(rng.integers(0, H - 30)), int(rng.integers(0, W - 30))
            first[b, o, y0:y0 + 25, x0:x0 + 28] = 1.0
    first = first.view(B, O, H * W)
    clips = []
    for T in (5, 9, 3):
        clips.append((torch.randn(B, T, 3, H, W, device=DEV),
                      [[_raw_proposals(rng, 20 + t + 3 * b, H, W) for t in range(T)] for b in range(B)]))

    class _StaticPool(_PoolEncoder):                                     # outputs alias fixed buffers, like a replayed graph
        static_outputs = True

        def __call__(self, x):
            out = super().__call__(x)
            key = tuple(x.shape)
            buf = self.__dict__.setdefault("buf", {})
            if key not in buf:
                buf[key] = {k: tuple(torch.empty_like(v) for v in vs) for k, vs in out.items()}
            for k, vs in out.items():
                for d, v in zip(buf[key][k], vs):
                    d.copy_(v)
            return buf[key]

    for enc_cls in (_PoolEncoder, _StaticPool):
        def make():
            return video.FrameLoop(enc_cls(), DMM_Model(cfgs, is_test=1, feature_extractor=FeatureExtractor()),
                                   nms_thresh=0.4, max_proposals=20)

        def walk(lp, prefetch, wrong=False):
            res = []
            for i, (fr, pr) in enumerate(clips):
                nxt = None
                if prefetch and i + 1 < len(clips):
                    nxt = clips[i + 1][0] if not wrong else clips[0][0]
                labs = {}
                h = lp.run(fr, first, pr, on_labels=lambda b, t, lab: labs.__setitem__((b, t), lab.clone()), next_frames=nxt)
                res.append(([x.clone() for x in h], labs))
            return res
        ref = walk(make(), False)
        for rep in range(3):
            for wrong in (False, True):
                got = walk(make(), True, wrong)
                for (rh, rl), (gh, gl) in zip(ref, got):
                    assert len(rh) == len(gh) and all(torch.equal(a, c) for a, c in zip(rh, gh)), (enc_cls.__name__, rep, wrong)
                    assert sorted(rl) == sorted(gl) and all(torch.equal(rl[k], gl[k]) for k in rl)
    lp = make()
    lp.run(clips[0][0], first, clips[0][1], next_frames=clips[1][0])
    assert lp._prefetched is not None                                    # ... and it is really taken by the next run
    lp.run(clips[1][0], first, clips[1][1])
    assert lp._prefetched is None


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_frame_step_fuzz_fixed_slots_equal_boxlist_path(seed):
    """Random clips -- image sizes with H*W not a multiple of 4 / 256, 1..8 template slots with random empty objects
    (non-prefix layouts, videos without templates), 1..70 raw proposals per frame (above 64 the general NMS kernel), top-k
    below / at / above the kept count, random NMS thresholds and clip lengths per video -- through the fixed-slot step
    (graph replay, fused and unfused epilogue) and the BoxList path: histories and label maps bit for bit."""
    rng = np.random.default_rng(1000 + seed)
    B, T, O = int(rng.integers(1, 5)), int(rng.integers(2, 7)), int(rng.integers(1, 9))
    H, W = [(37, 53), (64, 96), (51, 50), (96, 128), (33, 47), (80, 45)][seed % 6]
    Rmax = int(rng.integers(1, 71))
    K = int(rng.choice([3, 10, 20, Rmax]))
    nms_t = float(rng.choice([0.2, 0.4, 0.7]))
    cfgs = {"matching": {"algo": "relax"}, "relax_max_iter": int(rng.integers(1, 30)), "relax_proj_iter": int(rng.integers(1, 6)),
            "relax_learning_rate": 0.1, "score_weight": 0.3}
    n_frames = [int(rng.integers(1, T + 1)) for _ in range(B)]
    n_frames[0] = T

    def rawp(n):
        x1, y1 = rng.uniform(0, W - 12, n), rng.uniform(0, H - 12, n)
        bx = np.stack([x1, y1, np.minimum(x1 + rng.uniform(4, W * 0.6, n), W - 1), np.minimum(y1 + rng.uniform(4, H * 0.6, n), H - 1)], 1)
        bl = prop.SimpleBoxList(torch.from_numpy(bx.astype(np.float32)), (W, H))
        bl.add_field("scores", torch.from_numpy(np.round(rng.random(n), 2).astype(np.float32)))    # ties on purpose
        bl.add_field("mask", torch.from_numpy((rng.random((n, 1, 28, 28)) * 0.8 + 0.2).astype(np.float32)))
        return bl
    props = [[rawp(int(rng.integers(1, Rmax + 1))) for _ in range(n_frames[b])] for b in range(B)]
    frames = torch.randn(B, T, 3, H, W, device=DEV)
    first = torch.zeros(B, O, H, W, device=DEV)
    for b in range(B):
        for o in range(O):
            if rng.random() < 0.65:
                y0, x0 = int(rng.integers(0, H - 10)), int(rng.integers(0, W - 10))
                first[b, o, y0:y0 + int(rng.integers(4, 12)), x0:x0 + int(rng.integers(4, 12))] = 1.0
    first = first.view(B, O, H * W)

    def run(slots, graph, fe_=True):
        lp = video.FrameLoop(_PoolEncoder(), DMM_Model(cfgs, is_test=1, feature_extractor=FeatureExtractor()),
                             nms_thresh=nms_t, max_proposals=K)
        lp.slots, lp.graph, lp.fuse_epilogue = slots, graph, fe_
        labs = {}
        h = lp.run(frames, first, props, n_frames, on_labels=lambda b, t, lab: labs.__setitem__((b, t), lab.clone()))
        return [x.clone() for x in h], labs
    ref_h, ref_l = run(False, False)
    for (graph, fe_) in [(True, True), (True, False), (False, True)]:
        h, l = run(True, graph, fe_)
        assert sorted(l) == sorted(ref_l)
        assert all(torch.equal(a, c) for a, c in zip(ref_h, h)), (seed, graph, fe_)
        assert all(torch.equal(ref_l[k], l[k]) for k in ref_l), (seed, graph, fe_)


def _clip_case(seed, B, T, O, H, W, objs, n_frames, n_raw):
    """Frames, frame-0 annotation (objs[b] = the object slots that exist in frame 0) and ragged raw proposals; some
    proposals sit on top of an annotated object so that the matching is meaningful."""
    rng = np.random.default_rng(seed)
    frames = torch.randn(B, T, 3, H, W, generator=torch.Generator().manual_seed(seed)).to(DEV)
    first = np.zeros((B, O, H, W), np.float32)
    rects = {}
    for b in range(B):
        for o in objs[b]:
            y0, x0 = int(rng.integers(0, H - 34)), int(rng.integers(0, W - 40))
            h, w = int(rng.integers(14, 30)), int(rng.integers(16, 36))
            first[b, o, y0:y0 + h, x0:x0 + w] = 1.0
            rects[(b, o)] = (x0, y0, x0 + w - 1, y0 + h - 1)
    props = []
    for b in range(B):
        row = []
        for t in range(n_frames[b]):
            bl = _raw_proposals(rng, n_raw(b, t), H, W)
            for k, o in enumerate(objs[b]):                              # a proposal near every object, drifting with t
                if k < len(bl):
                    x0, y0, x1, y1 = rects[(b, o)]
                    d = 2 * t
                    bl.bbox[k] = torch.tensor([min(x0 + d, W - 8), min(y0 + d, H - 8), min(x1 + d, W - 1), min(y1 + d, H - 1)],
                                              dtype=torch.float32)
            row.append(bl)
        props.append(row)
    return frames, first, props


def test_frame_loop_equals_an_oracle_computed_clip():
    """the DEFAULT product path of the frame loop (fixed slots, one HIP-graph replay per frame:
    dmm_match_solve_packed + dmm_step_finish_f32 + the device frame cursor + the template history carried from frame to
    frame) against a clip computed WITHOUT this package's loop: tests/clip_oracle.py chains oracle.paste_masks -> nms ->
    ROI features -> match_forward -> merge_labels per frame with the evaluator's carry-over (evaluator.py:131-139,205;
    dmm_model.py:66-80).  Ragged proposal counts, O in {2, 0, 4 non-prefix, 3}, one video with 'extra' frames, one
    without templates, top-k cutting (K = 20) and not cutting (K = 48).

    Two comparisons.  STRICT: the chain is handed the same ROI feature rows the device computes (the ROI kernel has its own
    fixtures, G12 / G17) -- then everything else must agree exactly: solver iteration counts incl. the data-dependent
    exits, label maps bit for bit, masks <= 1e-5, in every frame, for the graph path, the unfused epilogue and the BoxList
    path.  INDEPENDENT: the chain on the oracle's own ROIAlign -- the reference's exits fire on exact fp32 equalities, so a
    1e-7 feature difference may move an exit and with it the mean of the iterates; per video the frames up to the first
    differing iteration count are held to the same bounds, and at least half of all live frames must be covered."""
    import clip_oracle
    from dmm_net_amd.roi_features import roialign4_mean_into
    B, T, O, H, W = 4, 4, 5, 96, 128
    objs = [(0, 1), (), (0, 2, 3, 4), (0, 1, 2)]                          # video 2: slot 1 is empty in frame 0 (non-prefix)
    n_frames = [2, 4, 4, 4]                                              # video 0: frames 2, 3 are 'extra'
    enc = _PoolEncoder()
    worst, covered, live_total = 0.0, 0, 0
    for seed, (max_iter, proj_iter) in [(31, (40, 5)), (32, (10, 5)), (33, (40, 5))]:
        cfgs = {"matching": {"algo": "relax"}, "relax_max_iter": max_iter, "relax_proj_iter": proj_iter,
                "relax_learning_rate": 0.1, "score_weight": 0.3}
        frames, first, props = _clip_case(seed, B, T, O, H, W, objs, n_frames, lambda b, t: 18 + 7 * b + 3 * t)
        dfeat = [[f.contiguous() for f in enc(frames[:, t])["backbone_feature"]] for t in range(T)]
        feats = [[f.cpu().numpy() for f in lv] for lv in dfeat]
        raw = [[(p.get_field("mask").numpy()[:, 0], p.bbox.numpy(), p.get_field("scores").numpy()) for p in row]
               for row in props]
        K = 20 if seed == 31 else 48                                    # top-k cuts every frame / no frame

        def device_roi(t, rois):
            rr = torch.from_numpy(np.ascontiguousarray(rois, np.float32)).to(DEV)
            out = torch.empty((rr.shape[0], 4 * dfeat[t][0].shape[1]), dtype=torch.float32, device=DEV)
            return roialign4_mean_into(rr, dfeat[t], out).cpu().numpy()
        kw = dict(max_iter=max_iter, proj_iter=proj_iter, max_proposals=K)
        strict = clip_oracle.run_clip(feats, first, raw, n_frames, roi_fn=device_roi, **kw)
        indep = clip_oracle.run_clip(feats, first, raw, n_frames, **kw)
        kept = strict[3]
        assert np.array_equal(kept, indep[3])                            # paste + NMS do not depend on the features
        assert kept[1:, 2:].min() >= 5 and (K == 20 or len(set(kept[1:, 2:].ravel().tolist())) > 1)   # ragged counts
        live = strict[2] >= 0
        assert live[1:, 2:].all() and not live[:, 1].any() and not live[2:, 0].any() and live[1, 0]
        for (slots, graph, kn) in [(True, True, {}), (True, False, dict(fuse_epilogue=False)), (False, False, {})]:
            lp = video.FrameLoop(enc, DMM_Model(cfgs, is_test=1, feature_extractor=FeatureExtractor()), nms_thresh=0.4,
                                 max_proposals=K)
            lp.slots, lp.graph, lp.record_iters = slots, graph, True
            for k, v in kn.items():
                setattr(lp, k, v)
            for rep in range(2):                                         # second run = replay of the captured step
                labs = {}
                h = lp.run(frames, torch.from_numpy(first).to(DEV).view(B, O, H * W), props, n_frames,
                           on_labels=lambda b, t, lab: labs.__setitem__((b, t), lab.clone()))
                got = torch.stack([x.view(B, O, H, W) for x in h], 0).cpu().numpy()
                assert sorted(labs) == sorted((b, t) for b in range(B) for t in range(n_frames[b]))
                # ---- STRICT: every frame
                exp_h, exp_l, exp_it, _ = strict
                if slots:
                    it = lp.last_iters.cpu().numpy()
                    assert np.array_equal(it[live], exp_it[live]), (seed, slots, graph, it, exp_it)
                err = float(np.abs(got - exp_h).max())
                worst = max(worst, err)
                assert err <= 1e-5, (seed, slots, graph, kn, rep, err)
                for (b, t), lab in labs.items():
                    assert np.array_equal(lab.cpu().numpy(), exp_l[t, b]), (seed, slots, graph, kn, rep, b, t)
                # ---- INDEPENDENT: per video, the frames before the first differing iteration count
                ind_h, ind_l, ind_it, _ = indep
                for b in range(B):
                    upto = T
                    for t in range(T):
                        if live[t, b] and ind_it[t, b] != exp_it[t, b]:
                            upto = t
                            break
                    if rep == 0 and slots and graph:
                        covered += int(live[:upto, b].sum())
                        live_total += int(live[:, b].sum())
                    assert float(np.abs(got[:upto, b] - ind_h[:upto, b]).max(initial=0.0)) <= 1e-5, (seed, b, upto)
                    for t in range(min(upto, n_frames[b])):
                        assert np.array_equal(labs[(b, t)].cpu().numpy(), ind_l[t, b]), (seed, b, t)
        assert strict[1][1:, 2:].max() >= 2                              # several objects really show up in the label maps
    assert covered * 2 >= live_total, (covered, live_total)
    from conftest import record_achieved
    record_achieved("frame_loop_vs_oracle_clip/max_abs_mask_err", worst)
    record_achieved("frame_loop_vs_oracle_clip/independent_roi_frames_covered", covered / max(live_total, 1))


class _RecordingEncoder:
    """Wraps an encoder and keeps a copy of the backbone features of every call (time-major batches of g x B images, in the
    order the loop issues its chunks) -- the clip chained on the CPU is handed EXACTLY the feature maps the loop matched on
    (a bf16 convolution stack is not batch invariant: the same frame encoded in another batch differs in the last bits)."""

    def __init__(self, inner):
        self.inner, self.calls = inner, []
        self.static_outputs = getattr(inner, "static_outputs", False)

    def __call__(self, xs):
        out = self.inner(xs)
        self.calls.append([f.clone() for f in out["backbone_feature"]])
        return out

    def per_frame(self, B):
        """-> feats[t] = the 4 levels of frame t for all B videos (chunks are issued in clip order)."""
        frames = []
        for lv in self.calls:
            g = lv[0].shape[0] // B
            frames += [[f[j * B:(j + 1) * B] for f in lv] for j in range(g)]   # (a batch slice keeps the memory format)
        return frames


def test_frame_loop_equals_an_oracle_computed_clip_at_product_size():
    """the oracle-computed clip at the sizes the product (and bench.py --config loop) runs -- 4 videos
    of 255 x 448, ``FastEncoder``(ResNet-50, bf16 NHWC, fixed seed) features, 50 raw 28 x 28 proposals per frame, top-50,
    F = 5 template slots, eval solver setting 40 x 5 -- against tests/clip_oracle.py (paste -> NMS -> ROI features ->
    match_forward -> label merge with the evaluator's carry-over, evaluator.py:83-149,151-213).  STRICT mode: the chain
    gets the feature maps the loop's encoder produced and the device's ROI rows on them (the exits of relax_matching fire on
    exact fp32 equalities); then labels must agree bit for bit, masks <= 1e-5, iteration counts exactly -- for the captured
    graph path on its first run AND on a replay."""
    import clip_oracle
    from dmm_net_amd.encoder import FastEncoder, FeatureEncoder, GraphedEncoder, fold_batchnorm
    from dmm_net_amd.roi_features import roialign4_mean_into
    B, T, O, H, W, R = 4, 4, 5, 255, 448, 50
    objs = [(0, 1, 2), (0,), (0, 1, 2, 3, 4), (0, 2)]                    # video 3: slot 1 empty in frame 0 (non-prefix)
    n_frames = [4, 4, 4, 3]                                              # video 3: frame 3 is 'extra'
    torch.manual_seed(0)
    enc = _RecordingEncoder(GraphedEncoder(FastEncoder(fold_batchnorm(FeatureEncoder("resnet50").to(DEV).eval())),
                                           miopen_find=True))
    cfgs = {"matching": {"algo": "relax"}, "relax_max_iter": 40, "relax_proj_iter": 5, "relax_learning_rate": 0.1,
            "score_weight": 0.3}
    frames, first, props = _clip_case(51, B, T, O, H, W, objs, n_frames, lambda b, t: R)
    raw = [[(p.get_field("mask").numpy()[:, 0], p.bbox.numpy(), p.get_field("scores").numpy()) for p in row] for row in props]
    lp = video.FrameLoop(enc, DMM_Model(cfgs, is_test=1, feature_extractor=FeatureExtractor()), nms_thresh=0.4,
                         max_proposals=50)
    lp.slots, lp.graph, lp.record_iters = True, True, True
    worst = 0.0
    for rep in range(2):                                                 # second run = replay of the captured step
        enc.calls.clear()
        labs = {}
        h = lp.run(frames, torch.from_numpy(first).to(DEV).view(B, O, H * W), props, n_frames,
                   on_labels=lambda b, t, lab: labs.__setitem__((b, t), lab.clone()))
        torch.cuda.synchronize()
        got = torch.stack([x.view(B, O, H, W) for x in h], 0).cpu().numpy()
        it = lp.last_iters.cpu().numpy()
        dfeat = enc.per_frame(B)
        assert len(dfeat) == T

        def device_roi(t, rois):
            rr = torch.from_numpy(np.ascontiguousarray(rois, np.float32)).to(DEV)
            out = torch.empty((rr.shape[0], 4 * 128), dtype=torch.float32, device=DEV)
            return roialign4_mean_into(rr, dfeat[t], out).cpu().numpy()
        feats = [[None] * 4 for _ in range(T)]                           # the chain only reaches the maps through roi_fn
        exp_h, exp_l, exp_it, kept = clip_oracle.run_clip(feats, first, raw, n_frames, roi_fn=device_roi, max_iter=40,
                                                          proj_iter=5, max_proposals=50)
        live = exp_it >= 0
        assert live[1:, :3].all() and live[1:3, 3].all() and not live[3, 3] and not live[0].any()
        assert kept[1:, :3].min() >= 10                                  # NMS leaves a real table to match
        assert np.array_equal(it[live], exp_it[live]), (rep, it, exp_it)
        err = float(np.abs(got - exp_h).max())
        worst = max(worst, err)
        assert err <= 1e-5, (rep, err)
        assert sorted(labs) == sorted((b, t) for b in range(B) for t in range(n_frames[b]))
        for (b, t), lab in labs.items():
            assert np.array_equal(lab.cpu().numpy(), exp_l[t, b]), (rep, b, t)
        assert exp_l[1:, 2].max() >= 2                                   # several objects really show up in the label maps
    from conftest import record_achieved
    record_achieved("frame_loop_vs_oracle_clip_255x448/max_abs_mask_err", worst)
    record_achieved("frame_loop_vs_oracle_clip_255x448/iters_min", float(exp_it[live].min()))
    record_achieved("frame_loop_vs_oracle_clip_255x448/iters_max", float(exp_it[live].max()))


def test_frame_loop_plan_follows_thresholds_and_solver_settings():
    """nms_thresh / mask_thresh / padding and the match layer's solver settings are immediates of the captured
    frame step.  Changing them between two runs must rebuild the plan (the BoxList path reads them live); a prefetch is
    only taken for the very tensor it was issued for."""
    rng = np.random.default_rng(41)
    B, T, O, H, W = 2, 3, 3, 64, 96
    cfgs = {"matching": {"algo": "relax"}, "relax_max_iter": 12, "relax_proj_iter": 3, "relax_learning_rate": 0.1,
            "score_weight": 0.3}
    frames = torch.randn(B, T, 3, H, W, device=DEV)
    props = [[_raw_proposals(rng, 24, H, W) for t in range(T)] for b in range(B)]
    first = torch.zeros(B, O, H, W, device=DEV)
    first[0, 0, 5:30, 8:40] = 1.0
    first[0, 1, 30:60, 50:90] = 1.0
    first[1, 0, 10:50, 20:60] = 1.0
    first = first.view(B, O, H * W)

    def make(slots):
        lp = video.FrameLoop(_PoolEncoder(), DMM_Model(cfgs, is_test=1, feature_extractor=FeatureExtractor()),
                             nms_thresh=0.4, max_proposals=10)
        lp.slots = lp.graph = slots
        return lp
    fast, ref = make(True), make(False)
    results = []
    for change in (lambda lp: None, lambda lp: setattr(lp, "nms_thresh", 0.05), lambda lp: setattr(lp, "mask_thresh", 0.7),
                   lambda lp: setattr(lp.dmm.match_layer, "max_iter", 3), lambda lp: setattr(lp, "padding", 2)):
        change(fast)
        change(ref)
        a, b_ = fast.run(frames, first, props), ref.run(frames, first, props)
        assert all(torch.equal(x, y) for x, y in zip(a, b_))
        results.append(torch.stack([x.clone() for x in a]))
    assert all(not torch.equal(results[0], r) for r in results[1:3])     # the changes really change the result
    # prefetch identity: issued for one tensor, a DIFFERENT tensor of the same shape (even at the same address) is refused
    lp = make(True)
    nxt = torch.randn(B, T, 3, H, W, device=DEV)
    lp.run(frames, first, props, next_frames=nxt)
    assert lp._prefetched is not None and lp._prefetched[0][0] is nxt
    ptr = nxt.data_ptr()
    del nxt
    other = torch.randn(B, T, 3, H, W, device=DEV)                        # may or may not reuse the address
    lp._prefetched = (lp._prefetched[0], lp._prefetched[1])
    held = lp._prefetched[0][0]
    assert held.data_ptr() == ptr and other.data_ptr() != ptr            # the record keeps the tensor alive: no reuse
    got = lp.run(other, first, props)
    exp = make(False).run(other, first, props)
    assert all(torch.equal(x, y) for x, y in zip(got, exp)) and lp._prefetched is None
