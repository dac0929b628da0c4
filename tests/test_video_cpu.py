"""Frame-loop side of the path on the CPU: oracle vs the G11 goldens, output / input file formats."""
import sys
import types

import numpy as np
import pytest
import torch

import oracle
from conftest import golden
from dmm_net_amd import synth, video
from dmm_net_amd.proposals import SimpleBoxList


def test_g11_oracle_mask_boxes_and_valid():
    g = golden("g11_frame_loop")
    for k in range(int(g["n_box"])):
        O, H, W = [int(v) for v in g[f"box{k}_shape"]]
        boxes, valid = oracle.mask_boxes(synth.template_planes(k, O, H, W))
        assert np.array_equal(boxes, g[f"box{k}_boxes"]), k
        assert np.array_equal(valid, g[f"box{k}_valid"]), k


def test_g11_oracle_merge_labels():
    g = golden("g11_frame_loop")
    for k in range(int(g["n_mrg"])):
        O, n_obj, H, W = [int(v) for v in g[f"mrg{k}_shape"]]
        outs = synth.refined_planes(k, O, H, W).reshape(1, O, H * W)
        lab = oracle.merge_labels(outs, [n_obj])[0].reshape(H, W)
        assert np.array_equal(lab, g[f"mrg{k}_labels"]), k
        assert lab.max() <= n_obj


def test_davis_palette_and_png_roundtrip(tmp_path):
    pal = video.davis_palette()
    assert len(pal) == 768 and pal[:12] == [0, 0, 0, 128, 0, 0, 0, 128, 0, 128, 128, 0]
    assert pal[3 * 8:3 * 8 + 3] == [64, 0, 0] and pal[3 * 255:] == [224, 224, 192]
    Image = pytest.importorskip("PIL.Image")
    lab = (np.arange(40 * 30).reshape(40, 30) % 7).astype(np.uint8)
    f = tmp_path / "merged" / "vid" / "00005.png"
    video.save_label_png(torch.from_numpy(lab)[None], str(f))           # [1,H,W] accepted like plot_scores_map
    im = Image.open(str(f))
    assert im.mode == "P" and im.size == (30, 40)
    assert np.array_equal(np.array(im), lab) and im.getpalette()[:768] == pal


def test_load_offline_proposals_maps_boxlist_without_maskrcnn_benchmark(tmp_path):
    """Pickle objects whose class path is maskrcnn_benchmark's BoxList (through a throw-away module), drop the
    module, and read the file back through the package's unpickler."""
    names = ["maskrcnn_benchmark", "maskrcnn_benchmark.structures", "maskrcnn_benchmark.structures.bounding_box"]
    assert all(n not in sys.modules for n in names)

    class BoxList:                                                      # attribute layout of the real class
        def __init__(self, bbox, size, mode="xyxy"):
            self.bbox, self.size, self.mode, self.extra_fields = bbox, size, mode, {}

    BoxList.__module__, BoxList.__qualname__ = names[2], "BoxList"
    mods = [types.ModuleType(n) for n in names]
    mods[2].BoxList = BoxList
    try:
        for n, m in zip(names, mods):
            sys.modules[n] = m
        preds = {}
        for fid, n in (("00000", 3), ("00005", 0)):
            b = BoxList(torch.arange(4.0 * n).view(n, 4), (448, 255))
            b.extra_fields["mask"] = torch.rand(n, 1, 28, 28)
            b.extra_fields["scores"] = torch.rand(n)
            preds[fid] = b
        torch.save({"vidA": preds}, str(tmp_path / "pred_DICT.pth"))
        torch.save([preds["00000"], preds["00005"]], str(tmp_path / "predictions.pth"))
    finally:
        for n in names:
            sys.modules.pop(n, None)
    d = video.load_offline_proposals(str(tmp_path / "pred_DICT.pth"))
    p = d["vidA"]["00000"]
    assert isinstance(p, SimpleBoxList) and len(p) == 3 and p.size == (448, 255) and p.mode == "xyxy"
    assert sorted(p.fields()) == ["mask", "scores"] and p.get_field("mask").shape == (3, 1, 28, 28)
    assert torch.equal(p.bbox, preds["00000"].bbox) and len(d["vidA"]["00005"]) == 0
    lst = video.load_offline_proposals(str(tmp_path / "predictions.pth"))
    assert [len(x) for x in lst] == [3, 0] and isinstance(lst[1], SimpleBoxList)
    r = p.resize((224, 255))                                            # BoxList.resize: per-axis ratios
    assert r.size == (224, 255) and torch.equal(r.bbox[:, 0], p.bbox[:, 0] * 0.5) and torch.equal(r.bbox[:, 1], p.bbox[:, 1])
    top = p[p.get_field("scores").sort(0, descending=True)[1][:2]]      # reduce_pth_size_by_videos.py:80-84
    assert len(top) == 2 and top.get_field("mask").shape[0] == 2


def test_video_ops_refuse_cpu_tensors():
    from dmm_net_amd._lib import DmmError
    with pytest.raises(DmmError):
        video.mask_boxes(torch.zeros(2, 4, 4))
    with pytest.raises(DmmError):
        video.merge_labels(torch.zeros(1, 2, 16))


def test_clip_proposals_packs_ragged_boxlists_on_the_host():
    """ClipProposals.from_boxlists (host logic of the fixed-slot frame step): frame-major [T,B,R,..] layout, zero padding
    past a list's count, the last entry of a short video reused (evaluator.py:101-106), boxes rescaled like
    BoxList.resize when a list was made for another image size, 'objectness' accepted for 'scores'."""
    from dmm_net_amd.proposals import ClipProposals
    rng = np.random.default_rng(4)
    H, W, T = 40, 60, 3

    def bl(n, size=(W, H), field="scores"):
        b = SimpleBoxList(torch.from_numpy(rng.uniform(0, 30, (n, 4)).astype(np.float32)), size)
        b.add_field(field, torch.from_numpy(rng.random(n).astype(np.float32)))
        b.add_field("mask", torch.from_numpy(rng.random((n, 1, 28, 28)).astype(np.float32)))
        return b
    props = [[bl(5), bl(2), bl(7)], [bl(3, size=(2 * W, 2 * H))]]            # video 1: one list, made at twice the size
    clip = ClipProposals.from_boxlists(props, T, H, W, "cpu")
    assert (clip.T, clip.B, clip.R, clip.M) == (T, 2, 7, 28)
    assert clip.counts.tolist() == [[5, 3], [2, 3], [7, 3]]
    for t in range(T):
        p = props[0][t]
        n = len(p)
        assert torch.equal(clip.boxes[t, 0, :n], p.bbox) and float(clip.boxes[t, 0, n:].abs().sum()) == 0.0
        assert torch.equal(clip.prob[t, 0, :n], p.get_field("mask")[:, 0]) and torch.equal(clip.scores[t, 0, :n], p.get_field("scores"))
        q = props[1][0]
        assert torch.equal(clip.boxes[t, 1, :3], q.bbox * 0.5)                 # resized to (W, H); reused for every frame
        assert torch.equal(clip.prob[t, 1, :3], q.get_field("mask")[:, 0])
    big = ClipProposals.empty(8, 2, 10, 28, "cpu")
    out = ClipProposals.from_boxlists(props, T, H, W, "cpu", out=big)
    assert out is big and torch.equal(big.prob[:T, :, :7], clip.prob) and big.counts[:T].tolist() == clip.counts.tolist()
    obj = [[bl(4, field="objectness")] for _ in range(2)]
    assert ClipProposals.from_boxlists(obj, 1, H, W, "cpu").scores.shape == (1, 2, 4)


def test_frame_loop_chunk_sizes_by_clip_length_and_mode():
    """Host logic of the encoder chunking: a third of the clip within [4, 9] frames when every clip pays its own pipeline
    fill, up to 12 frames when clips follow each other back to back (``run(next_frames=...)``); ``encode_ahead`` /
    ``encode_first`` override; chunks never exceed the clip."""
    from dmm_net_amd import video
    lp = video.FrameLoop(encoder=lambda x: None, dmm=None)
    assert [lp._frames_per_chunk(T) for T in (1, 4, 12, 18, 24, 27, 36, 100)] == [4, 4, 4, 6, 8, 9, 9, 9]
    assert [lp._frames_per_chunk(T, True) for T in (1, 4, 12, 18, 36)] == [4, 4, 12, 12, 12]
    assert [lp._first_chunk(T) for T in (1, 3, 12, 36)] == [1, 3, 4, 9]
    assert [lp._first_chunk(T, True) for T in (1, 3, 12, 36)] == [1, 3, 12, 12]
    lp.encode_ahead = 5
    assert lp._frames_per_chunk(36) == lp._frames_per_chunk(36, True) == 5 and lp._first_chunk(3, True) == 3
    lp.encode_first = 2
    assert lp._first_chunk(36) == lp._first_chunk(36, True) == 2
    assert lp._prefetched is None


def test_oracle_clip_chain_carries_the_history_and_skips_like_the_reference():
    """tests/clip_oracle.py (the CPU clip the GPU frame loop is compared with): frame 0 reports the annotation, a video
    without templates or past its length yields zeros and keeps its history, the template history of frame t feeds
    frame t + 1, and a one-frame-per-call replay of the chain from the carried history gives the same masks."""
    import clip_oracle
    rng = np.random.default_rng(3)
    B, T, O, H, W, C = 3, 3, 3, 48, 64, 4
    feats = [[rng.standard_normal((B, C, -(-H // s), -(-W // s))).astype(np.float32) for s in (4, 8, 16, 32)]
             for _ in range(T)]
    first = np.zeros((B, O, H, W), np.float32)
    first[0, 0, 4:20, 6:30] = 1
    first[0, 2, 22:44, 30:60] = 1                                         # slot 1 empty: non-prefix
    first[2, 0, 10:30, 10:40] = 1

    def raw(n):
        x1, y1 = rng.uniform(0, W - 24, n), rng.uniform(0, H - 24, n)
        bx = np.stack([x1, y1, np.minimum(x1 + rng.uniform(10, 40, n), W - 1), np.minimum(y1 + rng.uniform(10, 30, n), H - 1)], 1)
        return (rng.random((n, 28, 28)) * 0.6 + 0.4).astype(np.float32), bx.astype(np.float32), rng.random(n).astype(np.float32)
    props = [[raw(12 + b + t) for t in range(T)] for b in range(B)]
    hist, labels, iters, kept = clip_oracle.run_clip(feats, first, props, [3, 3, 2], max_iter=10, proj_iter=3,
                                                     max_proposals=8)
    # video 0: two live templates in slots 0 and 2 -> the reference looks at the first valid.sum() = 2 ROWS (slot 1 is
    # empty), so only label 1 shows up (evaluator.py:134-139 with a non-prefix layout)
    assert np.array_equal(hist[0], first) and labels[0, 0].max() == 1 and labels[0, 2].max() == 1
    assert not hist[1:, 1].any() and (iters[:, 1] == -1).all()            # no template at all
    assert not hist[2, 2].any() and iters[2, 2] == -1 and iters[1, 2] >= 0   # 'extra' frame
    assert (kept[1:, 0] > 0).all() and (kept <= 8).all()
    assert not hist[1:, 0, 2].any()        # non-prefix: n_live = 2 -> rows 0, 1 are matched, row 1 is scaled by valid = 0
    assert hist[1, 0, 0].any()
    # frame 2 of video 0 is the layer applied to frame 1's output (the carried history), nothing else
    again, _, _, _ = clip_oracle.run_clip(feats[:1] + feats[2:], hist[1], [[p[0], p[2]] for p in props], [2, 2, 1],
                                          max_iter=10, proj_iter=3, max_proposals=8)
    # (templates are re-derived from frame 1's masks there, so only shapes / skipping are comparable, not values)
    assert again.shape == (2, B, O, H, W)
