"""Proposal preprocessing kernels against the goldens / the oracle."""
import numpy as np
import pytest
import torch

import oracle
from conftest import golden
from dmm_net_amd import proposals

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_g9_paste_masks_matches_reference_steps():
    g = golden("g9_paste")
    for k in range(int(g["n"])):
        c = g.group(f"c{k}")
        h, w = [int(v) for v in c["size"]]
        planes, nb = proposals.paste_masks(torch.from_numpy(c["prob"]).to(DEV), torch.from_numpy(c["boxes"]).to(DEV), h, w,
                                           float(c["thresh"]), int(c["padding"]))
        got = planes[:, 0].cpu().numpy()
        assert float(np.abs(got - c["masks"]).max()) <= 2.4e-7          # torch's scalar-tail path differs by 1 ulp
        assert np.array_equal(nb.cpu().numpy(), c["new_boxes"])
        om, ob = oracle.paste_masks(c["prob"], c["boxes"], h, w, float(c["thresh"]), int(c["padding"]))
        assert np.array_equal(got, om) and np.array_equal(nb.cpu().numpy(), ob)   # bit exact vs the oracle


@pytest.mark.parametrize("H,W", [(256, 448), (480, 854), (1080, 1920)])
def test_paste_boxes_and_label_merge_at_the_products_plane_sizes(H, W):
    """Paste (28 x 28 -> frame), tight boxes, the 1-bit planes, the template boxes of ``ohw_mask2boxlist`` and the label merge
    at the evaluator's default size, DAVIS 480p and a 1080p frame, bit exact against the oracle (every fixture and list
    topped out at 255 x 448 before)."""
    from dmm_net_amd import ops, video
    rng = np.random.default_rng(H)
    P, O = 30, 5
    prob = rng.random((P, 28, 28), dtype=np.float32)
    x1, y1 = rng.uniform(-4, W - 30, P), rng.uniform(-4, H - 30, P)
    boxes = np.stack([x1, y1, np.minimum(x1 + rng.uniform(6, W * 0.6, P), W + 3), np.minimum(y1 + rng.uniform(6, H * 0.6, P), H + 3)],
                     1).astype(np.float32)
    planes, nb, packed = proposals.paste_masks(torch.from_numpy(prob).to(DEV), torch.from_numpy(boxes).to(DEV), H, W, 0.4, 1,
                                               want_packed=True)
    om, ob = oracle.paste_masks(prob, boxes, H, W, 0.4, 1)
    assert np.array_equal(planes[:, 0].cpu().numpy(), om) and np.array_equal(nb.cpu().numpy(), ob)
    assert torch.equal(packed, ops.pack_masks(planes.transpose(0, 1))[0])
    keep = oracle.nms(ob, rng.random(P).astype(np.float32), 0.4, 50)
    assert len(keep) >= 1
    outs = planes[:O, 0] * torch.from_numpy(rng.random((O, 1, 1), dtype=np.float32)).to(DEV)
    outs[1] = 0                                                        # an empty template plane: whole frame, invalid
    gb, gv = video.mask_boxes(outs)
    eb, ev = oracle.mask_boxes(outs.cpu().numpy())
    assert np.array_equal(gb.cpu().numpy(), eb) and np.array_equal(gv.cpu().numpy(), ev)
    lab = video.merge_labels(outs[None], torch.tensor([O], device=DEV))
    assert np.array_equal(lab.cpu().numpy().reshape(1, -1), oracle.merge_labels(outs.cpu().numpy().reshape(1, O, -1), [O]))


def test_nms_and_filter_results_match_oracle():
    rng = np.random.default_rng(4)
    lists, exp = [], []
    for n in (0, 1, 37, 90, 300):
        x1, y1 = rng.uniform(0, 200, n), rng.uniform(0, 200, n)
        b = np.stack([x1, y1, x1 + rng.uniform(1, 90, n), y1 + rng.uniform(1, 90, n)], 1).astype(np.float32)
        s = rng.random(n).astype(np.float32)
        if n > 10:
            s[5] = s[3]                                             # score tie: lower index first
            b[7] = b[2]                                             # duplicate box
        bl = proposals.SimpleBoxList(torch.from_numpy(b).to(DEV), (255, 255))
        bl.add_field("scores", torch.from_numpy(s).to(DEV))
        bl.add_field("mask", torch.arange(n, device=DEV).float())
        lists.append(bl)
        exp.append(oracle.nms(b, s, 0.4, 50))
    out = proposals.filter_results(lists, nms_thresh=0.4, max_proposals=50)
    for bl, e in zip(out, exp):
        assert len(bl) == len(e) <= 50
        assert np.array_equal(bl.get_field("mask").cpu().numpy().astype(np.int64), e.astype(np.int64))


def test_g13_nms_topk_matches_reference_filter_results():
    """G13: kept indices of the HIP NMS + top-k against the reference's own filter_results (imported,
    boxlist_ops.py:15-29) over the published greedy NMS -- duplicates, score ties, exact-threshold pairs."""
    g = golden("g13_nms")
    for k in range(int(g["n"])):
        c = g.group(f"c{k}")
        n = len(c["scores"])
        bl = proposals.SimpleBoxList(torch.from_numpy(c["boxes"]).to(DEV), (256, 256))
        bl.add_field("scores", torch.from_numpy(c["scores"]).to(DEV))
        bl.add_field("mask", torch.arange(n, device=DEV).float())
        out = proposals.filter_results([bl], nms_thresh=float(c["thresh"]), max_proposals=int(c["max_keep"]))[0]
        assert np.array_equal(out.get_field("mask").cpu().numpy().astype(np.int32), c["keep"]), k
        assert np.array_equal(out.bbox.cpu().numpy(), c["kept_boxes"])
        assert np.array_equal(out.get_field("scores").cpu().numpy(), c["kept_scores"])


def test_packed_planes_give_identical_tables():
    """DMM_PACKED1: pack once, count from 1/32 of the bytes -- same integer tables as the fp32 path."""
    from dmm_net_amd import ops
    rng = np.random.default_rng(8)
    for (B, N, M, H, W) in [(2, 50, 10, 255, 255), (3, 7, 3, 5, 9), (1, 130, 20, 33, 40), (2, 20, 4, 16, 16)]:
        pm = torch.from_numpy(rng.random((B, N, H, W), dtype=np.float32)).to(DEV)
        tm = torch.from_numpy(rng.random((B, M, H, W), dtype=np.float32)).to(DEV)
        pp, pt = ops.pack_masks(pm), ops.pack_masks(tm)
        assert pp.shape == (B, N, ops.pack_words(H * W))
        nv = torch.tensor([N, max(N - 3, 1), 1][:B], dtype=torch.int32, device=DEV)
        a = ops.iou_counts(pm, tm, nv, None)
        b = ops.iou_counts_packed(pp, pt, H * W, nv, None)
        for x, y in zip(a, b):
            assert torch.equal(x, y), (B, N, M, H, W)
        # pad bits are zero: total popcount of a packed plane == its area
        bits = pp.view(torch.uint8)
        pop = torch.tensor([bin(int(v)).count("1") for v in range(256)], device=DEV)[bits.long()].flatten(2).sum(2)
        assert torch.equal(pop.int(), ops.iou_counts(pm, tm)[1])
    # fp16 source
    pm16 = pm.half()
    assert torch.equal(ops.pack_masks(pm16), ops.pack_masks(pm16.float()))


def test_paste_emits_the_packed_form():
    from dmm_net_amd import ops
    g = golden("g9_paste")
    c = g.group("c1")
    h, w = [int(v) for v in c["size"]]
    planes, nb, packed = proposals.paste_masks(torch.from_numpy(c["prob"]).to(DEV), torch.from_numpy(c["boxes"]).to(DEV),
                                               h, w, float(c["thresh"]), int(c["padding"]), want_packed=True)
    assert torch.equal(packed, ops.pack_masks(planes.transpose(0, 1))[0])
    assert np.array_equal(nb.cpu().numpy(), c["new_boxes"])


def test_dmm_model_inference_counts_on_the_packed_planes_of_the_paste_kernel():
    """SURVEY 8f-3 / VERDICT r1 item 5: forward_mask_prop(want_packed=True) carries the paste kernel's 1-bit planes
    through NMS + top-k ('mask_packed' rows are selected with the rest), and DMM_Model.inference runs its cost pass on
    them: identical outputs to the float-plane cost pass."""
    from dmm_net_amd.dmm_model import DMM_Model
    from dmm_net_amd.roi_features import FeatureExtractor
    g = torch.Generator(device=DEV).manual_seed(12)
    B, F, H, W, C = 3, 5, 96, 128, 16
    raw = []
    for b in range(B):
        n = 30 + 7 * b
        x1 = torch.rand(n, generator=g, device=DEV) * (W - 40)
        y1 = torch.rand(n, generator=g, device=DEV) * (H - 40)
        box = torch.stack([x1, y1, x1 + 8 + torch.rand(n, generator=g, device=DEV) * 60,
                           y1 + 8 + torch.rand(n, generator=g, device=DEV) * 50], 1)
        bl = proposals.SimpleBoxList(box, (W, H))
        bl.add_field("mask", torch.rand((n, 1, 28, 28), generator=g, device=DEV))
        bl.add_field("scores", torch.rand(n, generator=g, device=DEV))
        raw.append(bl)
    props = proposals.forward_mask_prop([p.get_field("mask") for p in raw], raw, 0.4, 1, want_packed=True)
    props = proposals.filter_results(list(props), 0.4, 20)
    assert all("mask_packed" in p.fields() and p.get_field("mask_packed").shape[0] == len(p) for p in props)
    feats = tuple(torch.randn((B, C, -(-H // s), -(-W // s)), generator=g, device=DEV) for s in (4, 8, 16, 32))
    cfgs = {"matching": {"algo": "relax"}, "relax_max_iter": 20, "relax_proj_iter": 5, "relax_learning_rate": 0.1,
            "score_weight": 0.3}
    model = DMM_Model(cfgs, is_test=1, feature_extractor=FeatureExtractor())
    tplt = {b: {"feat": [torch.randn((F, 4 * C), generator=g, device=DEV)]} for b in range(B)}
    valid = torch.tensor([[1, 1, 1, 0, 0], [1, 1, 1, 1, 1], [1, 0, 0, 0, 0]], dtype=torch.float32, device=DEV)
    ml = torch.rand((B, F, H, W), generator=g, device=DEV)
    infos = {"args": None, "shape": None, "extra_frame": [0] * B, "valid": valid}
    with torch.no_grad():
        out_p, _, _, last_p = model.inference(infos, props, feats, ml, tplt)
        plain = []
        for p in props:
            q = proposals.SimpleBoxList(p.bbox, p.size)
            for f in p.fields():
                if f != "mask_packed":
                    q.add_field(f, p.get_field(f))
            plain.append(q)
        out_f, _, _, last_f = model.inference(infos, plain, feats, ml, tplt)
    assert torch.equal(out_p, out_f) and torch.equal(last_p, last_f)
    assert float(out_p[1].abs().sum()) > 0
