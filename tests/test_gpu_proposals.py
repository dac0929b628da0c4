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
