"""Feature path (reference a9 / a10) on the GPU: fused ROIAlign+mean vs the oracle's literal restatement, its
backward by the adjoint identity, the encoder's structure, and BASELINE config 3 end to end (bf16 encoder ->
ROI features -> matching layer)."""
import numpy as np
import pytest
import torch

import oracle
from dmm_net_amd import synth
from dmm_net_amd.encoder import FeatureEncoder
from dmm_net_amd.match_model import MatchModel
from dmm_net_amd.roi_features import FeatureExtractor, convert_to_roi_format

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


class Boxes:
    def __init__(self, bbox):
        self.bbox = bbox

    def __len__(self):
        return self.bbox.shape[0]


def random_boxes(rng, n, H, W):
    x1 = rng.uniform(-5, W - 2, n)
    y1 = rng.uniform(-5, H - 2, n)
    w = rng.uniform(0.2, W * 0.8, n)
    h = rng.uniform(0.2, H * 0.8, n)
    return np.stack([x1, y1, np.minimum(x1 + w, W + 6), np.minimum(y1 + h, H + 6)], 1).astype(np.float32)


@pytest.mark.parametrize("B,C,H,W,n", [(2, 8, 64, 64, 7), (1, 128, 255, 255, 50), (3, 5, 37, 91, 4)])
def test_roialign4_mean_matches_oracle(B, C, H, W, n):
    rng = np.random.default_rng(B * 100 + C)
    sizes = [((H + s - 1) // s, (W + s - 1) // s) for s in (4, 8, 16, 32)]
    feats = [rng.standard_normal((B, C, h, w)).astype(np.float32) for (h, w) in sizes]
    boxes = [random_boxes(rng, n + b, H, W) for b in range(B)]
    boxes[0][0] = [3.0, 4.0, 3.2, 4.1]                          # smaller than one feature pixel: size clamp >= 1
    boxes[-1][-1] = [-40.0, -40.0, -20.0, -20.0]                # fully outside: zeros
    rois = np.concatenate([np.concatenate([np.full((len(bb), 1), b, np.float32), bb], 1) for b, bb in enumerate(boxes)])
    exp = oracle.roialign4_mean(feats, rois)
    fe = FeatureExtractor()
    out = fe(tuple(torch.from_numpy(f).to(DEV) for f in feats), [Boxes(torch.from_numpy(bb).to(DEV)) for bb in boxes])
    assert out.shape == (rois.shape[0], 4 * C)
    err = float(np.abs(out.cpu().numpy() - exp).max())
    assert err < 2e-5 * max(1.0, float(np.abs(exp).max())), err
    assert torch.equal(convert_to_roi_format([Boxes(torch.from_numpy(bb)) for bb in boxes]), torch.from_numpy(rois))


def test_roialign4_mean_forward_and_backward_match_g12_fixture():
    """G12: forward values AND d feat_l of the fused HIP kernels against an independent differentiable torch
    formulation of legacy ROIAlign(14x14, sr 2, 4 levels) + mean (per-bin definition, gen_golden.py:_roialign_legacy);
    boxes include sub-pixel, clipped (both corners), fully-outside and whole-frame ones.  <= 1e-5 relative."""
    from conftest import golden
    g = golden("g12_roialign")
    for k in range(int(g["n"])):
        c = g.group(f"c{k}")
        B = int(c["shape"][0])
        feats = [torch.from_numpy(c[f"feat{l}"]).to(DEV).requires_grad_(True) for l in range(4)]
        rois = c["rois"]
        # the extractor takes per-image box lists in image order: G12's rois are regrouped by batch index
        order = np.argsort(rois[:, 0], kind="stable")
        boxes = [Boxes(torch.from_numpy(rois[rois[:, 0] == b][:, 1:]).to(DEV)) for b in range(B)]
        out = FeatureExtractor()(tuple(feats), boxes)
        exp = c["out"][order]
        err = float(np.abs(out.detach().cpu().numpy() - exp).max())
        from conftest import record_achieved
        record_achieved(f"g12_roialign/c{k}/fwd_rel_err", err / max(1.0, float(np.abs(exp).max())))
        assert err <= 1e-5 * max(1.0, float(np.abs(exp).max())), (k, err)
        out.backward(torch.from_numpy(c["wgt"][order]).to(DEV))
        for l in range(4):
            ge = c[f"grad{l}"]
            gerr = float(np.abs(feats[l].grad.cpu().numpy() - ge).max())
            record_achieved(f"g12_roialign/c{k}/grad{l}_rel_err", gerr / max(1.0, float(np.abs(ge).max())))
            assert gerr <= 1e-5 * max(1.0, float(np.abs(ge).max())), (k, l, gerr)


@pytest.mark.parametrize("B,C,H,W,n,dtype", [(2, 8, 64, 64, 7, torch.float32), (1, 128, 255, 255, 50, torch.bfloat16),
                                              (3, 16, 37, 91, 4, torch.float16), (2, 128, 255, 448, 20, torch.float32),
                                              (1, 1024, 40, 40, 3, torch.bfloat16), (2, 12, 40, 40, 3, torch.float32)])
def test_roialign4_mean_channels_last_kernel(B, C, H, W, n, dtype):
    """The NHWC form (channels-last levels, what FastEncoder produces) against the oracle on the same (rounded) values
    and against the NCHW kernel; dead rois (image index -1) give zero rows; C outside its envelope falls back."""
    from dmm_net_amd.roi_features import _layout, roialign4_mean_into
    rng = np.random.default_rng(B * 10 + C)
    sizes = [((H + s - 1) // s, (W + s - 1) // s) for s in (4, 8, 16, 32)]
    feats = [torch.from_numpy(rng.standard_normal((B, C, h, w)).astype(np.float32)).to(DEV).to(dtype) for (h, w) in sizes]
    boxes = [random_boxes(rng, n + b, H, W) for b in range(B)]
    boxes[0][0] = [0.0, 0.0, W - 1.0, H - 1.0]                  # whole frame: the largest patch
    rois = np.concatenate([np.concatenate([np.full((len(bb), 1), b, np.float32), bb], 1) for b, bb in enumerate(boxes)])
    rois[-1, 0] = -1.0                                          # a dead slot
    exp = oracle.roialign4_mean([f.float().cpu().numpy() for f in feats], rois[:-1])
    cl = [f.contiguous(memory_format=torch.channels_last) for f in feats]
    expect_nhwc = C % (4 if dtype == torch.float32 else 8) == 0 and C != 12
    assert (_layout(cl) == "nhwc") == expect_nhwc
    rt = torch.from_numpy(rois).to(DEV)
    out = roialign4_mean_into(rt, cl if expect_nhwc else feats, torch.empty((len(rois), 4 * C), device=DEV))
    ref = roialign4_mean_into(rt, feats, torch.empty((len(rois), 4 * C), device=DEV))
    scale = max(1.0, float(np.abs(exp).max()))
    assert float(np.abs(out[:-1].cpu().numpy() - exp).max()) < 2e-5 * scale
    assert float((out - ref).abs().max()) < 2e-5 * scale
    assert float(out[-1].abs().sum()) == 0.0 and float(ref[-1].abs().sum()) == 0.0
    fe_out = FeatureExtractor()(tuple(cl), [Boxes(torch.from_numpy(bb).to(DEV)) for bb in boxes])
    assert torch.equal(fe_out[:-1], out[:-1])                   # the module takes the same path


def test_roialign4_mean_backward_is_the_adjoint():
    rng = np.random.default_rng(5)
    B, C, H, W = 2, 16, 96, 80
    feats = [torch.from_numpy(rng.standard_normal((B, C, (H + s - 1) // s, (W + s - 1) // s)).astype(np.float32))
             .to(DEV).requires_grad_(True) for s in (4, 8, 16, 32)]
    boxes = [Boxes(torch.from_numpy(random_boxes(rng, 9, H, W)).to(DEV)) for _ in range(B)]
    out = FeatureExtractor()(tuple(feats), boxes)
    g = torch.randn_like(out)
    out.backward(g)
    lhs = float((out.detach().double() * g.double()).sum())           # <g, A f>
    rhs = float(sum((f.grad.double() * f.detach().double()).sum() for f in feats))   # <A^T g, f>
    assert abs(lhs - rhs) < 1e-4 * max(1.0, abs(lhs)), (lhs, rhs)


def test_config3_encoder_roi_match_end_to_end():
    """BASELINE config 3: ResNet-50 + prop heads in bf16 (MIOpen) -> fused 4-level ROIAlign-mean -> cosine+IoU cost ->
    solver, batch of 8 frames; compared with the same pipeline in fp32 (argmax identical, sim within bf16 noise)."""
    torch.manual_seed(0)
    # train() = BatchNorm on batch statistics: with random-init weights and identity running stats the
    # activations of a 50-layer residual net explode and bf16-vs-fp32 stops being meaningful
    enc = FeatureEncoder("resnet50").to(DEV).train().to(memory_format=torch.channels_last)
    fe = FeatureExtractor()
    B, P, O, H, W = 8, 50, 10, 255, 255
    img = torch.randn(B, 3, H, W, device=DEV).contiguous(memory_format=torch.channels_last)
    rng = np.random.default_rng(3)
    frames = [synth.make_frame(P, O, H, W, 8, seed=3000 + b, kind="structured") for b in range(B)]

    def tight_boxes(masks):
        out = []
        for m in masks:
            ys, xs = np.where(m > 0.5)
            out.append([xs.min(), ys.min(), xs.max() + 1, ys.max() + 1] if len(xs) else [0, 0, 8, 8])
        return np.asarray(out, np.float32)

    pboxes = [Boxes(torch.from_numpy(tight_boxes(fr.proposed_mask)).to(DEV)) for fr in frames]
    tboxes = [Boxes(torch.from_numpy(tight_boxes(fr.mask_last_occurence)).to(DEV)) for fr in frames]
    res, maps = {}, {}
    for tag, dtype in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
        with torch.no_grad(), torch.autocast("cuda", dtype=dtype, enabled=dtype != torch.float32):
            bf = enc(img)["backbone_feature"]
        assert [t.shape[1] for t in bf] == [128] * 4 and [t.shape[2] for t in bf] == [64, 32, 16, 8]
        maps[tag] = bf
        pf = fe(bf, pboxes).view(B, P, 512)
        tf = fe(bf, tboxes).view(B, O, 512)
        model = MatchModel({"matching": {"algo": "relax"}, "relax_max_iter": 20, "relax_proj_iter": 5,
                            "relax_learning_rate": 0.1, "score_weight": 0.3}, is_test=1)
        outs = []
        for b in range(B):
            fr = frames[b]
            fo, ms, ds, _, _ = model(pf[b], torch.from_numpy(fr.proposed_mask).to(DEV), [tf[b]],
                                     torch.from_numpy(fr.mask_last_occurence).to(DEV),
                                     torch.from_numpy(fr.proposal_score).to(DEV))
            outs.append((fo, ms))
        res[tag] = (pf, tf, outs)
    assert maps["bf16"][0].dtype == torch.bfloat16 and torch.isfinite(res["bf16"][0]).all()
    # (1) bf16 maps are finite.  (No bf16-vs-fp32 closeness bound: a RANDOM-INIT 50-layer residual net amplifies the
    # 2^-8 rounding of every layer chaotically -- measured 0.11 relative at stride 4, 0.44 at stride 32 -- and that
    # arithmetic is torch/MIOpen's, not this package's.)
    for lb in maps["bf16"]:
        assert torch.isfinite(lb).all()
    # (2) the ROI kernel's bf16 input path == its fp32 path on the same (upcast) maps
    pf_up = fe(tuple(t.float() for t in maps["bf16"]), pboxes).view(B, P, 512)
    assert float((pf_up - res["bf16"][0]).abs().max()) < 1e-5
    # (3) the layer runs on those features: finite outputs, one proposal plane per template (test-mode gather)
    for tag in ("fp32", "bf16"):
        for b in range(B):
            fo, ms = res[tag][2][b]
            assert fo.shape == (O, H, W) and torch.isfinite(fo).all() and torch.isfinite(ms).all()
            pm = torch.from_numpy(frames[b].proposed_mask).to(DEV).flatten(1)
            for o in range(O):
                # fo[o] = w * pm[p] for exactly one p: the best-fitting proposal reproduces it exactly
                proj = (pm @ fo[o].flatten()) / (pm * pm).sum(1)
                p = int(torch.argmax(proj * (pm * pm).sum(1).sqrt()))
                assert float((fo[o].flatten() - proj[p] * pm[p]).abs().max()) < 1e-5, (tag, b, o)


def test_training_step_end_to_end():
    """BASELINE config 4 in miniature on one GPU: encoder -> ROI features -> DMM_Model (ragged batch) -> loss ->
    backward through the HIP layer, the ROI kernel and the MIOpen encoder -> Adam step.  The matching loss on the
    template/proposal features must go down (the reference trains exactly these parameters, base.py:62-69)."""
    from dmm_net_amd.dmm_model import DMM_Model
    from dmm_net_amd.proposals import SimpleBoxList
    torch.manual_seed(0)
    B, F, P, H, W = 2, 5, 12, 96, 128
    enc = FeatureEncoder("resnet34", hidden_size=32).to(DEV).train()
    fe = FeatureExtractor()
    cfgs = {"matching": {"algo": "relax"}, "relax_max_iter": 10, "relax_proj_iter": 5, "relax_learning_rate": 0.1,
            "score_weight": 0.3}
    model = DMM_Model(cfgs, is_test=0, feature_extractor=fe)
    opt = torch.optim.Adam(list(enc.get_skip_params()) + list(enc.get_backbone_para()), lr=1e-3)
    frames = [synth.make_frame(P, F, H, W, 8, seed=7100 + b, kind="structured", with_targets=True) for b in range(B)]
    n_tplt = [3, 5]

    def boxes_of(masks):
        out = []
        for m in masks:
            ys, xs = np.where(m > 0.5)
            out.append([xs.min(), ys.min(), xs.max() + 1, ys.max() + 1] if len(xs) else [0, 0, 8, 8])
        return torch.from_numpy(np.asarray(out, np.float32)).to(DEV)

    img = torch.randn(B, 3, H, W, device=DEV)
    props, tboxes = [], []
    for fr in frames:
        bl = SimpleBoxList(boxes_of(fr.proposed_mask), (W, H))
        bl.add_field("mask", torch.from_numpy(fr.proposed_mask).to(DEV).unsqueeze(1))
        bl.add_field("scores", torch.from_numpy(fr.proposal_score).to(DEV))
        props.append(bl)
        tboxes.append(SimpleBoxList(boxes_of(fr.targets), (W, H)))
    mask_last = torch.stack([torch.from_numpy(fr.mask_last_occurence).to(DEV) for fr in frames])
    targets = torch.stack([torch.from_numpy(fr.targets).to(DEV) for fr in frames])
    valid = torch.zeros(B, F, device=DEV)
    for b in range(B):
        valid[b, :n_tplt[b]] = 1
    losses = []
    for it in range(6):
        feats = enc(img)
        tplt_dict = model.fill_template_dict(None, tboxes, feats, None, valid)        # templates from the GT boxes
        out, _, match_loss, last = model(None, props, feats["backbone_feature"], mask_last, tplt_dict, valid, targets)
        assert out.shape == (B, F, H, W) and len(match_loss) == B
        soft_iou = 1.0 - (out * targets).flatten(1).sum(1) / ((out + targets - out * targets).flatten(1).sum(1) + 1e-6)
        loss = soft_iou.mean() + sum(match_loss) / B
        opt.zero_grad()
        loss.backward()
        g_base, g_prop = enc.base.conv1.weight.grad, enc.prop2[0].weight.grad
        assert g_base is not None and torch.isfinite(g_base).all() and float(g_base.abs().sum()) > 0
        assert g_prop is not None and torch.isfinite(g_prop).all() and float(g_prop.abs().sum()) > 0
        opt.step()
        losses.append(float(sum(match_loss).detach()) / B)
    assert losses[-1] < losses[0], losses


def test_inference_step_end_to_end():
    """One inference step of the path with every stage from this package: encoder (MIOpen) -> proposal preprocessing
    (paste + tight boxes, NMS + top-k) -> ROI features -> DMM_Model.inference (one ragged batched launch sequence)."""
    from dmm_net_amd.dmm_model import DMM_Model
    from dmm_net_amd import proposals as prop
    torch.manual_seed(2)
    rng = np.random.default_rng(2)
    B, F, H, W, M28 = 3, 5, 96, 128, 28
    enc = FeatureEncoder("resnet34", hidden_size=32).to(DEV).eval()
    fe = FeatureExtractor()
    cfgs = {"matching": {"algo": "relax"}, "relax_max_iter": 40, "relax_proj_iter": 5, "relax_learning_rate": 0.1,
            "score_weight": 0.3, "encoder": {"nms_thresh": 0.4}, "sort_max_num": 50}
    model = DMM_Model(cfgs, is_test=1, feature_extractor=fe)
    img = torch.randn(B, 3, H, W, device=DEV)
    # raw detector output per image: boxes, scores, 28x28 mask probabilities (what the offline proposal files hold)
    raw = []
    for b in range(B):
        n = 60 + 7 * b
        x1, y1 = rng.uniform(0, W - 20, n), rng.uniform(0, H - 20, n)
        boxes = np.stack([x1, y1, np.minimum(x1 + rng.uniform(8, 60, n), W - 1), np.minimum(y1 + rng.uniform(8, 50, n), H - 1)], 1)
        bl = prop.SimpleBoxList(torch.from_numpy(boxes.astype(np.float32)).to(DEV), (W, H))
        bl.add_field("scores", torch.from_numpy(rng.random(n).astype(np.float32)).to(DEV))
        bl.add_field("mask", torch.from_numpy((rng.random((n, 1, M28, M28)) * 0.6 + 0.4).astype(np.float32)).to(DEV))
        raw.append(bl)
    with torch.no_grad():
        feats = enc(img)
        pasted = prop.forward_mask_prop([r.get_field("mask") for r in raw], raw, thresh=0.4, padding=1)
        kept = prop.filter_results(pasted, nms_thresh=cfgs["encoder"]["nms_thresh"], max_proposals=cfgs["sort_max_num"])
        for b, k in enumerate(kept):
            assert 0 < len(k) <= 50 and k.get_field("mask").shape == (len(k), 1, H, W)
        # templates: first-frame objects = the first n_obj kept proposals of each video
        n_obj = [2, 0, 4]
        tboxes = [prop.SimpleBoxList(kept[b].bbox[:max(n_obj[b], 1)].repeat(F, 1)[:F], (W, H)) for b in range(B)]
        valid = torch.zeros(B, F, device=DEV)
        mask_last = torch.zeros(B, F, H, W, device=DEV)
        for b in range(B):
            valid[b, :n_obj[b]] = 1
            mask_last[b, :n_obj[b]] = kept[b].get_field("mask")[:n_obj[b], 0]
        tplt_dict = model.fill_template_dict(None, tboxes, feats, None, valid)
        out, _, loss, last = model.inference({"args": None, "shape": None, "extra_frame": [0, 0, 0], "valid": valid},
                                             kept, feats["backbone_feature"], mask_last, tplt_dict)
    assert out.shape == (B, F, H, W) and loss == []
    assert float(out[1].abs().sum()) == 0.0 and torch.equal(last[1], mask_last[1])      # video without objects
    for b in (0, 2):
        assert float(out[b, n_obj[b]:].abs().sum()) == 0.0
        for o in range(n_obj[b]):
            # the template IS proposal o of this frame (identical mask and box): it must be matched to itself
            pm_o = kept[b].get_field("mask")[o, 0]
            w = (out[b, o] * pm_o).sum() / (pm_o * pm_o).sum()
            assert float((out[b, o] - w * pm_o).abs().max()) < 1e-5 and 0.3 < float(w) <= 1.2, (b, o, float(w))


def test_folded_graphed_encoder_matches_eager():
    """Inference encoder: BatchNorm folded into the convolutions and replayed from a captured HIP graph == eager."""
    from dmm_net_amd.encoder import GraphedEncoder, fold_batchnorm
    torch.manual_seed(4)
    enc = FeatureEncoder("resnet34", hidden_size=32).to(DEV)
    for m in enc.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.2)
            m.running_var.uniform_(0.5, 1.5)
    enc.eval()
    genc = GraphedEncoder(fold_batchnorm(enc))
    for seed in (0, 1, 2):                                   # replays with new inputs; second shape = second graph
        shape = (2, 3, 64, 96) if seed < 2 else (1, 3, 96, 64)
        img = torch.randn(*shape, device=DEV, generator=torch.Generator(device=DEV).manual_seed(seed))
        with torch.no_grad():
            ref = enc(img)
        out = genc(img)
        for k in ("backbone_feature", "refine_input_feat"):
            for x, y in zip(ref[k], out[k]):
                assert x.shape == y.shape
                assert float((x - y).abs().max()) <= 5e-4 * max(1.0, float(x.abs().max())), (seed, k)
    assert len(genc._graphs) == 2


# ------------------------------------------------------------------------------------ round 2: channels-last encoder
def test_bias_act_bf16_kernel_matches_fp32_reference():
    """dmm_bias_act_bf16: x = act(x + bias (+ residual)) in place on channels-last bf16 == the same in fp32, rounded once."""
    from dmm_net_amd.encoder import _bias_act_
    g = torch.Generator(device=DEV).manual_seed(9)
    for (B, C, H, W) in [(2, 64, 17, 23), (1, 8, 3, 5), (8, 256, 64, 64), (3, 2048, 8, 8), (2, 4, 9, 7), (1, 2, 5, 5)]:
        for use_res in (False, True):
            for relu in (False, True):
                x = torch.randn((B, C, H, W), generator=g, device=DEV).to(torch.bfloat16).contiguous(
                    memory_format=torch.channels_last)
                r = torch.randn((B, C, H, W), generator=g, device=DEV).to(torch.bfloat16).contiguous(
                    memory_format=torch.channels_last) if use_res else None
                b = torch.randn((C,), generator=g, device=DEV)
                ref = x.float() + b.view(1, C, 1, 1) + (r.float() if use_res else 0.0)
                if relu:
                    ref = ref.clamp_min(0.0)
                y = _bias_act_(x.clone(memory_format=torch.preserve_format), b, r, relu)
                assert torch.equal(y, ref.to(torch.bfloat16)), (B, C, H, W, use_res, relu)


@pytest.mark.parametrize("arch,hw", [("resnet50", (96, 128)), ("resnet34", (64, 96))])
def test_fast_channels_last_encoder_matches_the_folded_encoder(arch, hw):
    """FastEncoder (channels-last bf16, 1x1 convolutions as hipBLASLt GEMMs with fused epilogues, one HIP launch for
    bias + residual + ReLU) computes the function of fold_batchnorm(encoder): its distance to the fp32 result is that
    of the same network run eagerly in bf16 (both carry bf16 rounding of every layer), and it replays from a graph."""
    from dmm_net_amd.encoder import FastEncoder, GraphedEncoder, fold_batchnorm
    torch.manual_seed(7)
    enc = FeatureEncoder(arch, hidden_size=64).to(DEV)
    for m in enc.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.2)
            m.running_var.uniform_(0.5, 1.5)
    enc.eval()
    folded = fold_batchnorm(enc)
    img = torch.randn(2, 3, *hw, device=DEV)
    with torch.no_grad():
        ref = folded(img)
        import copy
        eager16 = copy.deepcopy(folded).to(torch.bfloat16)(img.to(torch.bfloat16))
    fast = FastEncoder(folded)
    out = fast(img)
    gout = GraphedEncoder(fast)(img)
    for k in ("backbone_feature", "refine_input_feat", "body_feature"):
        for x, y16, y, yg in zip(ref[k], eager16[k], out[k], gout[k]):
            assert x.shape == y.shape == yg.shape
            scale = max(1.0, float(x.abs().max()))
            e_eager = float((x - y16.float()).abs().max()) / scale
            e_fast = float((x - y.float()).abs().max()) / scale
            assert e_fast <= 2.0 * e_eager + 2e-2, (k, e_fast, e_eager)
            # graph replay == direct launches (MIOpen's split-K convolutions accumulate with atomics: rounding noise only)
            assert float((y.float() - yg.float()).abs().max()) <= 2e-2 * scale
    # channels-last, read in place by the NHWC form of the ROIAlign kernel
    assert all(p.is_contiguous(memory_format=torch.channels_last) for p in out["backbone_feature"])


@pytest.mark.parametrize("name,arch", [("r50", "resnet50"), ("r34", "resnet34")])
def test_fast_encoder_heads_match_reference_g16(name, arch):
    """G16 on the device: the reference's FeatureExtractorBase outputs (eval mode) against FastEncoder's head path --
    BatchNorm folded, channels-last bf16, MIOpen convolution + dmm_bias_act_bf16 epilogue -- within bf16 rounding."""
    from test_encoder_cpu import _g16_case
    from dmm_net_amd.encoder import FastEncoder
    g, hid, ker, body, sd = _g16_case(name)
    enc = FeatureEncoder(arch, hidden_size=hid, kernel_size=ker)
    enc.load_state_dict(sd, strict=False)
    fast = FastEncoder(enc.to(DEV).eval())
    cl = lambda t: t.to(DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    x = {5: cl(body[0]), 4: cl(body[1]), 3: cl(body[2]), 2: cl(body[3])}
    for lvl in (5, 4, 3, 2):
        outs = {f"x{lvl}_skip": fast._conv(x[lvl], getattr(fast.src, f"sk{lvl}"), relu=False),
                f"p{lvl}": fast._head(x[lvl], getattr(fast.src, f"prop{lvl}"))}
        for k, v in outs.items():
            exp = torch.from_numpy(g[f"{name}/eval/{k}"]).to(DEV)
            err = float((v.float() - exp).abs().max()) / max(1.0, float(exp.abs().max()))
            assert err <= 3e-2, (k, err)                      # bf16 inputs, weights and outputs; fp32 accumulation


def test_config3_bf16_path_is_bounded_against_fp32():
    """BASELINE config 3 (bf16 encoder) CHECKED, not only exercised: with trained-like BatchNorm statistics (eval mode)
    the bf16 inference encoder (FastEncoder: channels-last, library GEMMs with fused epilogues) -> NHWC ROI features ->
    cost + solver must stay close to the same pipeline on the fp32 folded encoder: the similarity table within a stated
    bound and the same proposal chosen for every template whose fp32 decision is not a near tie."""
    from conftest import record_achieved
    from dmm_net_amd import ops
    from dmm_net_amd.encoder import FastEncoder, fold_batchnorm
    torch.manual_seed(11)
    enc = FeatureEncoder("resnet50").to(DEV)
    for m in enc.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.2)
            m.running_var.uniform_(0.5, 1.5)
    enc.eval()
    folded = fold_batchnorm(enc)
    fast = FastEncoder(folded)
    fe = FeatureExtractor()
    B, P, O, H, W = 8, 50, 10, 255, 255
    img = torch.randn(B, 3, H, W, device=DEV)
    frames = [synth.make_frame(P, O, H, W, 8, seed=3100 + b, kind="structured") for b in range(B)]

    def tight(masks):
        out = []
        for m in masks:
            ys, xs = np.where(m > 0.5)
            out.append([xs.min(), ys.min(), xs.max() + 1, ys.max() + 1] if len(xs) else [0, 0, 8, 8])
        return Boxes(torch.from_numpy(np.asarray(out, np.float32)).to(DEV))
    pb, tb = [tight(fr.proposed_mask) for fr in frames], [tight(fr.mask_last_occurence) for fr in frames]
    pm = torch.stack([torch.from_numpy(fr.proposed_mask) for fr in frames]).to(DEV)
    tm = torch.stack([torch.from_numpy(fr.mask_last_occurence) for fr in frames]).to(DEV)
    sc = torch.stack([torch.from_numpy(fr.proposal_score) for fr in frames]).to(DEV)
    inter, ap, at = ops.iou_counts(pm, tm)
    res = {}
    with torch.no_grad():
        for tag, net in (("fp32", folded), ("bf16", fast)):
            bf = net(img)["backbone_feature"]
            pf, tf = fe(bf, pb).view(B, P, 512), fe(bf, tb).view(B, O, 512)
            cos = ops.cosine_features(tf, pf)
            res[tag] = ops.relax_match(cos, inter, ap, at, sc, score_weight=0.3, max_iter=20, proj_iter=5, lr=0.1, is_test=1)
            res[tag]["cos"] = cos
    assert res["bf16"]["cos"].dtype == torch.float32 and bool(torch.isfinite(res["bf16"]["sim"]).all())
    d_sim = float((res["bf16"]["sim"] - res["fp32"]["sim"]).abs().max())
    d_cos = float((res["bf16"]["cos"] - res["fp32"]["cos"]).abs().max())
    record_achieved("config3_bf16/sim_abs_err", d_sim)
    record_achieved("config3_bf16/cos_abs_err", d_cos)
    # sim = 0.7 cos + 0.3 iou, iou identical; achieved on the MI355X: cos 1.8e-3, sim 1.2e-3, every argmax equal
    assert d_cos <= 0.01 and d_sim <= 0.007 + 1e-6, (d_cos, d_sim)
    R32, R16 = res["fp32"]["R"][:, :, :P], res["bf16"]["R"][:, :, :P]
    top2 = R32.topk(2, dim=2).values
    decided = (top2[..., 0] - top2[..., 1]) > 0.1                             # fp32 decision is not a near tie
    same = R32.argmax(2) == R16.argmax(2)
    record_achieved("config3_bf16/argmax_agree_frac", float(same.float().mean()))
    assert int(decided.sum()) >= B * O // 2
    assert bool(same[decided].all()), (int((~same & decided).sum()), int(decided.sum()))


@pytest.mark.parametrize("stride,residual", [(1, False), (2, False), (1, True)])
def test_conv3x3_as_patch_matrix_gemm_matches_the_library_convolution(stride, residual):
    """FastEncoder's second form of a 3x3 convolution (dmm_im2col3x3_bf16 + dmm_conv1x1_bf16: patch matrix, ONE GEMM with
    bias (+ residual) + ReLU in the epilogue) against MIOpen's convolution + the bias / ReLU pass, and the patch matrix
    itself against torch's unfold, element for element."""
    import torch.nn as nn
    from dmm_net_amd import _lib
    from dmm_net_amd.encoder import FastEncoder, FeatureEncoder
    torch.manual_seed(5)
    B, C, H, W, Co = 3, 64, 13, 18, 96
    x = torch.randn(B, C, H, W, device=DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    cols = torch.empty((B * Ho * Wo, 9 * C), dtype=torch.bfloat16, device=DEV)
    _lib.check(_lib.load().dmm_im2col3x3_bf16(x.data_ptr(), B, H, W, C, stride, cols.data_ptr(),
                                              torch.cuda.current_stream().cuda_stream), "im2col")
    ref = torch.nn.functional.unfold(x.float(), 3, padding=1, stride=stride)            # [B, C * 9, L], (c, kh, kw) order
    ref = ref.view(B, C, 9, Ho * Wo).permute(0, 3, 2, 1).reshape(B * Ho * Wo, 9 * C)
    assert torch.equal(cols.float(), ref)
    fast = FastEncoder(FeatureEncoder("resnet34", hidden_size=16).to(DEV).eval())
    conv = nn.Conv2d(C, Co, 3, stride, 1).to(DEV)
    fast.src.add_module("extra", conv)
    fast._prepare()
    res = torch.randn(B, Co, Ho, Wo, device=DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last) \
        if residual else None
    a = fast._conv3x3_patches(x, conv, True, res)
    b = fast._conv_miopen(x, conv, True, res)
    assert a is not None and a.shape == b.shape and a.is_contiguous(memory_format=torch.channels_last)
    want = torch.relu(torch.nn.functional.conv2d(x.float(), conv.weight.to(torch.bfloat16).float(), conv.bias, stride, 1)
                      + (res.float() if residual else 0.0))
    err_a, err_b = float((a.float() - want).abs().max().detach()), float((b.float() - want).abs().max().detach())
    scale = float(want.abs().max().detach())
    assert err_a <= 1e-2 * scale and err_b <= 2e-2 * scale, (err_a, err_b, scale)     # bf16 output rounding: 2^-8 relative
    from conftest import record_achieved
    record_achieved(f"conv3x3_patch_gemm_vs_fp32/s{stride}r{int(residual)}", err_a / scale)


def test_stem_tail_bias_relu_maxpool_is_bit_identical_to_the_two_passes():
    """dmm_bias_relu_maxpool_bf16 (one pass over the stem convolution's output) == dmm_bias_act_bf16 then torch's
    3x3 / stride 2 / padding 1 max-pool, bit for bit (monotone operations commute with the maximum) -- odd sizes, negative
    and huge values included."""
    from dmm_net_amd import _lib
    from dmm_net_amd.encoder import _bias_act_
    torch.manual_seed(9)
    for (B, C, H, W) in [(2, 64, 13, 17), (1, 8, 1, 1), (3, 16, 128, 96)]:
        x = (torch.randn(B, C, H, W, device=DEV) * 3).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        x[0, :, 0, 0] = 1e30
        bias = torch.randn(C, device=DEV)
        want = torch.nn.functional.max_pool2d(_bias_act_(x.clone(memory_format=torch.channels_last), bias, None, True), 3, 2, 1)
        out = torch.empty_like(want, memory_format=torch.channels_last)
        _lib.check(_lib.load().dmm_bias_relu_maxpool_bf16(x.data_ptr(), bias.data_ptr(), B, H, W, C, out.data_ptr(),
                                                          torch.cuda.current_stream().cuda_stream), "stem tail")
        assert out.shape == want.shape and torch.equal(out, want)
