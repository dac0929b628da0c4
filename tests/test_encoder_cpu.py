"""Encoder structure (no GPU needed): published parameter counts, checkpoint key names, strides, gradient flow."""
import pytest
import torch

from dmm_net_amd.encoder import FeatureEncoder


def test_encoder_structure():
    # published architecture sizes: torchvision ResNet-50 = 25,557,032 parameters, ResNet-101 = 44,549,160
    for name, n in (("resnet50", 25557032), ("resnet101", 44549160)):
        enc = FeatureEncoder(name)
        assert sum(p.numel() for p in enc.base.parameters()) == n
        keys = enc.state_dict().keys()
        for k in ("base.conv1.weight", "base.layer1.0.downsample.0.weight", "base.layer4.2.bn3.running_var",
                  "base.fc.bias", "sk5.weight", "bn2.running_mean", "prop3.0.weight", "prop3.4.bias"):
            assert k in keys, k




def test_encoder_forward_shapes_and_grad_cpu():
    enc = FeatureEncoder("resnet34", hidden_size=32)
    img = torch.randn(2, 3, 64, 96)
    f = enc(img)
    assert [tuple(t.shape[1:]) for t in f["backbone_feature"]] == [(32, 16, 24), (32, 8, 12), (32, 4, 6), (32, 2, 3)]
    assert [t.shape[1] for t in f["refine_input_feat"]] == [32, 32, 16, 8]
    assert [t.shape[1] for t in f["body_feature"]] == [64, 128, 256, 512]
    sum(t.sum() for t in f["backbone_feature"]).backward()
    assert enc.base.conv1.weight.grad is not None and enc.prop2[0].weight.grad is not None
    assert enc.sk5.weight.grad is None                       # skip heads feed the decoder, not the matching path
    assert len(enc.get_skip_params()) == 4 * 2 + 4 * 2 + 4 * 8


def test_fold_batchnorm_same_outputs_cpu():
    from dmm_net_amd.encoder import fold_batchnorm
    torch.manual_seed(0)
    for name in ("resnet34", "resnet50"):
        enc = FeatureEncoder(name, hidden_size=32)
        for m in enc.modules():                              # non-trivial running statistics / affine parameters
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.2)
                m.running_var.uniform_(0.5, 1.5)
                m.weight.data.uniform_(0.5, 1.5)
                m.bias.data.normal_(0, 0.2)
        enc.eval()
        folded = fold_batchnorm(enc)
        assert not any(isinstance(m, torch.nn.BatchNorm2d) for m in folded.modules())
        img = torch.randn(2, 3, 64, 96)
        with torch.no_grad():
            a, b = enc(img), folded(img)
        for k in ("backbone_feature", "refine_input_feat", "body_feature"):
            for x, y in zip(a[k], b[k]):
                assert x.shape == y.shape
                assert float((x - y).abs().max()) <= 2e-4 * max(1.0, float(x.abs().max())), (name, k)
        assert any(isinstance(m, torch.nn.BatchNorm2d) for m in enc.modules())     # the original is untouched


def _g16_case(name):
    import numpy as np
    from conftest import golden
    g = golden("g16_encoder_heads")
    hid, ker, H, W = [int(v) for v in g[f"{name}/cfg"]]
    gen = torch.Generator().manual_seed(16)
    shapes = g[f"{name}/body_shapes"]
    body = [torch.randn(*[int(v) for v in shp], generator=gen) for shp in shapes]          # x5, x4, x3, x2
    for t, c in zip(body, g[f"{name}/body_checksum"]):
        assert abs(float(t.double().sum()) - float(c)) < 1e-6, "torch's CPU randn stream changed: regenerate G16"
    sd = {k[len(name) + 4:]: torch.from_numpy(np.asarray(g[k])) for k in g.keys() if k.startswith(f"{name}/sd/")}
    sd = {k: (v.float() if v.dtype == torch.float16 else v) for k, v in sd.items()}
    return g, hid, ker, body, sd


@pytest.mark.parametrize("name,arch", [("r50", "resnet50"), ("r34", "resnet34")])
def test_encoder_heads_match_the_reference_feature_extractor_base(name, arch):
    """G16: the sk / bn / prop heads against the reference's own FeatureExtractorBase (dmm/modules/base.py:18-69,
    imported) driven like model_encoder.py:136-146 -- the reference's state dict loads into FeatureEncoder by key
    (strict for the head keys) and the outputs agree in eval() and train() mode; get_skip_params covers the same
    parameters."""
    g, hid, ker, body, sd = _g16_case(name)
    enc = FeatureEncoder(arch, hidden_size=hid, kernel_size=ker)
    own = {k for k in enc.state_dict().keys() if not k.startswith("base.")}
    assert own == set(sd.keys()), (sorted(own - set(sd.keys())), sorted(set(sd.keys()) - own))
    missing, unexpected = enc.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.startswith("base.") for k in missing)
    assert sum(p.numel() for p in enc.get_skip_params()) == int(g[f"{name}/n_skip_params"])
    x5, x4, x3, x2 = body
    for mode in ("eval", "train"):
        enc.train(mode == "train")
        keep = {k: v.clone() for k, v in enc.state_dict().items()}
        with torch.no_grad():
            outs = {"x5_skip": enc.bn5(enc.sk5(x5)), "x4_skip": enc.bn4(enc.sk4(x4)), "x3_skip": enc.bn3(enc.sk3(x3)),
                    "x2_skip": enc.bn2(enc.sk2(x2)), "p5": enc.prop5(x5), "p4": enc.prop4(x4), "p3": enc.prop3(x3),
                    "p2": enc.prop2(x2)}
        enc.load_state_dict(keep)
        for k, v in outs.items():
            exp = torch.from_numpy(g[f"{name}/{mode}/{k}"])
            assert v.shape == exp.shape, k
            assert float((v - exp).abs().max()) <= 1e-5 * max(1.0, float(exp.abs().max())), (mode, k)


def test_vgg16_backbone_option_cpu():
    """``base_model='vgg16'`` (model_encoder.py:48-49, vision.py:57-115): the 13 convolutions of torchvision's VGG-16
    ``features`` (14,714,688 parameters, flat numbering 0..30), taps = the five pooled stage outputs with
    ``get_skip_dims('vgg16')`` channels; heads, folding and gradients as for the ResNets; unknown names raise like the
    reference."""
    from dmm_net_amd.encoder import FastEncoder, fold_batchnorm, get_skip_dims
    enc = FeatureEncoder("vgg16", hidden_size=32)
    assert sum(p.numel() for p in enc.base.parameters()) == 14714688
    keys = enc.state_dict().keys()
    for k in ("base.features.0.weight", "base.features.28.bias", "sk5.weight", "prop2.3.weight"):
        assert k in keys, k
    assert isinstance(enc.base.features[30], torch.nn.MaxPool2d) and enc.base.taps == [4, 9, 16, 23, 30]
    img = torch.randn(2, 3, 64, 96)
    x5, x4, x3, x2, x1 = enc.base(img)
    assert [t.shape[1] for t in (x5, x4, x3, x2, x1)] == get_skip_dims("vgg16")
    assert [tuple(t.shape[2:]) for t in (x1, x2, x3, x4, x5)] == [(32, 48), (16, 24), (8, 12), (4, 6), (2, 3)]
    f = enc(img)
    assert [tuple(t.shape[1:]) for t in f["backbone_feature"]] == [(32, 16, 24), (32, 8, 12), (32, 4, 6), (32, 2, 3)]
    sum(t.sum() for t in f["backbone_feature"]).backward()
    assert enc.base.features[0].weight.grad is not None
    enc.eval()
    folded = fold_batchnorm(enc)
    with torch.no_grad():
        a, b = enc(img), folded(img)
    for x, y in zip(a["backbone_feature"], b["backbone_feature"]):
        assert float((x - y).abs().max()) <= 2e-4 * max(1.0, float(x.abs().max()))
    with pytest.raises(NotImplementedError):
        FastEncoder(folded)
    with pytest.raises(Exception, match="not supported"):
        FeatureEncoder("alexnet")


def test_vgg16_loads_a_state_dict_that_carries_classifier_keys():
    """The reference's VGG16 / torchvision state dicts hold classifier.{0,3,6}.* (vision.py:57-76); the body here builds no
    classifier and drops those keys on load: a STRICT load works, directly and as the encoder's ``base``."""
    import torch
    from dmm_net_amd.encoder import VGG16, FeatureEncoder
    body = VGG16()
    sd = {k: v.clone() for k, v in body.state_dict().items()}
    sd["features.0.weight"] += 1.0
    for i, shape in ((0, (8, 4)), (3, (8, 8)), (6, (10, 8))):
        sd[f"classifier.{i}.weight"] = torch.zeros(shape)
        sd[f"classifier.{i}.bias"] = torch.zeros(shape[0])
    VGG16().load_state_dict(sd, strict=True)
    enc = FeatureEncoder("vgg16", hidden_size=16)
    full = {k: v.clone() for k, v in enc.state_dict().items()}
    full.update({"base." + k: v for k, v in sd.items()})
    enc.load_state_dict(full, strict=True)
    assert torch.equal(enc.base.features[0].weight, sd["features.0.weight"])



def test_train_encoder_segments_compute_the_encoder_on_the_cpu():
    """``train_encoder.TrainEncoder`` off the GPU runs its segment functions eagerly under autograd: in fp32 mode they ARE the
    encoder (same outputs, gradients and BatchNorm buffers as ``FeatureEncoder``; the 1x1 convolutions as products of the
    activation matrix), in bf16 mode they give finite gradients for the same set of parameters; with ``skips_need_grad=False``
    the decoder's skip projections get none, like under autograd when nothing consumes them."""
    import copy
    import math
    import torch
    from dmm_net_amd.encoder import FeatureEncoder
    from dmm_net_amd.train_encoder import TrainEncoder
    torch.manual_seed(0)
    ref = FeatureEncoder("resnet50", hidden_size=32).train()
    enc = copy.deepcopy(ref)
    te = TrainEncoder(enc, dtype=torch.float32)
    img = torch.randn(2, 3, 64, 96)
    o, r = te(img), ref(img)
    outs = lambda f: f["backbone_feature"] + f["refine_input_feat"]
    for a, b in zip(outs(o), outs(r)):
        assert a.shape == b.shape and float((a.float() - b).abs().max()) <= 2e-3 * float(b.abs().max())
    w = [torch.sin(torch.arange(t.numel(), dtype=torch.float32).view(t.shape) * 0.37 + k) for k, t in enumerate(outs(r))]
    sum((a.float() * t).mean() for a, t in zip(outs(o), w)).backward()
    sum((a * t).mean() for a, t in zip(outs(r), w)).backward()
    num = den = 0.0
    for (n, p), (_, q) in zip(enc.named_parameters(), ref.named_parameters()):
        assert (p.grad is None) == (q.grad is None), n
        if q.grad is not None:
            num += float((p.grad - q.grad).square().sum())
            den += float(q.grad.square().sum())
    assert math.sqrt(num / den) <= 2e-2                 # (a random-init ResNet-50 amplifies fp32 summation-order noise)
    for (n, a), (_, b) in zip(enc.named_buffers(), ref.named_buffers()):
        assert float((a.float() - b.float()).abs().max()) <= 1e-3 * (float(b.float().abs().max()) + 1.0), n
    half = TrainEncoder(copy.deepcopy(ref), skips_need_grad=False)
    f = half(img)
    assert f["backbone_feature"][0].dtype == torch.float32 and not f["refine_input_feat"][0].requires_grad
    sum(p.float().mean() for p in f["backbone_feature"]).backward()
    for n, p in half.src.named_parameters():
        if n.startswith(("sk", "bn")) or n.startswith("base.fc"):
            assert p.grad is None, n
        else:
            assert p.grad is not None and bool(torch.isfinite(p.grad).all()), n


def test_train_encoder_clip_call_keeps_the_per_frame_batchnorm_statistics_on_the_cpu():
    """``TrainEncoder.forward(cat(frames), bn_groups=T)`` == T calls of the encoder, one per frame step (the reference's
    trainer loop, trainer.py:95-131): features per frame, running statistics after T updates, ``num_batches_tracked``."""
    import copy
    import torch
    from dmm_net_amd.encoder import FeatureEncoder
    from dmm_net_amd.train_encoder import TrainEncoder
    torch.manual_seed(1)
    ref = FeatureEncoder("resnet34", hidden_size=16).train()
    enc = copy.deepcopy(ref)
    te = TrainEncoder(enc, dtype=torch.float32)
    frames = [torch.randn(2, 3, 48, 64) for _ in range(3)]
    per_frame = [ref(f) for f in frames]
    clip = te(torch.cat(frames, 0), bn_groups=3)
    for t, fr in enumerate(per_frame):
        for a, b in zip(clip["backbone_feature"] + clip["refine_input_feat"], fr["backbone_feature"] + fr["refine_input_feat"]):
            assert float((a[2 * t:2 * t + 2].float() - b).abs().max()) <= 2e-3 * float(b.abs().max()), t
    for (n, a), (_, b) in zip(enc.named_buffers(), ref.named_buffers()):
        assert float((a.float() - b.float()).abs().max()) <= 1e-4 * (float(b.float().abs().max()) + 1.0), n
    assert int(enc.base.bn1.num_batches_tracked) == 3
    one = te(torch.cat(frames, 0))                               # (one statistics group: a different normalisation)
    assert float((one["backbone_feature"][0] - clip["backbone_feature"][0]).abs().max()) > 1e-3
    import pytest
    with pytest.raises(AssertionError):
        te(torch.cat(frames, 0), bn_groups=4)                    # 6 images do not split into 4 groups
