"""Encoder structure (no GPU needed): published parameter counts, checkpoint key names, strides, gradient flow."""
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
