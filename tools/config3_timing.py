#!/usr/bin/env python3
"""BASELINE config 3 on one MI355X: ResNet-50 + prop heads (MIOpen, bf16 autocast, channels_last) -> fused 4-level
ROIAlign+mean (HIP) -> cost + solver + mix (HIP), batch of 8 frames with 50 proposals x 10 templates at 255x255.
Prints ms per stage (HIP events) and frames/s of the whole path.  Random-init weights, synthetic frames."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dmm_net_amd import ops
from dmm_net_amd.encoder import FeatureEncoder, GraphedEncoder, fold_batchnorm
from dmm_net_amd.roi_features import FeatureExtractor
from dmm_net_amd.proposals import SimpleBoxList

dev = "cuda:0"
B, P, O, H, W, D = 8, 50, 10, 255, 255, 512
torch.manual_seed(0)
enc = FeatureEncoder("resnet50").to(dev).eval().to(memory_format=torch.channels_last)
fe = FeatureExtractor()
img = torch.randn(B, 3, H, W, device=dev).contiguous(memory_format=torch.channels_last)
g = torch.Generator(device=dev).manual_seed(3)
pm = torch.rand((B, P, H, W), generator=g, device=dev)
tm = torch.rand((B, O, H, W), generator=g, device=dev)
sc = torch.rand((B, P), generator=g, device=dev)


def boxes(n):
    x1 = torch.rand(n, generator=g, device=dev) * (W - 40)
    y1 = torch.rand(n, generator=g, device=dev) * (H - 40)
    w = 8 + torch.rand(n, generator=g, device=dev) * 100
    h = 8 + torch.rand(n, generator=g, device=dev) * 100
    return torch.stack([x1, y1, (x1 + w).clamp(max=W - 1), (y1 + h).clamp(max=H - 1)], 1)


pbox = [SimpleBoxList(boxes(P), (W, H)) for _ in range(B)]
tbox = [SimpleBoxList(boxes(O), (W, H)) for _ in range(B)]
plan = ops.ForwardPlan(B, P, O, H, W, D, dev, pipeline=False)


def timed(fn, n=20, warm=5):
    for _ in range(warm):
        out = fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        out = fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n, out


for tag, dt in (("bf16", torch.bfloat16), ("fp32", torch.float32)):
    def encode():
        with torch.no_grad(), torch.autocast("cuda", dtype=dt, enabled=dt != torch.float32):
            return enc(img)["backbone_feature"]
    t_enc, bf = timed(encode)
    t_roi, (pf, tf) = timed(lambda: (fe(bf, pbox).view(B, P, D), fe(bf, tbox).view(B, O, D)))
    t_layer, _ = timed(lambda: plan.run(pm, tm, pf, tf, sc, score_weight=0.3, max_iter=20, proj_iter=5, lr=0.1, is_test=1))

    def whole():
        f = encode()
        a, b = fe(f, pbox).view(B, P, D), fe(f, tbox).view(B, O, D)
        plan.run(pm, tm, a, b, sc, score_weight=0.3, max_iter=20, proj_iter=5, lr=0.1, is_test=1)
    t_all, _ = timed(whole)
    # product inference encoder: NCHW (faster than channels_last with this MIOpen), BatchNorm folded, one HIP graph
    enc_nchw = fold_batchnorm(FeatureEncoder("resnet50").to(dev).eval())
    enc_nchw.load_state_dict(fold_batchnorm(enc.to(memory_format=torch.contiguous_format)).state_dict())
    enc.to(memory_format=torch.channels_last)
    img_nchw = img.contiguous()
    genc_ = GraphedEncoder(enc_nchw, weights_dtype=None if dt == torch.float32 else dt)
    genc = lambda x: genc_(img_nchw)
    t_genc, gbf = timed(lambda: genc(img)["backbone_feature"])
    err = max(float((a.float() - b.float()).abs().max()) for a, b in zip(gbf, bf))

    def whole_graphed():
        f = genc(img)["backbone_feature"]
        a, b = fe(f, pbox).view(B, P, D), fe(f, tbox).view(B, O, D)
        plan.run(pm, tm, a, b, sc, score_weight=0.3, max_iter=20, proj_iter=5, lr=0.1, is_test=1)
    t_gall, _ = timed(whole_graphed)
    print(f"config 3 [{tag}] B={B}: NCHW + BN-folded + HIP-graph encoder {t_genc:.3f} ms (max |diff| vs eager {err:.2e}); whole path with it "
          f"{t_gall:.3f} ms = {B / t_gall * 1e3:.0f} frames/s")
    print(f"config 3 [{tag}] B={B}: encoder {t_enc:.3f} ms | ROI features (60 rois/frame x 4 levels) {t_roi:.3f} ms | "
          f"matching layer {t_layer:.3f} ms | whole path {t_all:.3f} ms = {B / t_all * 1e3:.0f} frames/s")
