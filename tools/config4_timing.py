#!/usr/bin/env python3
"""BASELINE config 4, the per-GPU share on ONE MI355X: 4 videos x clip 3 = 12 frames of [3,255,448], ResNet-101
encoder + prop heads (MIOpen, fp32 / bf16 autocast), 50 proposals, 5 template slots -> ROI features (HIP) ->
DMM_Model training forward (ragged batched HIP layer, dual IoU with the targets) -> loss -> backward (HIP layer,
ROI scatter, MIOpen) -> Adam.  Prints ms per stage.  The RCCL gradient mean of the 8-GPU job is not part of this
single-GPU tool (distributed.GradBucketer; 222 MB of fp32 gradients per step)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dmm_net_amd.dmm_model import DMM_Model
from dmm_net_amd.encoder import FeatureEncoder
from dmm_net_amd.proposals import SimpleBoxList
from dmm_net_amd.roi_features import FeatureExtractor

dev = "cuda:0"
B, F, P, H, W = 12, 5, 50, 255, 448
torch.manual_seed(0)
g = torch.Generator(device=dev).manual_seed(0)
enc = FeatureEncoder("resnet101").to(dev).train()
CL = bool(os.environ.get("CHANNELS_LAST"))
if CL:
    enc = enc.to(memory_format=torch.channels_last)
fe = FeatureExtractor()
cfgs = {"matching": {"algo": "relax"}, "relax_max_iter": 10, "relax_proj_iter": 5, "relax_learning_rate": 0.1,
        "score_weight": 0.3}
model = DMM_Model(cfgs, is_test=0, feature_extractor=fe)
opt = torch.optim.Adam(list(enc.get_skip_params()) + list(enc.get_backbone_para()), lr=1e-4, fused=True)
img = torch.randn(B, 3, H, W, device=dev)
if CL:
    img = img.contiguous(memory_format=torch.channels_last)


def boxes(n):
    x1 = torch.rand(n, generator=g, device=dev) * (W - 60)
    y1 = torch.rand(n, generator=g, device=dev) * (H - 60)
    return torch.stack([x1, y1, x1 + 10 + torch.rand(n, generator=g, device=dev) * 150,
                        y1 + 10 + torch.rand(n, generator=g, device=dev) * 100], 1).clamp(max=W - 1)


props, tboxes = [], []
for b in range(B):
    bl = SimpleBoxList(boxes(P), (W, H))
    bl.add_field("mask", torch.rand((P, 1, H, W), generator=g, device=dev))
    bl.add_field("scores", torch.rand(P, generator=g, device=dev))
    props.append(bl)
    tboxes.append(SimpleBoxList(boxes(F), (W, H)))
mask_last = torch.rand((B, F, H, W), generator=g, device=dev)
targets = (torch.rand((B, F, H, W), generator=g, device=dev) > 0.5).float()
valid = torch.ones(B, F, device=dev)


def ev():
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


for tag, dt in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
    acc = [0.0] * 5
    n, warm = 8, 3
    for it in range(n + warm):
        e0 = ev()
        with torch.autocast("cuda", dtype=dt, enabled=dt != torch.float32):
            feats = enc(img)
        e1 = ev()
        tplt = model.fill_template_dict(None, tboxes, feats, None, valid)
        out, _, match_loss, last = model(None, props, feats["backbone_feature"], mask_last, tplt, valid, targets)
        soft = 1.0 - (out * targets).flatten(1).sum(1) / ((out + targets - out * targets).flatten(1).sum(1) + 1e-6)
        loss = soft.mean() + sum(match_loss) / B
        e2 = ev()
        opt.zero_grad()
        loss.backward()
        e3 = ev()
        opt.step()
        e4 = ev()
        torch.cuda.synchronize()
        if it >= warm:
            for k, (a, b) in enumerate(((e0, e1), (e1, e2), (e2, e3), (e3, e4), (e0, e4))):
                acc[k] += a.elapsed_time(b) / n
    print(f"config 4 per-GPU step [{tag}], {B} frames: encoder fwd {acc[0]:.2f} ms | ROI + matching layer + loss fwd "
          f"{acc[1]:.2f} ms | backward {acc[2]:.2f} ms | Adam {acc[3]:.2f} ms | step {acc[4]:.2f} ms = "
          f"{B / acc[4] * 1e3:.0f} frames/s per GPU")

if os.environ.get("PROFILE"):
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(3):
            feats = enc(img)
            tplt = model.fill_template_dict(None, tboxes, feats, None, valid)
            out, _, match_loss, last = model(None, props, feats["backbone_feature"], mask_last, tplt, valid, targets)
            soft = 1.0 - (out * targets).flatten(1).sum(1) / ((out + targets - out * targets).flatten(1).sum(1) + 1e-6)
            opt.zero_grad()
            (soft.mean() + sum(match_loss) / B).backward()
        torch.cuda.synchronize()
    rows = [(e.key, e.self_device_time_total / 3e3, e.count // 3) for e in prof.key_averages() if "dmm::" in e.key]
    for k, t, c in sorted(rows, key=lambda r: -r[1]):
        print(f"  {k[:90]:90s} {t:8.3f} ms/step  x{c}")
