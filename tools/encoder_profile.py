#!/usr/bin/env python3
"""Encoder kernel evidence.  Run on an MI355X:

    python tools/encoder_profile.py table  OUT.md      # per-convolution-shape table: FLOPs / device time vs bf16 MFMA peak
    rocprofv3 --kernel-trace --stats ... -- python tools/encoder_profile.py replay3    # config 3 graphed forward x 50
    rocprofv3 --kernel-trace --stats ... -- python tools/encoder_profile.py step4      # config 4 per-GPU training step x 5

``table`` uses torch.profiler (device activities, shapes, FLOP formulas) on the BatchNorm-folded bf16 ResNet-50 encoder
of BASELINE config 3 (8 frames of 255x255, NCHW): every aten::convolution call is one MIOpen launch (+ its bias add);
rows are grouped by input shape.  Peak = 2.5 PFLOP/s dense bf16 (MI355X_MICROARCH.md)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dmm_net_amd.encoder import FeatureEncoder, GraphedEncoder, fold_batchnorm

dev = "cuda:0"
PEAK = 2500.0
mode = sys.argv[1]
torch.manual_seed(0)

if mode in ("table", "replay3"):
    B, H, W = 8, 255, 255
    enc = fold_batchnorm(FeatureEncoder("resnet50").to(dev).eval())
    img = torch.randn(B, 3, H, W, device=dev)
    if mode == "replay3":
        g = GraphedEncoder(enc, weights_dtype=torch.bfloat16)
        for _ in range(50):
            g(img)
        torch.cuda.synchronize()
        sys.exit(0)
    from torch.profiler import ProfilerActivity, profile
    encb = enc.to(torch.bfloat16)
    x = img.to(torch.bfloat16)
    with torch.no_grad():
        for _ in range(5):
            encb(x)
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_flops=True) as prof:
            for _ in range(10):
                encb(x)
            torch.cuda.synchronize()
    rows = []
    tot_t = tot_f = 0.0
    for e in prof.key_averages(group_by_input_shape=True):
        t = e.device_time_total / 10.0            # us per forward (includes the op's child kernels)
        if e.key in ("aten::convolution", "aten::conv2d", "aten::_convolution"):
            continue
        if t <= 0:
            continue
        rows.append((e.key, str(e.input_shapes)[:90], e.count // 10, t, (e.flops or 0) / 10.0))
    rows.sort(key=lambda r: -r[3])
    kern = [(e.key, e.count // 10, e.self_device_time_total / 10.0) for e in prof.key_averages()
            if e.self_device_time_total > 0 and e.device_type is not None and "aten::" not in e.key]
    kern.sort(key=lambda r: -r[2])
    out = ["# Encoder (ResNet-50 + heads, BN folded, bf16, 8 x 255x255, NCHW, eager) -- per-op device time per forward",
           "", "| op | input shapes | calls | device us | GFLOP | TFLOP/s | frac of 2.5 PFLOP/s bf16 peak |", "|---|---|---|---|---|---|---|"]
    for k, shp, cnt, t, fl in rows[:40]:
        tf = fl / (t * 1e-6) / 1e12 if fl else 0.0
        out.append(f"| {k} | {shp} | {cnt} | {t:.1f} | {fl / 1e9:.3f} | {tf:.1f} | {tf / PEAK:.4f} |")
        tot_t += t
    out += ["", "## device kernels (self time per forward)", "", "| kernel | launches | us |", "|---|---|---|"]
    for k, cnt, t in kern[:30]:
        out.append(f"| {k[:110]} | {cnt} | {t:.1f} |")
    out.append("")
    out.append(f"total device kernel time per forward: {sum(t for _, _, t in kern):.1f} us in {sum(c for _, c, _ in kern)} launches")
    open(sys.argv[2], "w").write("\n".join(out) + "\n")
    print("\n".join(out[:30]))

elif mode == "step4":
    from dmm_net_amd.dmm_model import DMM_Model
    from dmm_net_amd.proposals import SimpleBoxList
    from dmm_net_amd.roi_features import FeatureExtractor
    B, F, P, H, W = 12, 5, 50, 255, 448
    g = torch.Generator(device=dev).manual_seed(0)
    enc = FeatureEncoder("resnet101").to(dev).train()
    cfgs = {"matching": {"algo": "relax"}, "relax_max_iter": 10, "relax_proj_iter": 5, "relax_learning_rate": 0.1,
            "score_weight": 0.3}
    model = DMM_Model(cfgs, is_test=0, feature_extractor=FeatureExtractor())
    opt = torch.optim.Adam(list(enc.get_skip_params()) + list(enc.get_backbone_para()), lr=1e-4)
    img = torch.randn(B, 3, H, W, device=dev)

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
    for it in range(int(os.environ.get('STEPS', '5'))):
        with torch.autocast("cuda", dtype=torch.bfloat16):
            feats = enc(img)
        tplt = model.fill_template_dict(None, tboxes, feats, None, valid)
        out, _, match_loss, last = model(None, props, feats["backbone_feature"], mask_last, tplt, valid, targets)
        soft = 1.0 - (out * targets).flatten(1).sum(1) / ((out + targets - out * targets).flatten(1).sum(1) + 1e-6)
        loss = soft.mean() + sum(match_loss) / B
        opt.zero_grad()
        loss.backward()
        opt.step()
    torch.cuda.synchronize()
