#!/usr/bin/env python3
"""The drop-in's per-frame calls repeated, for a kernel timeline:

    rocprofv3 --kernel-trace --output-format csv -d DIR -- python tools/dropin_trace.py train|eval
    python tools/trace_overlap.py DIR/.../*_kernel_trace.csv 12 [all]

train: MatchModel(cfgs, 0).forward with targets + backward (50 proposals x 5 templates, 255 x 448, 10 x 5);
eval : MatchModel(cfgs, 1).forward under no_grad (40 x 5);
model / model_eval: DMM_Model.forward + backward (10 x 5) / DMM_Model.inference (40 x 5) for 4 videos.  The host runs ahead of the device (a blocker is queued first), so
the timeline shows the device-side sequence of one call: kernels and the gaps between dependent launches."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from dmm_net_amd.match_model import MatchModel  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "train"      # train | eval | model (DMM_Model.forward + backward, 4 videos) | model_eval
dev = torch.device("cuda", 0)
P, O, H, W, D = 50, 5, 255, 448, 512
g = torch.Generator(device=dev).manual_seed(3)
pm = torch.rand((P, H, W), generator=g, device=dev)
tm = torch.rand((O, H, W), generator=g, device=dev)
tg = (torch.rand((O, H, W), generator=g, device=dev) > 0.5).float()
pf = torch.randn((P, D), generator=g, device=dev, requires_grad=mode == "train")
tf = torch.randn((O, D), generator=g, device=dev, requires_grad=mode == "train")
sc = torch.rand((P,), generator=g, device=dev)
dfull = torch.rand((O, H, W), generator=g, device=dev)
one = torch.ones((), device=dev)
cfgs = {"matching": {"algo": "relax"}, "relax_max_iter": 10 if mode in ("train", "model") else 40, "relax_proj_iter": 5,
        "relax_learning_rate": 0.1, "score_weight": 0.3}
model = MatchModel(cfgs, is_test=0 if mode == "train" else 1)
blk = torch.empty((1 << 28,), dtype=torch.float32, device=dev)
blk2 = torch.empty_like(blk)


if mode.startswith("model"):
    from dmm_net_amd.dmm_model import DMM_Model
    from dmm_net_amd.proposals import SimpleBoxList
    B, F = 4, 5
    props = []
    for b in range(B):
        x1 = torch.rand(P, generator=g, device=dev) * (W - 60)
        y1 = torch.rand(P, generator=g, device=dev) * (H - 60)
        bl = SimpleBoxList(torch.stack([x1, y1, x1 + 50, y1 + 40], 1), (W, H))
        bl.add_field("mask", torch.rand((P, 1, H, W), generator=g, device=dev))
        bl.add_field("scores", torch.rand(P, generator=g, device=dev))
        props.append(bl)
    feats = [torch.randn((P, D), generator=g, device=dev, requires_grad=mode == "model") for _ in range(B)]
    tplt = {b: {"feat": [torch.randn((F, D), generator=g, device=dev)], "refine_input_feat": [()]} for b in range(B)}
    mask_last = torch.rand((B, F, H, W), generator=g, device=dev)
    targets = (torch.rand((B, F, H, W), generator=g, device=dev) > 0.5).float()
    valid = torch.ones(B, F, device=dev)
    dm = DMM_Model(cfgs, is_test=0 if mode == "model" else 1, feature_extractor=lambda f, pr: torch.cat(feats, 0))
    infos = {"extra_frame": [False] * B, "valid": valid}


def call():
    if mode == "model":
        for x in feats:
            x.grad = None
        out, _, ml, _ = dm(None, props, None, mask_last, tplt, valid, targets)
        (out.sum() + sum(ml)).backward()
    elif mode == "model_eval":
        with torch.no_grad():
            dm.inference(infos, props, None, mask_last, tplt)
    elif mode == "train":
        pf.grad = tf.grad = None
        fo, ms, ds, _, loss = model(pf, pm, [tf], tm, sc, tg)
        torch.autograd.backward([fo, loss["cost_loss"]], [dfull, one])
    else:
        with torch.no_grad():
            model(pf, pm, [tf], tm, sc)


for _ in range(10):
    call()
torch.cuda.synchronize()
if len(sys.argv) > 2 and sys.argv[2] == "cprofile":          # where the HOST time of a call goes (python tools/dropin_trace.py model cprofile)
    import cProfile
    import pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(200):
        call()
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(28)
    sys.exit(0)
for _ in range(60 if mode.startswith("model") else 12):    # ~0.4 ms each: the host stays ahead of the device for the calls below
    blk2.copy_(blk)
for _ in range(40):
    call()
torch.cuda.synchronize()
print(mode, "done")
