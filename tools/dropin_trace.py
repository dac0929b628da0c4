#!/usr/bin/env python3
"""The drop-in's per-frame calls repeated, for a kernel timeline:

    rocprofv3 --kernel-trace --output-format csv -d DIR -- python tools/dropin_trace.py train|eval
    python tools/trace_overlap.py DIR/.../*_kernel_trace.csv 12

train: MatchModel(cfgs, 0).forward with targets + backward (50 proposals x 5 templates, 255 x 448, 10 x 5);
eval : MatchModel(cfgs, 1).forward under no_grad (40 x 5).  The host runs ahead of the device (a blocker is queued first), so
the timeline shows the device-side sequence of one call: kernels and the gaps between dependent launches."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from dmm_net_amd.match_model import MatchModel  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "train"
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
cfgs = {"matching": {"algo": "relax"}, "relax_max_iter": 10 if mode == "train" else 40, "relax_proj_iter": 5,
        "relax_learning_rate": 0.1, "score_weight": 0.3}
model = MatchModel(cfgs, is_test=0 if mode == "train" else 1)
blk = torch.empty((1 << 28,), dtype=torch.float32, device=dev)
blk2 = torch.empty_like(blk)


def call():
    if mode == "train":
        pf.grad = tf.grad = None
        fo, ms, ds, _, loss = model(pf, pm, [tf], tm, sc, tg)
        torch.autograd.backward([fo, loss["cost_loss"]], [dfull, one])
    else:
        with torch.no_grad():
            model(pf, pm, [tf], tm, sc)


for _ in range(10):
    call()
torch.cuda.synchronize()
for _ in range(12):
    blk2.copy_(blk)
for _ in range(40):
    call()
torch.cuda.synchronize()
print(mode, "done")
