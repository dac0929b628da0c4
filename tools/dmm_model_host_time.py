#!/usr/bin/env python3
"""Host time of one DMM_Model call for 4 videos, by part (forward / the caller's loss arithmetic / backward) and by function
(cProfile of the forward and of the backward): what keeps `--config dropin`'s dmm_model_* cases host bound."""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dmm_net_amd.dmm_model import DMM_Model
from dmm_net_amd.proposals import SimpleBoxList

dev = torch.device("cuda", 0)
P, F, H, W, D, B = 50, 5, 255, 448, 512, 4
g = torch.Generator(device=dev).manual_seed(3)
cfgs = lambda mi: {"matching": {"algo": "relax"}, "relax_max_iter": mi, "relax_proj_iter": 5, "relax_learning_rate": 0.1,
                   "score_weight": 0.3}
props = []
for b in range(B):
    x1 = torch.rand(P, generator=g, device=dev) * (W - 60)
    y1 = torch.rand(P, generator=g, device=dev) * (H - 60)
    bl = SimpleBoxList(torch.stack([x1, y1, x1 + 50, y1 + 40], 1), (W, H))
    bl.add_field("mask", torch.rand((P, 1, H, W), generator=g, device=dev))
    bl.add_field("scores", torch.rand(P, generator=g, device=dev))
    props.append(bl)
feats = [torch.randn((P, D), generator=g, device=dev, requires_grad=True) for _ in range(B)]
tplt = {b: {"feat": [torch.randn((F, D), generator=g, device=dev)], "refine_input_feat": [()]} for b in range(B)}
mask_last = torch.rand((B, F, H, W), generator=g, device=dev)
targets = (torch.rand((B, F, H, W), generator=g, device=dev) > 0.5).float()
valid = torch.ones(B, F, device=dev)
m_tr = DMM_Model(cfgs(10), is_test=0, feature_extractor=lambda f, pr: torch.cat(feats, 0))
m_ev = DMM_Model(cfgs(40), is_test=1, feature_extractor=lambda f, pr: torch.cat([x.detach() for x in feats], 0))
infos = {"extra_frame": [False] * B, "valid": valid}
t_f = t_l = t_b = t_i = 0.0
N = 300
import gc
for k in range(N + 20):
    if k == 20:
        torch.cuda.synchronize(); gc.collect(); gc.disable(); t_f = t_l = t_b = t_i = 0.0
    t0 = time.perf_counter()
    out, _, ml, _ = m_tr(None, props, None, mask_last, tplt, valid, targets)
    t1 = time.perf_counter()
    loss = out.sum() + sum(ml)
    t2 = time.perf_counter()
    loss.backward()
    t3 = time.perf_counter()
    with torch.no_grad():
        m_ev.inference(infos, props, None, mask_last, tplt)
    t4 = time.perf_counter()
    t_f += t1 - t0; t_l += t2 - t1; t_b += t3 - t2; t_i += t4 - t3
    if k % 50 == 49:
        torch.cuda.synchronize()
print(f"host us per call: forward {t_f / N * 1e6:.1f}  caller's loss {t_l / N * 1e6:.1f}  backward {t_b / N * 1e6:.1f}  inference {t_i / N * 1e6:.1f}")
for name, fn in (("forward", lambda: m_tr(None, props, None, mask_last, tplt, valid, targets)),
                 ("inference", lambda: m_ev.inference(infos, props, None, mask_last, tplt))):
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(200):
        if name == "inference":
            with torch.no_grad():
                fn()
        else:
            fn()
    pr.disable()
    torch.cuda.synchronize()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22)
    print("====", name); print("\n".join(s.getvalue().splitlines()[:40]))
