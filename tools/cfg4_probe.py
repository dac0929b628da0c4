#!/usr/bin/env python3
"""Config 4's encoder step (ResNet-101 + heads, 12 x 3 x 255 x 448, forward + backward) under the settings that decide
its time on MI355X: memory format, dtype policy, who runs the 1x1 convolutions, MIOpen's find mode.

    python tools/cfg4_probe.py <variant> [--find] [--steps K]       one variant, one JSON line
    python tools/cfg4_probe.py matrix <outfile>                     every variant in its own process (env differs)

Variants: f32_nchw (the reference's setting), f32_nhwc, ac_nchw (bf16 autocast), ac_nhwc, bf16_nhwc (bf16 parameters),
          train (dmm_net_amd.encoder.TrainEncoder: the shipped bf16 training form).
"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one(variant, find, steps, frames):
    import torch
    from dmm_net_amd.encoder import FeatureEncoder
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    torch.backends.cudnn.benchmark = bool(find)
    enc = FeatureEncoder("resnet101").to(dev).train()
    img = torch.randn(frames, 3, 255, 448, device=dev)
    nhwc = variant.endswith("nhwc")
    ctx = lambda: torch.autocast("cuda", dtype=torch.bfloat16, enabled=variant.startswith("ac_"))
    if variant.startswith("train"):
        from dmm_net_amd import train_encoder
        train_encoder._DGRAD_AS_FORWARD = "bwddata" not in variant
        from dmm_net_amd.train_encoder import TrainEncoder
        enc = TrainEncoder(enc, graphs="nograph" not in variant, linear_1x1="nolin" not in variant,
                           fused_bn="nofuse" not in variant, own_wgrad="nowgrad" not in variant, overlap_wgrad="inline" not in variant, skips_need_grad=False,
                           layer3_parts=int(sys.argv[sys.argv.index("--l3") + 1]) if "--l3" in sys.argv else 3,
                           miopen_find=find)
        torch.backends.cudnn.benchmark = False
    elif nhwc:
        enc = enc.to(memory_format=torch.channels_last)
        img = img.contiguous(memory_format=torch.channels_last)
    if variant == "bf16_nhwc":
        enc = enc.to(torch.bfloat16)
        img = img.to(torch.bfloat16)
    params = [p for p in enc.parameters() if p.requires_grad]
    tgt = [torch.randn(frames, 128, -(-255 // s), -(-448 // s), device=dev) for s in (4, 8, 16, 32)]
    if "--check" in sys.argv:
        # gradients of the graphed form against the same functions run eagerly (same weights, same input)
        import copy, math
        eager = TrainEncoder(copy.deepcopy(enc.src), graphs=False, linear_1x1=enc.linear_1x1, fused_bn=enc.fused_bn,
                             own_wgrad=enc.own_wgrad, skips_need_grad=False)
        for k in range(3):
            for m, e in ((enc, "graph"), (eager, "eager")):
                for p in m.parameters():
                    p.grad = None
                f = m(img)
                sum((p.float() * tgt[k]).mean() for k, p in enumerate(f["backbone_feature"])).backward()
            num = den = 0.0
            for p, q in zip(enc.parameters(), eager.parameters()):
                if q.grad is not None:
                    num += float((p.grad - q.grad).square().sum()); den += float(q.grad.square().sum())
            out = max(float((a.float() - b.float()).abs().max()) for a, b in zip(enc(img)["backbone_feature"], eager(img)["backbone_feature"]))
            print(json.dumps({"check_step": k, "rel_grad_err_graph_vs_eager": math.sqrt(num / max(den, 1e-30)), "out_max_abs": out}), flush=True)

    # (a fixed pseudo-random direction per head output: sum(p^2) behind a BatchNorm has a zero gradient)
    tgt = [torch.randn(frames, 128, -(-255 // s), -(-448 // s), device=dev) for s in (4, 8, 16, 32)]

    adam = torch.optim.Adam(params, lr=1e-6, fused=True) if "--adam" in sys.argv else None

    clip = int(sys.argv[sys.argv.index("--clip") + 1]) if "--clip" in sys.argv else 1
    # --bn-groups G (train variants): the call's images as G BatchNorm statistics groups -- a clip's frames in ONE call with the
    # per-frame statistics of --clip's loop of calls
    bn_groups = int(sys.argv[sys.argv.index("--bn-groups") + 1]) if "--bn-groups" in sys.argv else 1
    imgs = [img] + [torch.randn_like(img) for _ in range(clip - 1)]

    def step(ev=None):
        if ev:
            ev[0].record()
        loss = 0.0
        for im in imgs:                                       # --clip T: T encoder calls, ONE backward (trainer.py:95-131)
            with ctx():
                feats = enc(im, bn_groups=bn_groups) if bn_groups > 1 else enc(im)
            loss = loss + sum((p.float() * tgt[k]).mean() for k, p in enumerate(feats["backbone_feature"]))
        if ev:
            ev[1].record()
        for p in params:
            p.grad = None
        loss.backward()
        if ev:
            ev[2].record()
        if adam is not None:                                 # (something between two steps, like a trainer has)
            adam.step()
        return loss

    t0 = time.perf_counter()
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    settle_s = time.perf_counter() - t0
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(steps)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for e in evs:
        loss = step(e)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / steps * 1e3
    fwd = sorted(e[0].elapsed_time(e[1]) for e in evs)[steps // 2]
    bwd = sorted(e[1].elapsed_time(e[2]) for e in evs)[steps // 2]
    if variant.startswith("train") and "nograph" not in variant and "--segments" in sys.argv:
        # replay time of every captured graph alone (no overlap): where the step's time sits, segment by segment
        plan = next(iter(enc._plans.values()))[0]
        def rep(g, n=20):
            for _ in range(3):
                g.replay()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(n):
                g.replay()
            b.record()
            torch.cuda.synchronize()
            return round(a.elapsed_time(b) / n, 3)
        class _Seq:
            def __init__(self, gs): self.gs = gs
            def replay(self):
                for g_ in self.gs: g_.replay()
        seg = {"fwd_body": [rep(g) for g in plan.fwd_body], "fwd_heads": [rep(g) for g in plan.fwd_head],
               "bwd_heads": [rep(g) for g in plan.bwd_head], "bwd_chain": {k: rep(g) for k, g in plan.bwd.items()},
               "bwd_wgrad": {k: rep(_Seq(gs)) for k, gs in plan.wgrad.items()}}
        print(json.dumps({"segments_ms": seg}), flush=True)
    print(json.dumps({"variant": variant, "find": bool(find), "frames": frames, "clip_calls": clip, "bn_groups": bn_groups,
                      "suggest_nhwc": os.environ.get("PYTORCH_MIOPEN_SUGGEST_NHWC"),
                      "suggest_nhwc_bn": os.environ.get("PYTORCH_MIOPEN_SUGGEST_NHWC_BATCHNORM"),
                      "ms_per_step_wall": round(wall, 3), "fwd_ms": round(fwd, 3), "bwd_ms": round(bwd, 3),
                      "settle_s": round(settle_s, 1), "loss": float(loss),
                      "mem_GB": round(torch.cuda.max_memory_allocated() / 2**30, 2)}), flush=True)


def matrix(out):
    rows = []
    plan = []
    for v in ("f32_nchw", "ac_nchw"):
        plan.append((v, {}, []))
    for v in ("f32_nhwc", "ac_nhwc", "bf16_nhwc"):
        plan.append((v, {}, []))
        plan.append((v, {"PYTORCH_MIOPEN_SUGGEST_NHWC": "1"}, []))
        plan.append((v, {"PYTORCH_MIOPEN_SUGGEST_NHWC": "1", "PYTORCH_MIOPEN_SUGGEST_NHWC_BATCHNORM": "1"}, []))
    plan.append(("ac_nhwc", {"PYTORCH_MIOPEN_SUGGEST_NHWC": "1"}, ["--find"]))
    plan.append(("bf16_nhwc", {"PYTORCH_MIOPEN_SUGGEST_NHWC": "1"}, ["--find"]))
    for v, env, extra in plan:
        e = dict(os.environ, **env)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), v] + extra, env=e, capture_output=True, text=True,
                           timeout=900)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        rows.append(line[-1] if line else json.dumps({"variant": v, "env": env, "error": r.stderr[-600:]}))
        print(rows[-1], flush=True)
        open(out, "w").write("\n".join(rows) + "\n")


if __name__ == "__main__":
    if sys.argv[1] == "matrix":
        matrix(sys.argv[2])
    else:
        a = sys.argv[2:]
        steps = int(a[a.index("--steps") + 1]) if "--steps" in a else 10
        frames = int(a[a.index("--frames") + 1]) if "--frames" in a else 12
        one(sys.argv[1], "--find" in a, steps, frames)
