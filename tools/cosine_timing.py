"""Feature-similarity kernel timing: D-over-lanes kernel (dmm_cosine_lanes.hip) vs the tile kernel
(option COSINE_KERNEL = 1), device time per launch over back-to-back launches, plus a bit-equality check against the
three-launch path.  Usage: python tools/cosine_timing.py [lanes|tile]   (no argument: runs both as child processes)."""
import os
import subprocess
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
SHAPES = [(1, 50, 10, 512), (4, 50, 10, 512), (64, 50, 10, 512), (512, 50, 10, 512), (1024, 50, 10, 512),
          (1, 200, 20, 512), (64, 200, 20, 512), (512, 200, 20, 512), (512, 200, 20, 256), (256, 64, 16, 1024)]


def child():
    import torch
    from dmm_net_amd import _lib, ops
    _lib.set_option("COSINE_KERNEL", 1 if sys.argv[1] == "tile" else 0)
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(5)
    for (B, N, M, D) in SHAPES:
        tf = torch.randn((B, M, D), device=dev, generator=g)
        pf = torch.randn((B, N, D), device=dev, generator=g)
        pf = torch.relu(pf)                                           # sparse rows like pooled ReLU features
        a = ops.cosine_features(tf, pf)
        ref = ops.cosine(ops.feature_normalize(tf), ops.feature_normalize(pf))
        same = bool(torch.equal(a, ref))
        for _ in range(5):
            ops.cosine_features(tf, pf)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 50
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            ops.cosine_features(tf, pf)
        e1.record()
        torch.cuda.synchronize()
        print(f"  B={B:5d} N={N:3d} M={M:2d} D={D:4d}: {e0.elapsed_time(e1) / reps * 1e3:9.1f} us  bit-identical={same}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child()
    else:
        for kind in ("lanes", "tile"):
            print(kind, flush=True)
            subprocess.run([sys.executable, os.path.abspath(__file__), kind], check=False)
