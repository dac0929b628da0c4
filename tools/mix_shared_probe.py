#!/usr/bin/env python3
"""tools/mix_shared_probe.hip driver: GB/s of the train-mode mix's access patterns (see the .hip header)."""
import ctypes, os, subprocess, torch
HERE = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(HERE, "libmix_shared_probe.so")
if not os.path.exists(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC",
                           os.path.join(HERE, "mix_shared_probe.hip"), "-o", so])
L = ctypes.CDLL(so)
vp, ci = ctypes.c_void_p, ctypes.c_int
L.probe.argtypes = [ci, ci, vp, vp, ci, ci, ci, ci, ci, vp, vp, ci]
dev = "cuda:0"
B, P, M, HW = 512, 48, 10, 65025
src = torch.rand((B, P, HW), device=dev)
dst = torch.empty((B, M, 65056), device=dev)
sink = torch.zeros(1 << 16, device=dev)
st = torch.cuda.current_stream().cuda_stream
names = {0: "A: 1 KiB per wave and plane, 8 planes in flight", 1: "A: ..., 4 planes in flight", 4: "A: ..., 16 planes in flight",
         2: "C: wave = 4 KiB run of one plane, 4 planes per WG in flight", 3: "C: ..., 8 planes per WG in flight"}
for OS in (65025, 65056):
    for which in (0, 2):
        for spw in (1, 2):
            fn = lambda: L.probe(which, 1, src.data_ptr(), dst.data_ptr(), B, P, M, HW, spw, sink.data_ptr(), st, OS)
            fn(); torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(5):
                fn()
            b.record(); torch.cuda.synchronize()
            t = a.elapsed_time(b) / 5 * 1e-3
            nb = B * (P + M) * (HW // 1024) * 4096
            print(f"read+write, output plane stride {OS}: {names[which]:62s} steps/wg {spw}: {nb / t / 1e9:7.1f} GB/s", flush=True)
