#!/usr/bin/env python3
"""Streaming ceiling for BASELINE config 5's geometry: 220 planes of 65025 fp16 pixels (130 KB) per frame, read in
2 KiB runs (one 1024-pixel chunk of a 16-bit plane per wave visit) vs 4 / 8 KiB runs.  Bare nt read loops, no compute."""
import ctypes, os, subprocess, torch
HERE = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(HERE, "libhbm_probe.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", os.path.join(HERE, "hbm_probe.hip"), "-o", so])
L = ctypes.CDLL(so)
vp, ci = ctypes.c_void_p, ctypes.c_int
L.probe_read_planes_half.argtypes = [vp, ci, ci, ci, ci, ci, vp, vp]
L.probe_read_planes_run.argtypes = [vp, ci, ci, ci, ci, ci, vp, vp]
dev = "cuda:0"
n = 8 * (1 << 30)
src = torch.empty(n // 4, device=dev).normal_()
sink = torch.zeros(1 << 20, device=dev)
st = torch.cuda.current_stream().cuda_stream
def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3
PL, HWf = 220, 32512          # floats per plane ~ 65025 halves
B = n // (4 * HWf * PL)
for wgs in (1024, 2048, 4096, 8192):
    for unr in (4, 8):
        t = timed(lambda: L.probe_read_planes_half(src.data_ptr(), B, PL, HWf, wgs, unr, sink.data_ptr(), st))
        print(f"220 planes, 2 KiB runs, {unr} planes in flight, wgs {wgs:5d}: {B * PL * (HWf // 512) * 2048 / t / 1e9:7.1f} GB/s")
    for run in (1, 2):
        t = timed(lambda: L.probe_read_planes_run(src.data_ptr(), B, PL, HWf, wgs, run, sink.data_ptr(), st))
        print(f"220 planes, {4 * run} KiB runs, wgs {wgs:5d}: {B * PL * (HWf // (1024 * run)) * 4096 * run / t / 1e9:7.1f} GB/s")
