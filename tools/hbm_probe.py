#!/usr/bin/env python3
"""Measured HBM streaming ceilings of this MI355X (read-only and copy) -- the yardstick for the cost / mix kernels.
Builds tools/libhbm_probe.so with hipcc if missing.  Prints GB/s for several grid sizes / unroll factors."""
import ctypes
import os
import subprocess
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(HERE, "libhbm_probe.so")
if not os.path.exists(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC",
                           os.path.join(HERE, "hbm_probe.hip"), "-o", so])
L = ctypes.CDLL(so)
vp, i64, ci = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
L.probe_read.argtypes = [vp, i64, ci, ci, vp, vp]
L.probe_copy.argtypes = [vp, vp, i64, ci, ci, vp]
L.probe_read_plain.argtypes = [vp, i64, ci, ci, vp, vp]
dev = "cuda:0"
GB = 8
n = GB * (1 << 30)
src = torch.empty(n // 4, device=dev).normal_()
dst = torch.empty_like(src)
sink = torch.zeros(1 << 20, device=dev)
st = torch.cuda.current_stream().cuda_stream


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3


for unroll in (1, 4, 8):
    for wgs in (2048, 8192, 32768, 131072):
        t = timed(lambda: L.probe_read(src.data_ptr(), n, wgs, unroll, sink.data_ptr(), st))
        print(f"read  unroll {unroll} wgs {wgs:7d}: {n / t / 1e9:7.1f} GB/s")
L.probe_read_planes.argtypes = [vp, ci, ci, ci, ci, ci, vp, vp]
HW, PL = 65025, 60
BF = n // (4 * HW * PL)
for unroll in (1, 2, 4):
    for wgs in (1024, 2048, 4096, 8192):
        t = timed(lambda: L.probe_read_planes(src.data_ptr(), BF, PL, HW, wgs, unroll, sink.data_ptr(), st))
        print(f"read planes (cost-kernel pattern) unroll {unroll} wgs {wgs:6d}: {BF * PL * (HW // 1024) * 4096 / t / 1e9:7.1f} GB/s")
L.probe_read_planes_run.argtypes = [vp, ci, ci, ci, ci, ci, vp, vp]
L.probe_read_planes_split.argtypes = [vp, ci, ci, ci, vp, vp]
L.probe_read_planes_aligned.argtypes = [vp, ci, ci, ci, ci, ci, vp, vp]
for rep in range(2):
    for al in (16, 64, 128):
        t = timed(lambda: L.probe_read_planes_aligned(src.data_ptr(), BF, PL, HW, 8192, al, sink.data_ptr(), st))
        print(f"read planes, runs aligned down to {al:3d} B (8192 wgs)     : {BF * PL * (HW // 1024) * 4096 / t / 1e9:7.1f} GB/s")
for rep in range(2):
    t = timed(lambda: L.probe_read_planes_split(src.data_ptr(), BF, PL, HW, sink.data_ptr(), st))
    print(f"read planes split (1 chunk / WG, waves split planes): {BF * PL * (HW // 1024) * 4096 / t / 1e9:7.1f} GB/s")
    t = timed(lambda: L.probe_read_planes(src.data_ptr(), BF, PL, HW, 8192, 2, sink.data_ptr(), st))
    print(f"read planes (cost-kernel pattern, 8192 wgs)        : {BF * PL * (HW // 1024) * 4096 / t / 1e9:7.1f} GB/s")
for HWx in (65025, 65536):
    BFx = n // (4 * HWx * PL)
    for run in (1, 2, 4):
        for wgs in (2048, 8192):
            t = timed(lambda: L.probe_read_planes_run(src.data_ptr(), BFx, PL, HWx, wgs, run, sink.data_ptr(), st))
            print(f"read planes HW {HWx} run {run * 4:2d} KiB wgs {wgs:6d}: {BFx * PL * (HWx // (1024 * run)) * 4096 * run / t / 1e9:7.1f} GB/s")
for off in (0, 4, 8):                                     # byte misalignment of the stream (planes of 65025 floats)
    for unroll in (1, 4):
        for wgs in (8192, 131072):
            t = timed(lambda: L.probe_read_plain(src.data_ptr() + off, n - (1 << 20), wgs, unroll, sink.data_ptr(), st))
            print(f"read plain(cached) +{off}B unroll {unroll} wgs {wgs:7d}: {(n - (1 << 20)) / t / 1e9:7.1f} GB/s")
for unroll in (1, 4):
    for wgs in (8192, 32768, 131072):
        t = timed(lambda: L.probe_copy(src.data_ptr(), dst.data_ptr(), n, wgs, unroll, st))
        print(f"copy  unroll {unroll} wgs {wgs:7d}: {2 * n / t / 1e9:7.1f} GB/s (read + write)")
t = timed(lambda: src.sum())
print(f"torch.sum : {n / t / 1e9:7.1f} GB/s")
t = timed(lambda: dst.copy_(src))
print(f"torch copy: {2 * n / t / 1e9:7.1f} GB/s (read + write)")
