#!/usr/bin/env python3
"""What separates config 5's template-lane count kernel (0.745 of peak) from a bare read loop: the kernel's loop skeleton
(tools/hbm_probe.hip: tl_skeleton_kernel) with the counting work, the LDS atomic, the plane alignment and the occupancy
cap switched on one at a time.  Prints TB/s of plane bytes per combination."""
import ctypes, os, subprocess, torch
HERE = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(HERE, "libhbm_probe.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", os.path.join(HERE, "hbm_probe.hip"), "-o", so])
L = ctypes.CDLL(so)
vp, ci = ctypes.c_void_p, ctypes.c_int
L.probe_tl_skeleton.argtypes = [vp, ci, ci, ci, ci, ci, ci, vp, vp]
dev = "cuda:0"
import sys
PL, B = 220, int(sys.argv[1]) if len(sys.argv) > 1 else 256
src = torch.rand(B * PL * 65025 + 4096, device=dev).to(torch.float16)
sink = torch.zeros(1 << 20, device=dev, dtype=torch.int32)
st = torch.cuda.current_stream().cuda_stream
def timed(fn, reps=8):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3
QUICK = len(sys.argv) > 2
for hwh in ((65025,) if QUICK else (65024, 65025)):
    for lds in ((40960,) if QUICK else (0, 40960)):
        for mode in ((0, 3) if QUICK else (0, 1, 2, 3)):
            for wgs in (1024, 2048, 4096):
                t = timed(lambda: L.probe_tl_skeleton(src.data_ptr(), B, PL, hwh, wgs, mode, lds, sink.data_ptr(), st))
                bytes_ = B * PL * (hwh // 1024) * 2048
                print(f"plane {hwh} halves, {'4 waves/SIMD (40 KB LDS)' if lds else 'uncapped':24s}, counting={mode & 1}, lds_atomic={mode >> 1}, "
                      f"wgs {wgs}: {bytes_ / t / 1e12:5.2f} TB/s = {bytes_ / t / 8e12:.3f}", flush=True)
