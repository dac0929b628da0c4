#!/usr/bin/env python3
"""tools/clock_probe.hip driver: shader clock seen by a one-wave kernel (alone / with the chip kept busy on another
stream) and the cycles per dependent add / independent add / DPP add / LDS round trip / readlane."""
import ctypes, os, subprocess, torch
HERE = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(HERE, "libclock_probe.so")
if not os.path.exists(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-shared", "-fPIC",
                           os.path.join(HERE, "clock_probe.hip"), "-o", so])
L = ctypes.CDLL(so)
L.probe.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
L.busy.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
dev = "cuda:0"
out = torch.zeros(16, dtype=torch.int64, device=dev)
sink = torch.zeros(1024, device=dev)
side = torch.cuda.Stream()
st = torch.cuda.current_stream().cuda_stream
def run(n, label):
    L.probe(out.data_ptr(), sink.data_ptr(), n, st); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); L.probe(out.data_ptr(), sink.data_ptr(), n, st); b.record(); torch.cuda.synchronize()
    o = out.tolist()
    mhz = o[1] / o[0] * 100.0
    print(f"{label:28s} n={n:6d} kernel {a.elapsed_time(b) * 1e3:9.1f} us | shader clock {mhz:7.0f} MHz | cycles per: dependent add "
          f"{o[2] / (16 * n):5.2f}  independent add {o[3] / (16 * n):5.2f}  dependent DPP add {o[4] / (16 * n):5.2f}  "
          f"LDS write->read->add {o[5] / (4 * n):6.1f}  readlane->add {o[6] / (8 * n):5.2f}", flush=True)
for n in (50, 500, 5000, 50000):
    run(n, "alone")
for wgs in (256, 2048):
    with torch.cuda.stream(side):
        L.busy(sink.data_ptr(), 400000, wgs, side.cuda_stream)
    for n in (50, 500, 5000):
        run(n, f"beside {wgs} busy workgroups")
    torch.cuda.synchronize()
