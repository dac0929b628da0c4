#!/usr/bin/env python3
"""tools/grid_barrier_probe.hip driver: us per device-wide barrier for 64 / 128 / 256 resident workgroups."""
import ctypes, os, subprocess, torch
HERE = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(HERE, "libgrid_barrier_probe.so")
if not os.path.exists(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC",
                           os.path.join(HERE, "grid_barrier_probe.hip"), "-o", so])
L = ctypes.CDLL(so)
L.probe.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
dev = "cuda:0"
cnt = torch.zeros(16, dtype=torch.int32, device=dev)
sink = torch.zeros(1024, device=dev)
st = torch.cuda.current_stream().cuda_stream
for wgs in (64, 128, 256, 512):
    res = []
    for n in (1, 101):
        L.probe(cnt.data_ptr(), wgs, n, sink.data_ptr(), st); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            L.probe(cnt.data_ptr(), wgs, n, sink.data_ptr(), st)
        b.record(); torch.cuda.synchronize()
        res.append(a.elapsed_time(b) / 5 * 1e3)
    print(f"{wgs:4d} workgroups: kernel with 1 barrier {res[0]:7.1f} us, with 101 barriers {res[1]:7.1f} us -> {(res[1] - res[0]) / 100:5.2f} us per barrier", flush=True)
