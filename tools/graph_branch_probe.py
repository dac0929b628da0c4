#!/usr/bin/env python3
"""Do two independent branches of ONE captured HIP graph run side by side on replay?  Two chains of small-grid kernels (32
workgroups each, ~40 us per kernel: a quarter of the GPU each), captured (a) on one stream, (b) forked onto a side stream and
joined; replay time of each form, and of the same two chains as two graphs replayed on two streams."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dev = torch.device("cuda", 0)
n = 32 * 256 * 64
a = [torch.rand(n, device=dev) for _ in range(2)]

def chain(x, k=20):
    for _ in range(k):
        x = torch.sin(x) * 1.0001 + torch.cos(x)        # small elementwise kernels: 8 k blocks... make them long via size
    return x

def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

s_cap = torch.cuda.Stream()
out = {}
# (a) serial in one graph
g1 = torch.cuda.CUDAGraph()
with torch.cuda.stream(s_cap):
    chain(a[0]); chain(a[1]); torch.cuda.synchronize()
    with torch.cuda.graph(g1, stream=s_cap):
        r0 = chain(a[0]); r1 = chain(a[1])
out["one_graph_serial_ms"] = timed(g1.replay)
# (b) forked branch inside one graph
g2 = torch.cuda.CUDAGraph()
side = torch.cuda.Stream()
with torch.cuda.stream(s_cap):
    with torch.cuda.graph(g2, stream=s_cap):
        side.wait_stream(s_cap)
        with torch.cuda.stream(side):
            q1 = chain(a[1])
        q0 = chain(a[0])
        s_cap.wait_stream(side)
out["one_graph_forked_ms"] = timed(g2.replay)
# (c) two graphs on two streams
ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
with torch.cuda.stream(s_cap):
    with torch.cuda.graph(ga, stream=s_cap):
        p0 = chain(a[0])
    with torch.cuda.graph(gb, stream=s_cap):
        p1 = chain(a[1])
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def two():
    with torch.cuda.stream(s1):
        ga.replay()
    with torch.cuda.stream(s2):
        gb.replay()
out["two_graphs_two_streams_ms"] = timed(two)
out["one_chain_alone_ms"] = timed(ga.replay)
print(json.dumps(out))
