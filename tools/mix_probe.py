import sys
sys.path.insert(0, "/root/repo")
import torch
from dmm_net_amd import ops
dev = "cuda:0"
B, N, M, H, W = 1024, 50, 10, 255, 255
pm = torch.rand((B, N, H, W), device=dev)
def timeit(fn, iters=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
Pp = N
for name, sel in (("plane=m", lambda b, m: m), ("plane=5m", lambda b, m: 5 * m), ("random", None)):
    Rb = torch.zeros((B, M, Pp), device=dev)
    if sel is None:
        idx = torch.randint(0, N, (B, M), device=dev)
    else:
        idx = torch.tensor([[sel(0, m) for m in range(M)]] * B, device=dev)
    Rb.scatter_(2, idx[:, :, None], 0.7)
    t = timeit(lambda: ops.mask_mix(Rb, pm))
    print(f"{name:10s}: {t:8.1f} us  {2 * B * M * H * W * 4 / t / 1e3:7.0f} GB/s")
# aligned planes: HW = 65536
pm2 = torch.rand((B, N, 256, 256), device=dev)
Rb = torch.zeros((B, M, Pp), device=dev); Rb.scatter_(2, torch.randint(0, N, (B, M), device=dev)[:, :, None], 0.7)
t = timeit(lambda: ops.mask_mix(Rb, pm2))
print(f"random 256x256: {t:8.1f} us  {2 * B * M * 65536 * 4 / t / 1e3:7.0f} GB/s")
src = pm[:, :M].contiguous(); dst = torch.empty_like(src)
t = timeit(lambda: dst.copy_(src)); print(f"torch copy {2*src.numel()*4/t/1e3:7.0f} GB/s")
