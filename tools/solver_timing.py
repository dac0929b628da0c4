import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from dmm_net_amd import ops
dev="cuda:0"
def timeit(fn, iters=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)/iters*1e3
SHAPES = [(10,50),(5,50),(16,64),(20,200)] if not os.environ.get('SHAPES') else [tuple(int(v) for v in t.split('x')) for t in os.environ['SHAPES'].split(',')]
for (n,m) in SHAPES:
    C=-torch.rand((1024,n,m),device=dev)
    out=[]
    for (mi,pi) in [(0,0),(20,0),(20,1),(20,5),(20,10),(40,5)]:
        out.append(f"({mi},{pi}) {timeit(lambda: ops.relax_solve(C,mi,pi,0.1)):7.1f}us")
    print((n,m), "  ".join(out))
