#!/usr/bin/env python3
"""Would a GEMM launch be faster if its weights had been pulled into the Infinity Cache one launch earlier?  Sequence per step i:
touch(weights[i + 1]) (a plain read of the image, its own launch), GEMM(weights[i]).  The GEMM's share = pair time - touch time, against
the cold (no touch, rotating images) and warm (one image) launch times."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mquant_amd import ops
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")

def timed(fn, n, iters=15):
    fn(0); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(n):
            fn(i)
    g.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / n)
    ts.sort()
    return ts[len(ts) // 2]

for name, M, N, K in (("down_proj", 768, 3584, 19968), ("gate|up", 768, 37888, 3584)):
    NW = 12
    a = [ops.TiledAct.from_rows(torch.randint(-128, 128, (M, K), dtype=torch.int8, device=dev)) for _ in range(2)]
    imgs = [ops.prepack(torch.randint(-8, 8, (N, K), dtype=torch.int8, device=dev), 4) for _ in range(NW)]
    flat = [im.view(torch.int32).reshape(-1) if hasattr(im, "view") else None for im in imgs]
    s_w = torch.full((N,), 0.01, device=dev)
    out = torch.empty((M, N), dtype=torch.float16, device=dev)
    sink = torch.zeros((), dtype=torch.int32, device=dev)
    def touch(i):
        torch.amax(flat[i % NW], dim=0, out=sink)
    def gemm(i, j):
        ops.gemm_w4a8(a[i & 1], imgs[j % NW], 4, N, 0.02, s_w, out=out)
    warm = timed(lambda i: gemm(i, 0), 24)
    cold = timed(lambda i: gemm(i, i), 24)
    t_touch = timed(lambda i: touch(i), 24)
    def pair(i):
        touch(i + 1); gemm(i, i)
    t_pair = timed(pair, 24)
    print(f"{name:10s}: warm {warm:.2f} us | cold {cold:.2f} us | touch alone {t_touch:.2f} us | touch(next) + GEMM {t_pair:.2f} us -> GEMM's share {t_pair - t_touch:.2f} us")
