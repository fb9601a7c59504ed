#!/usr/bin/env python3
"""Does a W4A8 GEMM launch care where its weights come from?  The same shape over ONE weight image back to back (the image stays in the
256 MB Infinity Cache) against 12 rotating images (every launch reads its weights from HBM, as in the step) -- the upper bound of what
prefetching a layer's weights during the previous launch could give."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mquant_amd import ops
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")

def timed(fn, n, iters=20):
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

for name, M, N, K in (("down_proj", 768, 3584, 19968), ("gate|up", 768, 37888, 3584), ("q|k|v", 768, 4608, 3584), ("o_proj", 768, 3584, 3584),
                      ("vis.fc2", 1024, 1280, 5120), ("vis.proj", 1024, 1280, 1280), ("vis.qkv", 1024, 3840, 1280), ("vis.fc1", 1024, 5120, 1280)):
    NW = 12
    a = [ops.TiledAct.from_rows(torch.randint(-128, 128, (M, K), dtype=torch.int8, device=dev)) for _ in range(2)]
    imgs = [ops.prepack(torch.randint(-8, 8, (N, K), dtype=torch.int8, device=dev), 4) for _ in range(NW)]
    s_w = torch.full((N,), 0.01, device=dev)
    out = torch.empty((M, N), dtype=torch.float16, device=dev)
    warm = timed(lambda i: ops.gemm_w4a8(a[i & 1], imgs[0], 4, N, 0.02, s_w, out=out), 24)
    cold = timed(lambda i: ops.gemm_w4a8(a[i & 1], imgs[i % NW], 4, N, 0.02, s_w, out=out), 24)
    warm2 = timed(lambda i: ops.gemm_w4a8(a[i & 1], imgs[0], 4, N, 0.02, s_w, out=out), 24)
    print(f"{name:10s} {M}x{N}x{K}: weights {N * K / 2 / 1e6:6.1f} MB | one image (cache-resident) {min(warm, warm2):7.2f} us | 12 rotating images (HBM) {cold:7.2f} us | +{cold - min(warm, warm2):.2f} us")
