#!/usr/bin/env python3
"""mq_gemv_f16 against torch.matmul (hipBLASLt) on the lm_head shapes: us per call from a hipGraph over rotating weight copies? no -- the
weights (1.09 GB) exceed every cache, one copy is enough"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
torch.set_grad_enabled(False)
from mquant_amd import ops
dev = "cuda:0"

def timed(fn, iters=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(5):
            fn()
    g.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 200.0)
    ts.sort()
    return ts[len(ts) // 2]

for name, M, N, K in (("qwen2vl_7b", 1, 152064, 3584), ("internvl2_8b x4", 4, 92553, 4096), ("qwen2vl_72b", 1, 152064, 8192)):
    x = torch.randn(M, K, device=dev).half(); w = (torch.randn(N, K, device=dev) * 0.02).half()
    out = torch.empty(M, N, device=dev, dtype=torch.float16)
    a = timed(lambda: ops.gemv_f16(x, w, out=out)); b = timed(lambda: torch.matmul(x, w.t(), out=out))
    a2 = timed(lambda: ops.gemv_f16(x, w, out=out)); b2 = timed(lambda: torch.matmul(x, w.t(), out=out))
    gb = N * K * 2 / 1e9
    print(f"{name}: M={M} N={N} K={K} ({gb:.2f} GB): mq_gemv_f16 {min(a, a2):.1f} us ({gb / min(a, a2) * 1e3:.2f} TB/s) | torch.matmul {min(b, b2):.1f} us ({gb / min(b, b2) * 1e3:.2f} TB/s)")
