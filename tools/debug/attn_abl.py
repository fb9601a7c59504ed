#!/usr/bin/env python3
"""attention kernel: time of the three prefill shapes with the library named by MQUANT_HIP_LIB (ablation builds: -DMQ_ATTN_ABL=n)"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
torch.set_grad_enabled(False)
from mquant_amd import ops
dev = "cuda:0"

def timed(fn, iters=30):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10):
            fn()
    g.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 100.0)
    ts.sort()
    return ts[len(ts) // 2]

out_line = [os.path.basename(os.environ.get("MQUANT_HIP_LIB", "main"))]
for name, T, H, HKV, D, causal in (("7B", 768, 28, 4, 128, True), ("vit", 1024, 16, 16, 80, False), ("72B", 768, 64, 8, 128, True), ("7B-4096", 4096, 28, 4, 128, True)):
    torch.manual_seed(T + H)
    qkv = torch.randn(T, (H + 2 * HKV) * D, device=dev).half()
    q = qkv[:, :H * D].view(T, H, D); k = qkv[:, H * D:(H + HKV) * D].view(T, HKV, D); v = qkv[:, (H + HKV) * D:].view(T, HKV, D)
    out = torch.empty(T, H * D, device=dev, dtype=torch.float16)
    t = timed(lambda: ops.attn_prefill(q, k, v, causal=causal, out=out))
    out_line.append(f"{name} {t:7.2f} [{int(out.view(torch.int16).long().sum()) & 0xffffffff:08x}]")
    if D == 128 and T == 768:
        kv = qkv[:, H * D:].view(T, 2 * HKV, D); sc = ops.kv_scale_from_absmax(kv); cache = ops.kv_quant_fp8(kv, sc)
        t8 = timed(lambda: ops.attn_prefill_fp8kv(q, cache, sc, causal=causal, out=out))
        out_line.append(f"{name}-e4m3 {t8:7.2f} [{int(out.view(torch.int16).long().sum()) & 0xffffffff:08x}]")
print(" | ".join(out_line))
