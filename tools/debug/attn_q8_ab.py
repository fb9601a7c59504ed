#!/usr/bin/env python3
"""attention with the fused int8 store (the whole prefill's form: mq_attn_prefill_quant_i8, token-type mask) on the two prefill shapes, us per launch"""
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

line = [os.path.basename(os.environ.get("MQUANT_HIP_LIB", "main"))]
for name, T, H, HKV, D, causal in (("7B", 768, 28, 4, 128, True), ("vit", 1024, 16, 16, 80, False)):
    torch.manual_seed(T + H)
    qkv = torch.randn(T, (H + 2 * HKV) * D, device=dev).half()
    q = qkv[:, :H * D].view(T, H, D); k = qkv[:, H * D:(H + HKV) * D].view(T, HKV, D); v = qkv[:, (H + HKV) * D:].view(T, HKV, D)
    sel = (torch.arange(T, device=dev) % 3 == 0).to(torch.uint8)
    out = ops.attn_prefill_quant_i8(q, 0.01, 0.02, k=k, v=v, causal=causal, tiled=True, row_sel=sel)
    a = timed(lambda: ops.attn_prefill_quant_i8(q, 0.01, 0.02, k=k, v=v, causal=causal, tiled=True, row_sel=sel, out=out))
    b = timed(lambda: ops.attn_prefill_quant_i8(q, 0.01, 0.02, k=k, v=v, causal=causal, tiled=True, out=out))
    c = timed(lambda: ops.attn_prefill(q, k, v, causal=causal))
    line.append(f"{name}: int8 store with mask {a:.2f} | without {b:.2f} | 16-bit store {c:.2f} [{int(out.data.view(torch.int8).long().sum()) & 0xffffffff:08x}]" if hasattr(out, "data") else f"{name}: {a:.2f} {b:.2f} {c:.2f}")
print(" | ".join(line))
