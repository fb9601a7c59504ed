#!/usr/bin/env python3
"""mq_attn_prefill_fp8kv: time against the number of heads / tokens (where does the time of a small prefill go?)"""
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

for T in (32, 128, 256, 768, 1536):
    for H, HKV in ((4, 4), (8, 4), (28, 4), (56, 8), (112, 16)):
        for causal in (True, False):
            D = 128
            qkv = torch.randn(T, (H + 2 * HKV) * D, device=dev).half()
            q = qkv[:, :H * D].view(T, H, D); kv = qkv[:, H * D:].view(T, 2 * HKV, D)
            sc = ops.kv_scale_from_absmax(kv); cache = ops.kv_quant_fp8(kv, sc)
            out = torch.empty(T, H * D, device=dev, dtype=torch.float16)
            us = timed(lambda: ops.attn_prefill_fp8kv(q, cache, sc, causal=causal, out=out))
            wgs = ((T + 31) // 32) * H
            print(f"T={T:5d} H={H:3d}/{HKV:2d} causal={int(causal)} workgroups={wgs:5d}: {us:7.2f} us")
