#!/usr/bin/env python3
"""attention kernels: 4-wave (causal: shallow tiles paired) against 2-wave workgroups and 4-wave without the pairing (mq_attn_debug_waves 4 / 2 / 5; 0 = the
launcher's choice) on the prefill shapes"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
torch.set_grad_enabled(False)
from mquant_amd import ops
from mquant_amd._lib import call
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

for name, T, H, HKV, D, causal in (("7B decoder", 768, 28, 4, 128, True), ("72B decoder", 768, 64, 8, 128, True), ("vision tower", 1024, 16, 16, 80, False),
                                   ("7B, 512 tokens", 512, 28, 4, 128, True), ("7B, 1536 tokens", 1536, 28, 4, 128, True), ("7B, 4096 tokens", 4096, 28, 4, 128, True)):
    qkv = torch.randn(T, (H + 2 * HKV) * D, device=dev).half()
    q = qkv[:, :H * D].view(T, H, D); k = qkv[:, H * D:(H + HKV) * D].view(T, HKV, D); v = qkv[:, (H + HKV) * D:].view(T, HKV, D)
    out = torch.empty(T, H * D, device=dev, dtype=torch.float16)
    res = []
    for nw in (4, 2, 5, 0, 4, 2, 5, 0):
        call("mq_attn_debug_waves", nw)
        res.append(timed(lambda: ops.attn_prefill(q, k, v, causal=causal, out=out)))
    res = [min(res[i], res[i + 4]) for i in range(4)]
    line = f"{name:18s} T={T:5d} H={H:3d}/{HKV:2d} D={D:3d} workgroups={((T + 31) // 32) * H:5d}: 16-bit K/V  4 waves {res[0]:7.2f} us | 2 waves {res[1]:7.2f} us | 4 unpaired {res[2]:7.2f} | by shape {res[3]:7.2f} us"
    if D == 128:
        kv = qkv[:, H * D:].view(T, 2 * HKV, D); sc = ops.kv_scale_from_absmax(kv); cache = ops.kv_quant_fp8(kv, sc)
        r8 = []
        for nw in (4, 2, 5, 0, 4, 2, 5, 0):
            call("mq_attn_debug_waves", nw)
            r8.append(timed(lambda: ops.attn_prefill_fp8kv(q, cache, sc, causal=causal, out=out)))
        r8 = [min(r8[i], r8[i + 4]) for i in range(4)]
        line += f" || e4m3 cache  4 waves {r8[0]:7.2f} | 2 waves {r8[1]:7.2f} | 4 unpaired {r8[2]:7.2f} | by shape {r8[3]:7.2f}"
    print(line)
call("mq_attn_debug_waves", 0)
