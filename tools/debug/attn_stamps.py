#!/usr/bin/env python3
"""attention kernel built with phase stamps (a patched working copy: s_memtime at loop top / LDS stores issued / next loads issued / S + maximum
done / probabilities packed / PV issued; workgroup (0, 0) = the deepest query tile of head 0): per-wave, per-iteration phase lengths in shader cycles"""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
torch.set_grad_enabled(False)
from mquant_amd import ops, _lib
dev = "cuda:0"
lib = _lib.load()
for name, T, H, HKV, D, causal in (("7B", 768, 28, 4, 128, True), ("7B-4096", 4096, 28, 4, 128, True)):
    torch.manual_seed(T + H)
    qkv = torch.randn(T, (H + 2 * HKV) * D, device=dev).half()
    q = qkv[:, :H * D].view(T, H, D); k = qkv[:, H * D:(H + HKV) * D].view(T, HKV, D); v = qkv[:, (H + HKV) * D:].view(T, HKV, D)
    out = torch.empty(T, H * D, device=dev, dtype=torch.float16)
    for _ in range(3):
        ops.attn_prefill(q, k, v, causal=causal, out=out)
    torch.cuda.synchronize()
    buf = (C.c_longlong * 256)()
    assert lib.mq_attn_debug_stamps(buf) == 0
    print(name, os.path.basename(os.environ.get("MQUANT_HIP_LIB", "")))
    for w in range(4):
        g = buf[w * 64:(w + 1) * 64]
        t0 = g[0]
        n = g[4]
        print(f"  wave {w}: iterations {n}, entry->loop end {g[1] - t0}, ->merge math {g[2] - t0}, ->end {g[3] - t0}")
        prev_end = None
        for i in range(min(n, 7)):
            s = g[8 + i * 6: 8 + i * 6 + 6]
            gap = "" if prev_end is None else f" (top - previous end {s[0] - prev_end})"
            print(f"    it {i}: top at {s[0] - t0:6d} | stores {s[1] - s[0]:5d} | loads {s[2] - s[1]:5d} | S+max {s[3] - s[2]:5d} | rescale+exp {s[4] - s[3]:5d} | PV {s[5] - s[4]:5d} | total {s[5] - s[0]:5d}{gap}")
            prev_end = s[5]
