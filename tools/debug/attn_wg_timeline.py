#!/usr/bin/env python3
"""attention kernel built with per-workgroup s_memrealtime stamps (entry, loop end, end; 100 MHz): when do workgroups start / finish"""
import ctypes as C, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
torch.set_grad_enabled(False)
from mquant_amd import ops, _lib
dev = "cuda:0"
lib = _lib.load()
for name, T, H, HKV, D, causal in (("7B", 768, 28, 4, 128, True), ("vit", 1024, 16, 16, 80, False)):
    torch.manual_seed(T + H)
    qkv = torch.randn(T, (H + 2 * HKV) * D, device=dev).half()
    q = qkv[:, :H * D].view(T, H, D); k = qkv[:, H * D:(H + HKV) * D].view(T, HKV, D); v = qkv[:, (H + HKV) * D:].view(T, HKV, D)
    out = torch.empty(T, H * D, device=dev, dtype=torch.float16)
    for _ in range(5):
        ops.attn_prefill(q, k, v, causal=causal, out=out)
    torch.cuda.synchronize()
    buf = (C.c_longlong * (3 * 8192))()
    assert lib.mq_attn_debug_wg(buf) == 0
    a = np.frombuffer(buf, dtype=np.int64).reshape(3, 8192)
    n = H * ((T + 31) // 32)
    st, lp, en = a[0, :n].astype(np.float64), a[1, :n].astype(np.float64), a[2, :n].astype(np.float64)
    t0 = st.min()
    st, lp, en = (st - t0) / 100.0, (lp - t0) / 100.0, (en - t0) / 100.0     # us
    print(f"{name}: {n} workgroups; first start 0, last start {st.max():.2f} us, last end {en.max():.2f} us")
    order = np.argsort(st)
    print("  start-time deciles (us):", " ".join(f"{np.percentile(st, p):.2f}" for p in range(0, 101, 10)))
    print("  end-time deciles   (us):", " ".join(f"{np.percentile(en, p):.2f}" for p in range(0, 101, 10)))
    nt = (T + 31) // 32
    for row in range(0, nt, max(1, nt // 8)):
        sl = slice(row * H, (row + 1) * H)
        print(f"  grid row {row:3d} (query tile {nt - 1 - row:3d}): start {st[sl].min():6.2f}..{st[sl].max():6.2f}, loop end {lp[sl].min():6.2f}..{lp[sl].max():6.2f}, end {en[sl].min():6.2f}..{en[sl].max():6.2f}, duration mean {np.mean(en[sl] - st[sl]):.2f}")
