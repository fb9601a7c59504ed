#!/usr/bin/env python3
"""attention kernel built with stamps around the merge (patched working copy): loop end -> partials written -> barrier passed -> factors ->
d-tile summed -> stored, shader cycles, workgroup (0, 0), both output forms"""
import ctypes as C, os, sys
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
    for form in ("16-bit out", "int8 tiled out"):
        for _ in range(3):
            if form == "16-bit out":
                ops.attn_prefill(q, k, v, causal=causal)
            else:
                ops.attn_prefill_quant_i8(q, 0.01, 0.02, k=k, v=v, causal=causal, tiled=True)
        torch.cuda.synchronize()
        buf = (C.c_longlong * 128)()
        assert lib.mq_attn_debug_fin(buf) == 0
        print(name, form)
        for w in range(4):
            g = buf[w * 16:(w + 1) * 16]
            print(f"  wave {w}: partials written {g[1] - g[0]:5d} | barrier {g[2] - g[1]:5d} | factors {g[3] - g[2]:5d} | d-tile summed {g[4] - g[3]:5d} | stored {g[5] - g[4]:5d} | total {g[5] - g[0]:5d}")
