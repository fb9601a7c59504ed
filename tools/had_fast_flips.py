#!/usr/bin/env python3
"""Fast (half-precision matrix core) vs exact (sequential fp32 chain) online Hadamard: int8 level flip rate,
largest difference of the rotated activations, and kernel time, on the online-rotation geometries of the
BASELINE configurations at full size.  Output goes to profiles/r3_hadamard_fast_mode.txt."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fake_quant import hadamard_utils as hu  # noqa: E402
from mquant_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def bench(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


shapes = [("Qwen2-VL-7B vis.fc2", 1024, 5120, 5120), ("Qwen2-VL-7B llm.down_proj", 768, 18944, 19968),
          ("Qwen-VL-7B mlp.c_proj", 768, 11008, 11008), ("InternVL2-8B w2", 768, 14336, 14336),
          ("Qwen2-VL-72B down_proj", 768, 29568, 30720)]
print("shape | dtype | exact us | fast us | int8 levels that differ | max |level diff| | rotated activations: max abs diff / max abs")
for name, M, n_in, n in shapes:
    _, K = hu.get_hadK(n)
    bits = hu.had_sign_bits(K, dev)
    for dt in (torch.float16, torch.bfloat16):
        g = torch.Generator(device=dev).manual_seed(n + (1 if dt == torch.bfloat16 else 0))
        x = torch.randn((M, n_in), device=dev, generator=g, dtype=torch.float32)
        idx = torch.randperm(n_in, device=dev, generator=g)[: max(1, n_in // 1000)]
        x[:, idx] *= 20.0
        x = x.to(dt)
        res = {}
        for fast in (0, 1):
            y = ops.hadamard(x, n, K, bits, fast=bool(fast))
            s = float(y.float().abs().max()) / 127.0
            if fast == 0:
                s_exact = s
            q, _ = ops.hadamard_quant_i8(x, n, K, bits, s_exact, fast=bool(fast))
            out = ops.TiledAct.empty(M, (n + 127) // 128 * 128, dev)
            us = bench(lambda: ops.hadamard_quant_i8(x, n, K, bits, s_exact, out=out, fast=bool(fast)))
            res[fast] = (y.float(), q.to_rows()[:, :n].to(torch.int16) if hasattr(q, "to_rows") else q[:, :n].to(torch.int16), us)
        (y0, q0, t0), (y1, q1, t1) = res[0], res[1]
        flips = int((q0 != q1).sum())
        print(f"{name} (K = {K} x {n // K}) | {str(dt).replace('torch.', '')} | {t0:7.1f} | {t1:7.1f} | {flips} of {q0.numel()} = {flips / q0.numel():.2e} | "
              f"{int((q0 - q1).abs().max())} | {float((y0 - y1).abs().max()):.3e} / {float(y0.abs().max()):.3e}")
