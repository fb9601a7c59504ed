#!/usr/bin/env python3
"""Hadamard(+quant) micro-benchmark on the two online-rotation geometries of Qwen2-VL-7B."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fake_quant import hadamard_utils as hu  # noqa: E402
from mquant_amd import ops  # noqa: E402


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


from mquant_amd._lib import call
dev = torch.device("cuda:0")
threads = int(os.environ.get("HAD_THREADS", "0"))
call("mq_hadamard_debug_threads", threads)
fast = int(os.environ.get("HAD_FAST", "0"))
tiled = bool(int(os.environ.get("HAD_TILED", "1")))
print("threads per row:", threads, "fast mode:", fast, "tiled out:", tiled)
rows = [int(v) for v in os.environ.get("HAD_ROWS", "0").split(",")]
shapes = [("vis.fc2", 1024, 5120, 5120), ("llm.down", 768, 18944, 19968),
          ("qwenvl.c_proj", 768, 11008, 11008), ("internvl.w2", 768, 14336, 14336), ("72b.down", 768, 29568, 30720),
          ("pow2.8192", 1024, 8192, 8192)]
only = os.environ.get("HAD_SHAPES", "")
if only:
    shapes = [sh for sh in shapes if sh[0] in only.split(",")]
if rows != [0]:
    shapes = [(nm, r, a, b) for (nm, _, a, b) in shapes[:2] for r in rows]
for name, M, n_in, n in shapes:
    _, K = hu.get_hadK(n)
    bits = hu.had_sign_bits(K, dev) if K > 1 else None
    for dt in ((torch.float16,) if only else (torch.float16, torch.float32)):
        x = torch.randn((M, n_in), device=dev, dtype=torch.float32).to(dt)
        out = ops.TiledAct.empty(M, (n + 127) // 128 * 128, dev) if tiled else torch.empty((M, (n + 127) // 128 * 128), dtype=torch.int8, device=dev)
        us = bench(lambda: ops.hadamard_quant_i8(x, n, K, bits, 0.05, out=out, fast=bool(fast)))
        byts = M * n_in * x.element_size() + M * n
        print(f"{name:14s} {str(dt):14s} M={M} n={n} K={K}: {us:8.1f} us  {byts / us / 1e3:7.1f} GB/s")
