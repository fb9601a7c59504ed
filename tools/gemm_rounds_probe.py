#!/usr/bin/env python3
"""Is the 256 x 256 gate|up GEMM bound by ROUNDS of tiles (1.73 rounds on 256 CUs) or by chip THROUGHPUT?
Times M = 768, K = 3584 with N chosen for 255 tiles (one round), 444 (the model shape) and 510 (two full rounds)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mquant_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def bench(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


M, K = 768, 3584
a = ops.TiledAct.from_rows(torch.randint(-128, 128, (M, K), dtype=torch.int8, device=dev))
print("n-blocks tiles    N     us    us/tile-round   TOP/s")
for nb in (43, 85, 128, 148, 170, 213, 256):
    N = nb * 256
    q = torch.randint(-8, 8, (N, K), dtype=torch.int8, device=dev)
    copies = [ops.prepack(q, 4) for _ in range(max(2, int(700e6 // (N * K // 2))))]     # cold weights: > 600 MB rotated
    s_w = torch.full((N,), 0.01, device=dev)
    out = torch.empty((M, N), dtype=torch.float16, device=dev)
    st = {"i": 0}

    def call():
        st["i"] = (st["i"] + 1) % len(copies)
        ops.gemm_w4a8(a, copies[st["i"]], 4, N, 0.02, s_w, out=out)
    us = bench(call, iters=max(30, 2 * len(copies)))
    tiles = 3 * nb
    print(f"{nb:7d} {tiles:6d} {N:6d} {us:7.1f} {us / -(-tiles // 256):10.1f}   {2.0 * M * N * K / us / 1e6:8.0f}")
    del copies, q
