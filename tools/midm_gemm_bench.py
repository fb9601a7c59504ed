#!/usr/bin/env python3
"""W4A8 GEMM at mid-size row counts (short prompts: M = 65 ... 512) on the Qwen2-VL-7B decoder's Linears, cold weights (8 copies per hipGraph):
time per launch, plan (tile, splits) and MFMA-only time at the nominal int8 peak.  usage (GPU box): python3 tools/midm_gemm_bench.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mquant_amd import ops  # noqa: E402
from tools.decode_gemm_bench import graph_time  # noqa: E402

dev = torch.device("cuda:0")
SHAPES = {"llm.qkv": (4608, 3584), "llm.o": (3584, 3584), "llm.gate_up": (37888, 3584), "llm.down": (3584, 19968)}


def main():
    ops.splitk_workspace(dev, 512 << 20)
    G = 8
    for name, (N, K) in SHAPES.items():
        q = torch.randint(-8, 8, (N, K), dtype=torch.int8, device=dev)
        imgs = [ops.prepack(q, 4) for _ in range(G)]
        s_w = torch.full((N,), 0.01, device=dev)
        line = f"{name:12s} N={N} K={K}:"
        for M in (65, 96, 128, 192, 256, 384, 512, 768):
            a = ops.TiledAct.from_rows(torch.randint(-128, 128, (M, K), dtype=torch.int8, device=dev))
            out = torch.empty((M, N), dtype=torch.float16, device=dev)
            t = graph_time(lambda g: ops.gemm_w4a8(a, imgs[g], 4, N, 0.02, s_w, out=out), G)
            pl = torch.zeros(2, dtype=torch.int32)
            ops.call("mq_gemm_debug_plan", M, N, a.K_pad, 4, 1, 1, pl[0:].data_ptr(), pl[1:].data_ptr())
            line += f" M={M}: {t:6.1f} us (tile {pl[0].item()}x{pl[1].item()}, {2.0 * M * N * K / t / 5e9 * 100:4.1f} %)"
        print(line, flush=True)


if __name__ == "__main__":
    main()
