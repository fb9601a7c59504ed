#!/usr/bin/env python3
"""W4A8 GEMM at decode shapes (M = 1 ... 64 rows, the Qwen2-VL-7B decoder's Linears): time per launch against the time HBM needs for the
packed weights alone (weights stream once: N * K / 2 bytes).  Cold weights: G copies per hipGraph.  usage (GPU box): python3 tools/decode_gemm_bench.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mquant_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
SHAPES = {"llm.qkv": (4608, 3584), "llm.o": (3584, 3584), "llm.gate_up": (37888, 3584), "llm.down": (3584, 19968)}


def graph_time(fn_of_g, G, reps=20):
    for g in range(G):
        fn_of_g(g)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        with torch.cuda.graph(graph, stream=st, capture_error_mode="thread_local"):
            for g in range(G):
                fn_of_g(g)
        for _ in range(3):
            graph.replay()
        st.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(reps):
            graph.replay()
        e1.record(st)
        st.synchronize()
    return e0.elapsed_time(e1) / (reps * G) * 1e3


def main():
    ops.splitk_workspace(dev, 512 << 20)
    for name, (N, K) in SHAPES.items():
        G = 8
        q = torch.randint(-8, 8, (N, K), dtype=torch.int8, device=dev)
        imgs = [ops.prepack(q, 4) for _ in range(G)]
        s_w = torch.full((N,), 0.01, device=dev)
        hbm_us = N * K / 2 / 6.3e12 * 1e6
        line = f"{name:12s} N={N} K={K}: weights alone at 6.3 TB/s {hbm_us:6.1f} us |"
        for M in (1, 8, 16, 32, 64):
            a = ops.TiledAct.from_rows(torch.randint(-128, 128, (M, K), dtype=torch.int8, device=dev))
            out = torch.empty((M, N), dtype=torch.float16, device=dev)
            t = graph_time(lambda g: ops.gemm_w4a8(a, imgs[g], 4, N, 0.02, s_w, out=out), G)
            line += f" M={M}: {t:6.1f} us ({t / hbm_us:4.1f} x)"
        print(line, flush=True)


if __name__ == "__main__":
    main()
