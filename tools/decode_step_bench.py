#!/usr/bin/env python3
"""One decoder layer's hot path at generation shapes (M rows): the four quantize(+Hadamard) launches and the four W4A8 GEMMs of a Qwen2-VL-7B
layer (q|k|v, o_proj, gate|up, down_proj with its online Hadamard over 19968 padded channels), replayed from one hipGraph over 8 copies of
the weights (cold weights).  usage (GPU box): python3 tools/decode_step_bench.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fake_quant import hadamard_utils as hu  # noqa: E402
from mquant_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
G = 8


def graph_time(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        with torch.cuda.graph(graph, stream=st, capture_error_mode="thread_local"):
            fn()
        for _ in range(3):
            graph.replay()
        st.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(reps):
            graph.replay()
        e1.record(st)
        st.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    ops.splitk_workspace(dev, 512 << 20)
    shapes = {"qkv": (4608, 3584), "o": (3584, 3584), "gate_up": (37888, 3584), "down": (3584, 19968)}
    imgs = {k: [ops.prepack(torch.randint(-8, 8, (n, kk), dtype=torch.int8, device=dev), 4) for _ in range(G)] for k, (n, kk) in shapes.items()}
    s_w = {k: torch.full((n,), 0.01, device=dev) for k, (n, _) in shapes.items()}
    _, K = hu.get_hadK(19968)
    bits = hu.had_sign_bits(K, dev)
    for M in (1, 4, 16, 64):
        x = torch.randn((M, 3584), device=dev, dtype=torch.float16)
        xd = torch.randn((M, 18944), device=dev, dtype=torch.float16)
        a3584 = ops.TiledAct.empty(M, 3584, dev)
        a_down = ops.TiledAct.empty(M, 19968, dev)
        outs = {k: torch.empty((M, n), dtype=torch.float16, device=dev) for k, (n, _) in shapes.items()}

        def quant_only():
            for g in range(G):
                for _ in range(3):
                    ops.quantize_act_i8(x, 0.05, out=a3584)
                ops.hadamard_quant_i8(xd, 19968, K, bits, 0.05, out=a_down)

        def gemm_only():
            for g in range(G):
                for k in ("qkv", "o", "gate_up"):
                    ops.gemm_w4a8(a3584, imgs[k][g], 4, shapes[k][0], 0.02, s_w[k], out=outs[k])
                ops.gemm_w4a8(a_down, imgs["down"][g], 4, 3584, 0.02, s_w["down"], out=outs["down"])

        def layer():
            for g in range(G):
                for k in ("qkv", "o", "gate_up"):
                    ops.quantize_act_i8(x, 0.05, out=a3584)
                    ops.gemm_w4a8(a3584, imgs[k][g], 4, shapes[k][0], 0.02, s_w[k], out=outs[k])
                ops.hadamard_quant_i8(xd, 19968, K, bits, 0.05, out=a_down)
                ops.gemm_w4a8(a_down, imgs["down"][g], 4, 3584, 0.02, s_w["down"], out=outs["down"])
        tq, tg, tl = graph_time(quant_only) / G, graph_time(gemm_only) / G, graph_time(layer) / G
        print(f"M={M:3d}: quantizers + Hadamard {tq:6.1f} us | GEMMs {tg:6.1f} us | layer (8 launches + reduces) {tl:6.1f} us "
              f"-> {28 * tl / 1e3:5.2f} ms per token-step of 28 layers (weights alone at 6.3 TB/s: 0.53 ms)", flush=True)


if __name__ == "__main__":
    main()
