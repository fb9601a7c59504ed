#!/usr/bin/env python3
"""Would an int8 copy of the int4 weight levels (no nibble unpack in the k-loop, twice the weight bytes through L2 -> LDS -> registers)
be cheaper under the power limit?  DESIGN.md section 8 listed it as a candidate; this measures it: the hot shapes of the Qwen2-VL-7B
prefill with (a) the packed W4 image (the product), (b) the same levels as a W8 image, (c) the same levels x 16 as a W8 image (the
byte patterns the W4 unpack feeds the matrix cores).  Activations: int8 levels of Gaussian rows through the static quantizer (bench-like
operands).  G launches per hipGraph with G copies of the weights (cold weights, as in the prefill).
usage (GPU box): python3 tools/w8_image_ab.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mquant_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
SHAPES = {"llm.qkv": (768, 4608, 3584), "llm.o": (768, 3584, 3584), "llm.gate_up": (768, 37888, 3584), "llm.down": (768, 3584, 19968),
          "vit.fc1": (1024, 5120, 1280), "vit.qkv": (1024, 3840, 1280)}


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
    for name, (M, N, K) in SHAPES.items():
        G = 4 if N * K > (64 << 20) else 8
        x = torch.randn((M, K), device=dev)
        a_rows = torch.clamp(torch.round(x / (x.abs().max() / 127.0 * 0.5)), -127, 127).to(torch.int8)
        a = ops.TiledAct.from_rows(a_rows)
        w = torch.randn((N, K), device=dev)
        q = torch.clamp(torch.round(w / (w.abs().amax(dim=1, keepdim=True) / 7.0)), -8, 7).to(torch.int8)
        s_w = torch.full((N,), 0.01, device=dev)
        outs = [torch.empty((M, N), dtype=torch.float16, device=dev) for _ in range(2)]
        img4 = [ops.prepack(q, 4) for _ in range(G)]
        t4 = graph_time(lambda g: ops.gemm_w4a8(a, img4[g], 4, N, 0.02, s_w, out=outs[g & 1]), G)
        ref = outs[(G - 1) & 1].clone()
        del img4
        img8 = [ops.prepack(q, 8) for _ in range(G)]
        t8 = graph_time(lambda g: ops.gemm_w4a8(a, img8[g], 8, N, 0.02, s_w, out=outs[g & 1]), G)
        same = torch.equal(ref, outs[(G - 1) & 1])
        del img8
        img8h = [ops.prepack(q * 16, 8) for _ in range(G)]
        s_w16 = s_w / 16
        t8h = graph_time(lambda g: ops.gemm_w4a8(a, img8h[g], 8, N, 0.02, s_w16, out=outs[g & 1]), G)
        del img8h
        print(f"{name:12s} {M} x {N} x {K}: packed W4 image {t4:7.1f} us | levels as int8 {t8:7.1f} us ({t8 / t4:4.2f} x, "
              f"{'bit-identical' if same else 'DIFFERENT'}) | levels x 16 as int8 {t8h:7.1f} us ({t8h / t4:4.2f} x)", flush=True)


if __name__ == "__main__":
    main()
