#!/usr/bin/env python3
"""GEMM micro-benchmark over the distinct Qwen2-VL-7B shapes: TOP/s per tile config.
Usage (GPU box): python tools/gemm_bench.py [--configs 0:1,1:1,2:1,1:6] [--shapes llm|all]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mquant_amd import ops  # noqa: E402

SHAPES = {
    "k128": (768, 3584, 128), "k512": (768, 3584, 512), "k1024": (768, 3584, 1024),
    "vis.qkv": (1024, 3840, 1280), "vis.proj": (1024, 1280, 1280), "vis.fc1": (1024, 5120, 1280),
    "vis.fc2": (1024, 1280, 5120), "llm.q/o": (768, 3584, 3584), "llm.kv": (768, 512, 3584),
    "llm.qkv*": (768, 4608, 3584), "llm.gate": (768, 18944, 3584), "llm.gate_up*": (768, 37888, 3584),
    "llm.down": (768, 3584, 19968),
    # the other BASELINE configurations (workload.py): Qwen-VL-7B, InternVL2-8B (batch 1 and 4), Qwen2-VL-72B
    "qvl.in_proj": (1024, 4992, 1664), "qvl.out_proj": (1024, 1664, 1664), "qvl.c_fc": (1024, 8192, 1664),
    "qvl.vc_proj": (1024, 1664, 8192), "qvl.c_attn": (768, 12288, 4096), "qvl.c_proj": (768, 4096, 4096),
    "qvl.w1w2*": (768, 22016, 4096), "qvl.down": (768, 4096, 11008),
    "ivl.qkv": (1025, 3072, 1024), "ivl.proj": (1025, 1024, 1024), "ivl.fc1": (1025, 4096, 1024), "ivl.fc2": (1025, 1024, 4096),
    "ivl.wqkv": (768, 6144, 4096), "ivl.wo": (768, 4096, 4096), "ivl.w1w3*": (768, 28672, 4096), "ivl.w2": (768, 4096, 14336),
    "ivl4.wqkv": (3072, 6144, 4096), "ivl4.wo": (3072, 4096, 4096), "ivl4.w1w3*": (3072, 28672, 4096), "ivl4.w2": (3072, 4096, 14336),
    "72b.qkv*": (768, 10240, 8192), "72b.o": (768, 8192, 8192), "72b.gate_up*": (768, 59136, 8192), "72b.down": (768, 8192, 30720),
}


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
    return e0.elapsed_time(e1) / iters * 1e3  # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="-1:0,0:1,2:1,1:1")
    ap.add_argument("--only", default="")
    ap.add_argument("--bits", type=int, default=4)
    ap.add_argument("--tiled", action="store_true", help="activations in the tiled layout for every config (ids >= 40 always)")
    ap.add_argument("--cold", action="store_true",
                    help="rotate over enough copies of the weight image (> 600 MB) that every call streams it from HBM, as in the prefill")
    args = ap.parse_args()
    cfgs = [tuple(int(v) for v in c.split(":")) for c in args.configs.split(",")]
    dev = torch.device("cuda:0")
    ops.splitk_workspace(dev, 512 << 20)
    print(f"{'shape':14s} {'M':>5s} {'N':>6s} {'K':>6s} | " + " | ".join(f"t{t}s{s}: us   TOP/s" for t, s in cfgs))
    for name, (M, N, K) in SHAPES.items():
        if args.only and args.only not in name:
            continue
        a = torch.randint(-128, 128, (M, K), dtype=torch.int8, device=dev)
        lim = 1 << (args.bits - 1)
        q = torch.randint(-lim, lim, (N, K), dtype=torch.int8, device=dev)
        img = ops.prepack(q, args.bits)
        s_w = torch.full((N,), 0.01, device=dev)
        out = torch.empty((M, N), dtype=torch.float16, device=dev)
        ref = None
        cols = []
        a_t = ops.TiledAct.from_rows(a)
        for tile, splits in cfgs:
            ops.gemm_debug_force(-1, 0)
            if ref is None:
                ref = ops.gemm_w4a8_i32(a, img, args.bits, N)
                yref = ops.gemm_w4a8(a, img, args.bits, N, 0.02, s_w)
            ops.gemm_debug_force(tile, splits)
            aa = a_t if (args.tiled or tile >= 40) else a
            acc = ops.gemm_w4a8_i32(aa, img, args.bits, N)
            ok = bool(torch.equal(acc, ref)) and bool(torch.equal(ops.gemm_w4a8(aa, img, args.bits, N, 0.02, s_w), yref))
            if args.cold:
                copies = [img] + [img.clone() for _ in range(max(1, int(600e6 // max(img.numel(), 1))))]
                state = {"i": 0}

                def call_cold():
                    state["i"] = (state["i"] + 1) % len(copies)
                    ops.gemm_w4a8(aa, copies[state["i"]], args.bits, N, 0.02, s_w, out=out)
                us = bench(call_cold, iters=max(20, 2 * len(copies)))
                del copies
            else:
                us = bench(lambda: ops.gemm_w4a8(aa, img, args.bits, N, 0.02, s_w, out=out))
            cols.append(f"{us:8.1f} {2.0 * M * N * K / us / 1e6:7.0f}{'' if ok else ' MISMATCH'}")
        print(f"{name:14s} {M:5d} {N:6d} {K:6d} | " + " | ".join(cols))
    ops.gemm_debug_force(-1, 0)


if __name__ == "__main__":
    main()
