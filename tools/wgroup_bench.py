#!/usr/bin/env python3
"""Group-wise scales (--w_groupsize / --a_groupsize): what the grouped GEMMs (mq_gemm_w4a8_wgroupscale, mq_gemm_w4a8_groupscale) cost next to
the per-channel GEMM of the same shape.  --round1 forces the round-1 128 x 128 kernel (tile 26) instead of the fold inside the
wave-specialised tiles.  usage (GPU box): python3 tools/wgroup_bench.py [--round1] [--modes w,x,wx]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mquant_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--round1", action="store_true")
ap.add_argument("--modes", default="w,x,wx")
ap.add_argument("--tile", type=int, default=-1, help="force a tile id of the fold (45 / 47 / 48)")
ap.add_argument("--shapes", default="")
args = ap.parse_args()
dev = torch.device("cuda:0")


def bench(fn, iters=50):
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


for name, (M, N, K) in {"llm.qkv": (768, 4608, 3584), "llm.o": (768, 3584, 3584), "llm.gate_up": (768, 37888, 3584), "llm.down": (768, 3584, 19968),
                        "72b.gate_up": (768, 59136, 8192), "vit.fc1": (1024, 5120, 1280)}.items():
    if args.shapes and name not in args.shapes.split(","):
        continue
    a = ops.TiledAct.from_rows(torch.randint(-128, 128, (M, K), dtype=torch.int8, device=dev))
    img = ops.prepack(torch.randint(-8, 8, (N, K), dtype=torch.int8, device=dev), 4)
    s_w = torch.full((N,), 0.01, device=dev)
    out = torch.empty((M, N), dtype=torch.float16, device=dev)
    t0 = bench(lambda: ops.gemm_w4a8(a, img, 4, N, 0.02, s_w, out=out))
    print(f"{name:12s} {M} x {N} x {K}: per-channel scales {t0:8.1f} us", flush=True)
    for mode in args.modes.split(","):
        line = f"    {dict(w='weight groups', x='activation groups', wx='both')[mode]:18s}"
        for g in (64, 128, 256):
            if K % g:
                continue
            s_wg = torch.full((K // g, N), 0.01, device=dev)
            s_xg = torch.full((M, K // g), 0.02, device=dev)
            if mode == "w":
                fn = lambda: ops.gemm_w4a8_wgroupscale(a, img, 4, N, s_wg, g, s_x0=0.02, out=out)        # noqa: E731
            elif mode == "wx":
                fn = lambda: ops.gemm_w4a8_wgroupscale(a, img, 4, N, s_wg, g, s_x_groups=s_xg, out=out)   # noqa: E731
            else:
                fn = lambda: ops.gemm_w4a8_groupscale(a, img, 4, N, s_xg, g, s_w, out=out)               # noqa: E731
            if args.round1 or args.tile >= 0:
                ops.gemm_debug_force(26 if args.round1 else args.tile, 0)
            t = bench(fn)
            ops.gemm_debug_force(-1, 0)
            line += f" | g = {g}: {t:8.1f} us ({t / t0:4.2f} x)"
        print(line, flush=True)
