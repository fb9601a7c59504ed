#!/usr/bin/env python3
"""Round 5 A/B: the wave-specialised tiles with V_MFMA_I32_32X32X32_I8 (ids 40-43) against their V_MFMA_I32_16X16X64_I8 twins
(ids 44-47), per model shape, alternating on one box, cold weights (rotating copies), two kinds of operand data (uniform random
bytes / the bench's quantized Gaussians): the question is energy per MAC under the package power limit.
usage (GPU box): python3 tools/r5_ws_mfma_ab.py [--rounds 3]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from mquant_amd import ops  # noqa: E402
from clock_recon import operands  # noqa: E402

SHAPES = {"gate_up": (768, 37888, 3584), "vit_proj": (1024, 1280, 1280), "vit_fc2": (1024, 1280, 5120), "vit_qkv": (1024, 3840, 1280), "vit_fc1": (1024, 5120, 1280),
          "llm_o": (768, 3584, 3584), "llm_qkv": (768, 4608, 3584), "llm_down": (768, 3584, 19968),
          "qvl_c_attn": (768, 12288, 4096), "qvl_down": (768, 4096, 11008), "72b_down": (768, 8192, 30720), "72b_o": (768, 8192, 8192),
          "72b_qkv": (768, 10240, 8192), "ivl_w2": (768, 4096, 14336), "ivl_wqkv": (768, 6144, 4096), "ivl4_w2": (3072, 4096, 14336)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--shapes", default=",".join(SHAPES))
    ap.add_argument("--tiles", default="", help="comma list of tile ids to compare on every shape (default: the plan's tile and its twin)")
    ap.add_argument("--kinds", default="bench-like,random")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    ops.splitk_workspace(dev)
    for name in args.shapes.split(","):
        M, N, K = SHAPES[name]
        for kind in args.kinds.split(","):
            a8, q = operands(kind, M, N, K, dev, torch)
            a = ops.TiledAct.from_rows(a8)
            n_copies = max(2, min(12, int(700e6 // (N * K // 2))))
            copies = [ops.prepack(q, 4) for _ in range(n_copies)]
            s_w = torch.full((N,), 0.01, device=dev)
            out = torch.empty((M, N), dtype=torch.float16, device=dev)
            plan = torch.zeros(2, dtype=torch.int32)
            ops.call("mq_gemm_debug_plan", M, N, a.K_pad, 4, 1, 1, plan[0:].data_ptr(), plan[1:].data_ptr())
            t0 = int(plan[0])
            tiles = [int(t) for t in args.tiles.split(",")] if args.tiles else [t0, t0 + 4]
            if not args.tiles and not 40 <= t0 <= 43:
                print(f"{name}: plan tile {t0} is not a wave-specialised one, skipped")
                break
            ops.gemm_debug_force(26, 1)
            ref = ops.gemm_w4a8_i32(a8, copies[0], 4, N)
            res = {t: [] for t in tiles}
            exact = True
            for r in range(args.rounds):
                for tile in tiles:
                    ops.gemm_debug_force(tile, 1)
                    if r == 0:
                        exact = exact and bool(torch.equal(ops.gemm_w4a8_i32(a, copies[0], 4, N), ref))
                    st = {"i": 0}

                    def call():
                        st["i"] = (st["i"] + 1) % n_copies
                        ops.gemm_w4a8(a, copies[st["i"]], 4, N, 0.02, s_w, out=out)
                    for _ in range(10):
                        call()
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    iters = 200
                    e0.record()
                    for _ in range(iters):
                        call()
                    e1.record()
                    torch.cuda.synchronize()
                    res[tile].append(e0.elapsed_time(e1) / iters * 1e3)
            ops.gemm_debug_force(-1, 0)
            f = lambda v: " / ".join(f"{x:6.2f}" for x in v)   # noqa: E731
            b0 = min(res[tiles[0]])
            print(f"{name:11s} {M} x {N} x {K} {kind:10s} (plan: tile {t0}): " + " | ".join(f"tile {t}: {f(res[t])} best {min(res[t]):6.2f} ({(min(res[t]) / b0 - 1) * 100:+5.1f} %)" for t in tiles)
                  + ("" if exact else "  INEXACT"), flush=True)
            del copies, a, q


if __name__ == "__main__":
    main()
