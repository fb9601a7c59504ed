#!/usr/bin/env python3
"""Same-box A/B of two GEMM tile ids per shape on the benchmark's own layers (real weight images, calibrated scales, bias / split
terms as the workload has them): every shape's launches back to back from a hipGraph, alternating the two ids.
usage (GPU box): python3 tools/r6_direct_ab.py [--pairs 47:53,48:54,45:51,46:52] [--rounds 5]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mquant_amd import floor_model, ops, workload  # noqa: E402
from mquant_amd.engine import WORKSPACE  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", default="53:47,54:48,51:45,52:46", help="plan tile : the tile to compare it with")
    ap.add_argument("--rounds", type=int, default=5)
    args = ap.parse_args()
    pairs = dict(tuple(int(v) for v in p.split(":")) for p in args.pairs.split(","))
    torch.set_grad_enabled(False)
    dev = torch.device("cuda:0")
    pf = workload.Prefill(workload.qwen2vl_7b_specs(msq=True), device=dev, share_groups=True)
    groups = floor_model.shape_groups(pf.layers)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def graph_of(grp, tile):
        def run():
            for L in grp["layers"]:
                a = WORKSPACE.act(dev, L.spec.M, L.lin.K_pad)
                x0 = WORKSPACE.x0(dev, L.spec.M) if L.lin.split else None
                L.lin.gemm(a, x0, torch.float16, L.row_sel, L.out)
        ops.gemm_debug_force(tile, 0)
        run()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            run()
        ops.gemm_debug_force(-1, 0)
        return g

    def timed(g, n):
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(10):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 10 / n * 1e3

    total = {}
    for grp in groups:
        tile = floor_model.plan_tile(grp["M"], grp["N"], grp["K_pad"], grp["w_bits"])
        if tile not in pairs:
            continue
        n = len(grp["layers"])
        # outputs must agree bit for bit
        L = grp["layers"][0]
        a, x0 = L.lin.quantize(L.x, L.row_sel)
        ys = []
        for t in (tile, pairs[tile]):
            ops.gemm_debug_force(t, 0)
            ys.append(L.lin.gemm(a, x0, torch.float16, L.row_sel).clone())
        ops.gemm_debug_force(-1, 0)
        same = bool(torch.equal(ys[0], ys[1]))
        ga, gb = graph_of(grp, tile), graph_of(grp, pairs[tile])
        ta, tb = [], []
        for _ in range(args.rounds):
            ta.append(timed(ga, n))
            tb.append(timed(gb, n))
        ta.sort(), tb.sort()
        ma, mb = ta[len(ta) // 2], tb[len(tb) // 2]
        total[grp["name"]] = (n, ma, mb)
        print(f"{grp['name']:18s} {grp['M']}x{grp['N']}x{grp['K_pad']:<6d} x{n:<3d} tile {tile}: {ma:7.2f} us   tile {pairs[tile]}: {mb:7.2f} us   "
              f"({(mb - ma):+.2f} us, {(mb / ma - 1) * 100:+.1f} %)  equal={same}")
    da = sum(n * a for n, a, b in total.values()) * 1e-3
    db = sum(n * b for n, a, b in total.values()) * 1e-3
    print(f"per step over these shapes: {da:.3f} ms -> {db:.3f} ms ({db - da:+.3f} ms)")


if __name__ == "__main__":
    main()
