#!/usr/bin/env python3
"""Per-k-tile cost of a GEMM tile shape: one full round of tiles (M = 768, N chosen for ~255 workgroups) timed at three
reduction lengths, cold weights; the slope is the steady-state k-loop, the intercept the fixed cost per launch.
Usage (GPU box): python tools/gemm_kslope.py --tile 14 [--bm 256 --bn 256] [--libs main,abl1,...]"""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(tile, bm, bn, ks):
    import torch
    sys.path.insert(0, ROOT)
    from mquant_amd import ops
    dev = torch.device("cuda:0")
    M = 768
    mb = -(-M // bm)
    nb = 255 // mb
    N = nb * bn
    # exactness of the forced tile against the 128 x 128 symmetric kernel on a small ragged problem
    at = torch.randint(-128, 128, (300, 1792), dtype=torch.int8, device=dev)
    qt = torch.randint(-8, 8, (520, 1792), dtype=torch.int8, device=dev)
    it = ops.prepack(qt, 4)
    ops.gemm_debug_force(26, 1)
    ref = ops.gemm_w4a8_i32(at, it, 4, 520)
    ops.gemm_debug_force(tile, 1)
    got = ops.gemm_w4a8_i32(ops.TiledAct.from_rows(at), it, 4, 520)
    exact = bool(torch.equal(ref, got))
    res = []
    stamps = []
    for K in ks:
        a = ops.TiledAct.from_rows(torch.randint(-128, 128, (M, K), dtype=torch.int8, device=dev))
        q = torch.randint(-8, 8, (N, K), dtype=torch.int8, device=dev)
        copies = [ops.prepack(q, 4) for _ in range(1 if os.environ.get('MQ_WARM') else max(2, int(700e6 // (N * K // 2))))]
        s_w = torch.full((N,), 0.01, device=dev)
        out = torch.empty((M, N), dtype=torch.float16, device=dev)
        ops.gemm_debug_force(tile, 1)
        st = {"i": 0}

        def call():
            st["i"] = (st["i"] + 1) % len(copies)
            ops.gemm_w4a8(a, copies[st["i"]], 4, N, 0.02, s_w, out=out)
        for _ in range(5):
            call()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            iters = max(30, 2 * len(copies))
            e0.record()
            for _ in range(iters):
                call()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / iters * 1e3)
        res.append(best)
        if os.environ.get("MQ_STAMPS"):
            torch.cuda.synchronize()
            per = 8 if tile in (14, 15, 16, 17, 18, 19) else 2      # ints per workgroup: gemm_pp.hip writes 8 (incl. s_memrealtime), gemm_ws.hip 2
            st_ = ops.splitk_workspace(dev)[: mb * nb * per * 4].view(torch.int32).view(-1, per).cpu()
            cyc = st_[:, 0].float()
            real = f", in-kernel s_memtime / s_memrealtime = {(cyc / st_[:, 2].float() * 0.1).median().item():.3f} GHz" if per == 8 else ""
            print(f"    K={K}: loop cycles (incl. prologue) median {cyc.median().item():.0f} min {cyc.min().item():.0f} max {cyc.max().item():.0f}, k-tiles {st_[0, 1].item()}{real}", flush=True)
            stamps.append(cyc.median().item())
        del copies, q
    if len(stamps) == len(ks):
        cs = (stamps[-1] - stamps[0]) / ((ks[-1] - ks[0]) / 64)
        ts = (res[-1] - res[0]) / ((ks[-1] - ks[0]) / 64) * 1e3
        print(f"    cycles per k-tile {cs:.0f}, ns per k-tile {ts:.1f} -> shader clock {cs / ts:.2f} GHz", flush=True)
    return mb * nb, N, res, exact


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tile", type=int, default=14)
    ap.add_argument("--bm", type=int, default=256)
    ap.add_argument("--bn", type=int, default=256)
    ap.add_argument("--ks", default="1792,3584,7168")
    ap.add_argument("--libs", default="")
    ap.add_argument("--worker", action="store_true")
    args = ap.parse_args()
    ks = [int(k) for k in args.ks.split(",")]
    if args.worker or not args.libs:
        tiles, N, res, exact = worker(args.tile, args.bm, args.bn, ks)
        slope = (res[-1] - res[0]) / ((ks[-1] - ks[0]) / 64)
        fixed = res[0] - slope * ks[0] / 64
        ideal = 2.0 * args.bm * args.bn * 64 / (5e15 / 256) * 1e6
        print(f"tile {args.tile} ({args.bm}x{args.bn}, {tiles} workgroups, N={N}): " + "  ".join(f"K={k}: {r:7.1f} us" for k, r in zip(ks, res))
              + f"  | per 64-wide k-tile {slope * 1e3:7.1f} ns (MFMA floor {ideal * 1e3:.1f} ns -> {ideal / slope:.3f}), fixed {fixed:5.1f} us{'' if exact else '  INEXACT'}", flush=True)
        return
    for lib in args.libs.split(","):
        path = os.path.join(ROOT, "mquant_amd", "libmquant_hip.so" if lib == "main" else f"libmquant_hip_{lib}.so")
        env = dict(os.environ, MQUANT_HIP_LIB=path)
        out = subprocess.run([sys.executable, __file__, "--worker", "--tile", str(args.tile), "--bm", str(args.bm), "--bn", str(args.bn),
                              "--ks", args.ks], env=env, capture_output=True, text=True)
        print(f"{lib:8s} {out.stdout.strip()}{out.stderr.strip()[-300:] if out.returncode else ''}", flush=True)


if __name__ == "__main__":
    main()
