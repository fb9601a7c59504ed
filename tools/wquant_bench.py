#!/usr/bin/env python3
"""mq_wquant_sym timings on the Qwen2-VL-7B weight shapes (GPU box), with the torch restatement
of the same search (fake_quant.WeightQuantizer, use_kernel=False) beside it."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fake_quant import quant_utils as qu  # noqa: E402
from mquant_amd import ops  # noqa: E402


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    dev = torch.device("cuda:0")
    print(f"{'shape':>16s} {'mode':>4s} {'kernel ms':>10s} {'GB/s(alg)':>10s} {'torch ms':>10s} {'same scales':>12s}")
    for N, K in ((3584, 3584), (1280, 5120), (18944, 3584), (3584, 19968)):
        w = (torch.randn((N, K), device=dev) * 0.02).half()
        for mse in (False, True):
            ms = timed(lambda: ops.wquant_sym(w, 4, mse, want_packed=True), 5 if mse else 20)
            ref = qu.WeightQuantizer()
            ref.configure(4, perchannel=True, sym=True, mse=mse)
            ref.use_kernel = False
            tms = timed(lambda: ref.find_params(w), 2 if mse else 10)
            s = ops.wquant_sym(w, 4, mse, want_levels=False)[0]
            same = (s == ref.scale.reshape(-1)).float().mean().item()
            passes = 81 if mse else 2
            print(f"{N:>7d}x{K:<8d} {'mse' if mse else 'rtn':>4s} {ms:10.3f} {passes * N * K * 2 / ms / 1e6:10.1f} {tms:10.2f} {same:12.4f}")


def gptq_bench():
    from fake_quant.gptq.gptq_utils import GPTQ
    dev = torch.device("cuda:0")
    torch.linalg.cholesky(torch.eye(256, device=dev) * 2.0)      # rocSOLVER start-up outside the timings
    print(f"\n{'GPTQ shape':>16s} {'fused ms':>10s} {'torch-loop ms':>14s} {'identical':>10s}")
    for N, K in ((1280, 1280), (3584, 3584), (1280, 5120), (3584, 19968)):
        x = torch.randn((2048, K), device=dev)
        res = {}
        for fused in (True, False):
            if not fused and K > 8192:
                res[fused] = (float("nan"), None)
                continue
            torch.manual_seed(0)
            lin = torch.nn.Linear(K, N, bias=False).to(dev)
            lin.weight.data = torch.randn((N, K), device=dev) * 0.02
            solver = GPTQ(lin)
            solver.use_kernel = fused
            solver.quantizer = qu.WeightQuantizer()
            solver.quantizer.configure(4, perchannel=True, sym=True, mse=False)
            solver.add_batch(x, None)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            solver.fasterquant()
            torch.cuda.synchronize()
            res[fused] = ((time.perf_counter() - t0) * 1e3, lin.weight.data.clone())
        same = "-" if res[False][1] is None else str(bool(torch.equal(res[True][1], res[False][1])))
        print(f"{N:>7d}x{K:<8d} {res[True][0]:10.1f} {res[False][0]:14.1f} {same:>10s}")


if __name__ == "__main__":
    main()
    gptq_bench()
