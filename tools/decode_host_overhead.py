#!/usr/bin/env python3
"""Host time of ONE integer Linear forward at a generation shape (M = 1), eager: the engine (mquant_amd.engine.W4A8Linear), the drop-in
wrapper around it (fake_quant.quant_utils.ActQuantWrapper after model_quant), and for scale a plain fp16 nn.Linear of the same shape.
The kernels need ~9 us (quantize 3 + GEMM 6); whatever is above that is Python / ctypes / allocator time per call.
usage (GPU box): python3 tools/decode_host_overhead.py"""
import os
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fake_quant import quant_utils as qu  # noqa: E402
from fake_quant.gptq.rtn import rtn_module  # noqa: E402

dev = torch.device("cuda:0")
torch.set_grad_enabled(False)


def per_call(fn, n=2000):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def main():
    N, K = 3584, 3584
    root = torch.nn.Module()
    root.lin = torch.nn.Linear(K, N, bias=True, device=dev, dtype=torch.float16)
    plain = torch.nn.Linear(K, N, bias=True, device=dev, dtype=torch.float16)
    qu.add_actquant(root)
    wrap = root.lin
    wrap.quantizer.configure(bits=8, sym=True, static=True, observer_type="minmax")
    rtn_module(root, "m", 4, True, False, [], {})
    args = types.SimpleNamespace(skip_names=[], no_sibling_fusion=True)
    x = torch.randn((1, K), device=dev, dtype=torch.float16)
    qu.model_open_calibrate(root, args)
    qu.model_open_last_calibrate(root, args)
    wrap(torch.randn((64, K), device=dev, dtype=torch.float16))
    qu.model_close_calibrate(root, args)
    qu.model_quant(root, args)
    wrap(x)
    print("backend:", wrap.backend())
    eng = wrap._real
    out = torch.empty((1, N), device=dev, dtype=torch.float16)
    print(f"plain fp16 nn.Linear            {per_call(lambda: plain(x)):7.1f} us per call")
    print(f"engine.forward (out given)      {per_call(lambda: eng.forward(x, out=out)):7.1f} us per call")
    print(f"engine.forward                  {per_call(lambda: eng.forward(x)):7.1f} us per call")
    print(f"ActQuantWrapper.forward         {per_call(lambda: wrap(x)):7.1f} us per call")
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        for _ in range(10):
            wrap(x)
    print(f"the same from a hipGraph        {per_call(g.replay, 500) / 10:7.1f} us per call")


if __name__ == "__main__":
    main()
