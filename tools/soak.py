#!/usr/bin/env python3
"""Sustained run of the benchmark step (the drop-in-built Qwen2-VL-7B prefill, one hipGraph): tokens/s per 10-second
window over several minutes -- does the rate hold once the part is warm?  Output -> profiles/r3_soak.txt"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
torch.set_grad_enabled(False)
from mquant_amd import workload  # noqa: E402

dev = torch.device("cuda:0")
minutes = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
pf = workload.WrapperPrefill(workload.qwen2vl_7b_specs(msq=True), device=dev)
pf.step()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    pf.step()
for _ in range(5):
    g.replay()
torch.cuda.synchronize()
t_end = time.perf_counter() + minutes * 60.0
w = 0
first = None
while time.perf_counter() < t_end:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 0
    t0 = time.perf_counter()
    e0.record()
    while time.perf_counter() - t0 < 10.0:
        for _ in range(50):
            g.replay()
        n += 50
        torch.cuda.current_stream().synchronize()
    e1.record()
    e1.synchronize()
    ms = e0.elapsed_time(e1) / n
    first = first or ms
    print(f"window {w:3d} ({w * 10:4d} s): {ms:7.4f} ms per step, {768 / ms * 1e3:9.1f} tokens/s  ({ms / first:5.3f} x the first window)", flush=True)
    w += 1
