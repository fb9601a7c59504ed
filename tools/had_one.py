#!/usr/bin/env python3
"""Run the down_proj Hadamard+quant kernel a few times (for rocprofv3 --pmc passes)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fake_quant import hadamard_utils as hu
from mquant_amd import ops
dev = torch.device("cuda:0")
M, n_in, n = 768, 18944, 19968
_, K = hu.get_hadK(n)
bits = hu.had_sign_bits(K, dev)
x = torch.randn((M, n_in), device=dev, dtype=torch.float32).half()
out = torch.empty((M, n), dtype=torch.int8, device=dev)
for _ in range(5):
    ops.hadamard_quant_i8(x, n, K, bits, 0.05, out=out)
torch.cuda.synchronize()
