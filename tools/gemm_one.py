#!/usr/bin/env python3
"""Run one GEMM shape a few times (for rocprofv3 --pmc passes), activations in the tiled layout (what the engine feeds: the plan's
ping-pong / wave-specialised kernels); MQ_ROWS=1: row-major activations.  argv: M N K [tile splits]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mquant_amd import ops
M, N, K = (int(v) for v in sys.argv[1:4])
tile, splits = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (-1, 0)
dev = torch.device("cuda:0")
ops.splitk_workspace(dev, 512 << 20)
a = torch.randint(-128, 128, (M, K), dtype=torch.int8, device=dev)
if not os.environ.get("MQ_ROWS"):
    a = ops.TiledAct.from_rows(a)
q = torch.randint(-8, 8, (N, K), dtype=torch.int8, device=dev)
img = ops.prepack(q, 4)
s_w = torch.full((N,), 0.01, device=dev)
out = torch.empty((M, N), dtype=torch.float16, device=dev)
ops.gemm_debug_force(tile, splits)
for _ in range(5):
    ops.gemm_w4a8(a, img, 4, N, 0.02, s_w, out=out)
torch.cuda.synchronize()
