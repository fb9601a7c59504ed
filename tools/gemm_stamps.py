#!/usr/bin/env python3
"""Phase stamps of one workgroup of the wave-specialised GEMM (library built with -DMQ_STAMP=<block>).
Usage (GPU box): MQUANT_HIP_LIB=.../libmquant_hip_stamp.so python tools/gemm_stamps.py M N K tile"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mquant_amd import ops  # noqa: E402

M, N, K, tile = (int(v) for v in sys.argv[1:5])
dev = torch.device("cuda:0")
ws = ops.splitk_workspace(dev, 64 << 20)
a = torch.randint(-128, 128, (M, K), dtype=torch.int8, device=dev)
q = torch.randint(-8, 8, (N, K), dtype=torch.int8, device=dev)
img = ops.prepack(q, 4)
s_w = torch.full((N,), 0.01, device=dev)
bias = torch.zeros((N,), device=dev)
a_t = ops.TiledAct.from_rows(a)
ops.gemm_debug_force(tile, 1)
names = ["entry", "setup done / loads issued", "stage 0 landed (loader) / at B(0) (math)", "past B(0)", "k-loop done",
         "ring free", "slab parked", "stores issued"]
for rep in range(4):
    ws[:4096].zero_()
    ops.gemm_w4a8(a_t, img, 4, N, 0.02, s_w, bias=bias)
    torch.cuda.synchronize()
    st = ws[:4096].view(torch.int64).cpu().tolist()
    t0 = min(v for v in st if v)
    print(f"rep {rep}: math  " + "  ".join(f"{i}:{st[i] - t0 if st[i] else -1}" for i in range(8)))
    print(f"        load  " + "  ".join(f"{i}:{st[8 + i] - t0 if st[8 + i] else -1}" for i in range(8)))
    nk = min(K // 128, 200)
    rel = [st[32 + i] - st[8] for i in range(nk)]
    print("        step boundaries (cycles since loader entry): " + " ".join(str(v) for v in rel[:40]))
    d = [rel[i + 1] - rel[i] for i in range(nk - 1)]
    print("        step lengths: " + " ".join(str(v) for v in d[:40]) + (f" ... mean of the rest {sum(d[40:]) / max(len(d[40:]), 1):.0f}" if len(d) > 40 else ""))
print("slots:", "; ".join(f"{i}={n}" for i, n in enumerate(names)))
