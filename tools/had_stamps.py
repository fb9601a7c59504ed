#!/usr/bin/env python3
"""Phase split of the vector-ALU Hadamard kernel from a -DMQ_HV_STAMP build (cycles per wave: A load+butterflies incl. barrier, B K x K, C quantize+store)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fake_quant import hadamard_utils as hu
from mquant_amd import ops
dev = torch.device("cuda:0")
impl = int(os.environ.get("HAD_IMPL", "0"))
ops.hadamard_debug_impl(impl)
for name, M, n_in, n, NW in (("llm.down", 768, 18944, 19968, 8 if impl else 6), ("vis.fc2", 1024, 5120, 5120, 4 if impl else 2)):
    _, K = hu.get_hadK(n)
    bits = hu.had_sign_bits(K, dev)
    x = torch.randn((M, n_in), device=dev, dtype=torch.float16)
    out = ops.TiledAct.empty(M, n, dev)
    st = torch.zeros((M * NW * 4,), dtype=torch.float32, device=dev)
    for _ in range(3):
        ops.hadamard_quant_i8(x, n, K, bits, 0.05, out=out, x0_out=st)
    torch.cuda.synchronize()
    s = st.view(M, NW, 4).cpu()
    a, b, c, t0 = s[..., 0], s[..., 1], s[..., 2], s[..., 3]
    print(f"{name}: A {a.median():.0f} (min {a.min():.0f} max {a.max():.0f})  B {b.median():.0f} (min {b.min():.0f} max {b.max():.0f})  C {c.median():.0f} (max {c.max():.0f}) cycles;"
          f" start spread {(t0.max() - t0.min()):.0f}")
    for w in range(NW):
        print(f"   wave {w}: A {a[:, w].median():.0f} B {b[:, w].median():.0f} C {c[:, w].median():.0f}")
