#!/usr/bin/env python3
"""Goldens for the dynamic GROUP-WISE activation quantizer with ASYMMETRIC levels (``--a_groupsize`` + ``--a_asym``; reference
fake_quant/quant_utils.py:181-203 ``find_params_per_token_groupwise`` (sym=False branch) + ``asym_quant_dequant`` :27-38):
the REFERENCE's ActQuantWrapper.forward (quant_utils.py:330-391) with ``quantizer.configure(bits, groupsize=g, sym=False,
clip_ratio=c)`` after an RTN weight pass, on CPU, for fp32 / fp16 / bf16 activations (every intermediate of the group search stays
in x's dtype).  Writes tests/golden/wrapper_grpa_<case>.npz: output, per-(row, group) scales and zero points, the stored int8
levels (q - 2^(bits-1)) and the per-group integer accumulators / weight sums.  Build-container only."""
import functools
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_golden  # noqa: E402
from golden_inputs import make_w, make_x  # noqa: E402

DT = {0: torch.float32, 1: torch.float16, 2: torch.bfloat16}
# tag: (K_in, K_pad, N, M, seed, had, bias, clip_ratio, a_bits, groupsize, dtype code)
CASES = {
    "g128_3584_f32": (3584, 3584, 48, 16, 2200, False, True, 1.0, 8, 128, 0),
    "g128_had_5120_f16": (5120, 5120, 32, 12, 2210, True, True, 1.0, 8, 128, 1),
    "g256_1280_f16_clip": (1280, 1280, 40, 24, 2220, False, False, 0.9, 8, 256, 1),
    "g64_2048_bf16": (2048, 2048, 32, 10, 2230, False, False, 1.0, 8, 64, 2),
    "g128_down_19968_f16": (18944, 19968, 24, 6, 2240, True, True, 1.0, 8, 128, 1),
    "g128_a6_1024_f32": (1024, 1024, 16, 9, 2250, False, False, 0.95, 6, 128, 0),
}


def main():
    gen_golden._install_shims()
    torch.set_grad_enabled(False)
    from fake_quant import hadamard_utils as hu
    from fake_quant import quant_utils as qu
    from fake_quant import utils as ru
    assert qu.__file__.startswith(gen_golden.REF)
    for tag, (K_in, K_pad, N, M, seed, had, bias, clip, a_bits, g, dtc) in CASES.items():
        dt = DT[dtc]
        lin = torch.nn.Linear(K_pad, N, bias=bias)
        lin.weight.data = torch.from_numpy(make_w(seed, (N, K_pad)))
        if bias:
            lin.bias.data = torch.from_numpy(make_w(seed + 1, (N,), std=0.1))
        lin = lin.to(dt)
        wrap = qu.ActQuantWrapper(lin)
        if had:
            hadK, Kh = hu.get_hadK(K_pad)
            wrap.online_full_had, wrap.had_K, wrap.K = True, hadK, Kh
        if K_pad != K_in:
            wrap.register_forward_pre_hook(functools.partial(ru.revise_down_input, new_size=K_pad))
        wq = qu.WeightQuantizer()
        wq.configure(4, perchannel=True, sym=True, mse=False)
        wq.find_params(lin.weight.data)
        lin.weight.data = wq.quantize(lin.weight.data)
        wscale = wq.scale.float().numpy().reshape(-1)
        wrap.quantizer.configure(bits=a_bits, groupsize=g, sym=False, clip_ratio=clip)
        x = torch.from_numpy(make_x(seed + 20, (1, M, K_in))).to(dt)
        if M > 2:
            x[0, 2, :min(g, K_in)] = x[0, 2, :min(g, K_in)].abs() + 0.5      # an all-positive group: the range does not include 0
            x[0, 1, :min(g, K_in)] = 0                                      # an all-zero group: range (-1, +1)
        y = wrap(x.clone())
        xt = x.clone()
        if K_pad != K_in:
            xt = torch.nn.functional.pad(xt, (0, K_pad - K_in))
        if had:
            xt = hu.matmul_hadU_cuda(xt, wrap.had_K, wrap.K)
        aq = qu.ActQuantizer()
        aq.configure(bits=a_bits, groupsize=g, sym=False, clip_ratio=clip)
        aq.find_params(xt)
        qx, scale, zero = aq.quantize(xt)
        G = K_pad // g
        s_groups = scale.reshape(M, G, g)[:, :, 0].float().numpy()
        z_groups = zero.reshape(M, G, g)[:, :, 0].float().numpy()
        half = float(1 << (a_bits - 1))
        stored = (qx.reshape(M, K_pad).float() - half)
        assert stored.min() >= -half and stored.max() <= half - 1
        Wq = lin.weight.data.float()
        qw = torch.round(Wq / torch.from_numpy(wscale).reshape(-1, 1)).to(torch.int64)
        acc = torch.einsum("mgk,ngk->mgn", stored.reshape(M, G, g).to(torch.int64), qw.reshape(N, G, g))
        wsum = qw.reshape(N, G, g).sum(dim=2).T.contiguous()                  # [G][N]
        gen_golden.save(f"wrapper_grpa_{tag}", y=y.float().numpy().reshape(M, N), s_w=wscale, s_groups=s_groups.astype(np.float32),
                        z_groups=z_groups.astype(np.float32), qx=stored.numpy().astype(np.int8), acc_groups=acc.numpy().astype(np.int32),
                        wsum_groups=wsum.numpy().astype(np.int32),
                        meta=np.array([K_in, K_pad, N, M, seed, int(had), int(bias), a_bits, g, dtc], np.int64),
                        clip=np.float32(clip))
        print(tag, "max|y|", float(y.abs().max()), "zero range", float(z_groups.min()), float(z_groups.max()))


if __name__ == "__main__":
    main()
