#!/usr/bin/env python3
"""Goldens for the DYNAMIC per-token activation mode (the reference's default when no *_static
flag is given): the REFERENCE's ActQuantWrapper.forward (fake_quant/quant_utils.py:330-391) with
``quantizer.configure(bits=8, sym=True)`` after an RTN weight pass, on CPU.  Writes
tests/golden/wrapper_dyn_<case>.npz with the outputs, the per-row scales and the integer
accumulators restated from the reference's own quantizers.  Build-container only."""
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

# tag: (K_in, K_pad, N, M, seed, had, split, bias, clip_ratio, a_bits[, sym[, per_tensor]])
CASES = {
    # one range for the whole tensor (act_per_tensor)
    "pt_sym_3584": (3584, 3584, 48, 16, 1800, False, False, True, 1.0, 8, True, True),
    "pt_sym_had_5120_split": (5120, 5120, 32, 12, 1810, True, True, True, 0.9, 8, True, True),
    "pt_asym_1280": (1280, 1280, 40, 24, 1820, False, False, False, 1.0, 8, False, True),
    "pt_asym_down_19968": (18944, 19968, 24, 6, 1830, True, False, True, 0.95, 8, False, True),
    # asymmetric (--a_asym): levels 0..2^bits-1 with a zero point per row
    "asym_3584": (3584, 3584, 48, 16, 1700, False, False, True, 1.0, 8, False),
    "asym_clip_1280": (1280, 1280, 40, 24, 1710, False, False, False, 0.9, 8, False),
    "asym_down_19968": (18944, 19968, 24, 6, 1720, True, False, True, 1.0, 8, False),
    "asym_a6_2048": (2048, 2048, 32, 10, 1730, False, False, False, 1.0, 6, False),
    "plain_3584": (3584, 3584, 48, 16, 1600, False, False, True, 1.0, 8),
    "clip_1280": (1280, 1280, 40, 24, 1610, False, False, False, 0.9, 8),
    "had_5120_split": (5120, 5120, 32, 12, 1620, True, True, True, 1.0, 8),
    "down_19968": (18944, 19968, 24, 6, 1630, True, False, False, 1.0, 8),
}


def main():
    gen_golden._install_shims()
    torch.set_grad_enabled(False)
    from fake_quant import hadamard_utils as hu
    from fake_quant import quant_utils as qu
    from fake_quant import utils as ru
    assert qu.__file__.startswith(gen_golden.REF)
    for tag, case in CASES.items():
        K_in, K_pad, N, M, seed, had, split, bias, clip, a_bits = case[:10]
        sym = case[10] if len(case) > 10 else True
        per_tensor = case[11] if len(case) > 11 else False
        lin = torch.nn.Linear(K_pad, N, bias=bias)
        lin.weight.data = torch.from_numpy(make_w(seed, (N, K_pad)))
        if bias:
            lin.bias.data = torch.from_numpy(make_w(seed + 1, (N,), std=0.1))
        wrap = qu.ActQuantWrapper(lin)
        if had:
            hadK, Kh = hu.get_hadK(K_pad)
            wrap.online_full_had, wrap.had_K, wrap.K = True, hadK, Kh
        if split:
            wrap.split = True
            wrap.split_weights()
        if K_pad != K_in:
            wrap.register_forward_pre_hook(functools.partial(ru.revise_down_input, new_size=K_pad))
        wscale = None
        for name, sub in qu.find_qlayers(wrap, layers=[torch.nn.Linear]).items():
            if "L1" in name:
                continue
            wq = qu.WeightQuantizer()
            wq.configure(4, perchannel=True, sym=True, mse=False)
            wq.find_params(sub.weight.data)
            sub.weight.data = wq.quantize(sub.weight.data)
            if name in ("module", "L2"):
                wscale = wq.scale.float().numpy().reshape(-1)
        wrap.quantizer.configure(bits=a_bits, sym=sym, clip_ratio=clip, act_per_tensor=per_tensor)
        x = make_x(seed + 20, (M, K_in))
        y = wrap(torch.from_numpy(x.copy()))
        # integer restatement from the reference's own dynamic quantizer
        xt = torch.from_numpy(x.copy())
        if K_pad != K_in:
            xt = torch.nn.functional.pad(xt, (0, K_pad - K_in))
        if had:
            xt = hu.matmul_hadU_cuda(xt, wrap.had_K, wrap.K)
        xq_in = xt[..., 1:] if split else xt
        aq = qu.ActQuantizer()
        aq.configure(bits=a_bits, sym=sym, clip_ratio=clip, act_per_tensor=per_tensor)
        aq.find_params(xq_in)
        zero = None
        if sym:
            qx, s_rows = aq.quantize(xq_in)
        else:                       # levels 0 .. 2^bits - 1; the int8 GEMM stores them minus 2^(bits-1)
            qx, s_rows, zero = aq.quantize(xq_in)
            qx = qx - float(1 << (a_bits - 1))
        Wq = (wrap.L2.weight.data if split else wrap.module.weight.data).float()
        qw = torch.round(Wq / torch.from_numpy(wscale).reshape(-1, 1)).to(torch.int64)
        acc = qx.to(torch.int64) @ qw.T
        if per_tensor:               # scalars: one value for every row
            s_rows = torch.as_tensor(s_rows, dtype=torch.float32).reshape(1, 1).expand(xq_in.shape[0], 1)
            if zero is not None:
                zero = torch.as_tensor(zero, dtype=torch.float32).reshape(1, 1).expand(xq_in.shape[0], 1)
        out = dict(y=y.numpy(), s_w=wscale, s_rows=s_rows[:, 0].numpy().astype(np.float32),
                   acc=acc.numpy().astype(np.int32), qx_head=qx[:, :64].numpy().astype(np.int8),
                   meta=np.array([K_in, K_pad, N, M, seed, int(had), int(split), int(bias), a_bits], np.int64),
                   clip=np.float32(clip))
        if split:
            out["x0"] = xt[..., 0].numpy()
        if zero is not None:
            out["zero"] = zero[:, 0].numpy().astype(np.float32)
            out["sym"] = np.int64(0)
        if per_tensor:
            out["per_tensor"] = np.int64(1)
        gen_golden.save(f"wrapper_dyn_{tag}", **out)


if __name__ == "__main__":
    main()
