#!/usr/bin/env python3
"""Goldens for the dynamic PER-TENSOR activation quantizer (``act_per_tensor=True``; reference
fake_quant/quant_utils.py:214-237) on HALF-precision activations: the REFERENCE's ActQuantWrapper.forward
(quant_utils.py:330-391) after an RTN weight pass, on CPU, fp16 / bf16.  The reference keeps the range, the scale, the
zero point, x / scale and the level sum of this mode in x's dtype (``torch.tensor(0).to(x)``; the int64 maxq tensor does
not promote), so its grid differs from an fp32 evaluation.  Writes tests/golden/wrapper_dynpt16_<case>.npz: the output,
scale / zero, the int8 levels as the integer GEMM stores them and the integer accumulators restated from the reference's
own quantizer.  Build-container only."""
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

DT = {1: torch.float16, 2: torch.bfloat16}
# tag: (K_in, K_pad, N, M, seed, had, split, bias, clip_ratio, a_bits, sym, dtype code)
CASES = {
    "sym_3584_f16": (3584, 3584, 48, 16, 2300, False, False, True, 1.0, 8, True, 1),
    "sym_had_5120_split_f16": (5120, 5120, 32, 12, 2310, True, True, True, 0.9, 8, True, 1),
    "asym_1280_bf16": (1280, 1280, 40, 24, 2320, False, False, False, 1.0, 8, False, 2),
    "asym_down_19968_f16": (18944, 19968, 24, 6, 2330, True, False, True, 0.95, 8, False, 1),
    "sym_2048_bf16_clip": (2048, 2048, 32, 10, 2340, False, False, False, 0.9, 8, True, 2),
}


def main():
    gen_golden._install_shims()
    torch.set_grad_enabled(False)
    from fake_quant import hadamard_utils as hu
    from fake_quant import quant_utils as qu
    from fake_quant import utils as ru
    assert qu.__file__.startswith(gen_golden.REF)
    for tag, (K_in, K_pad, N, M, seed, had, split, bias, clip, a_bits, sym, dtc) in CASES.items():
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
        wrap.quantizer.configure(bits=a_bits, sym=sym, clip_ratio=clip, act_per_tensor=True)
        x = torch.from_numpy(make_x(seed + 20, (M, K_in))).to(dt)
        y = wrap(x.clone())
        xt = x.clone()
        if K_pad != K_in:
            xt = torch.nn.functional.pad(xt, (0, K_pad - K_in))
        if had:
            xt = hu.matmul_hadU_cuda(xt, wrap.had_K, wrap.K)
        xq_in = xt[..., 1:] if split else xt
        aq = qu.ActQuantizer()
        aq.configure(bits=a_bits, sym=sym, clip_ratio=clip, act_per_tensor=True)
        aq.find_params(xq_in)
        zero = 0.0
        if sym:
            qx, s = aq.quantize(xq_in)
        else:                       # levels 0 .. 2^bits - 1; the int8 GEMM stores them minus 2^(bits-1)
            qx, s, zero = aq.quantize(xq_in)
            qx = qx.float() - float(1 << (a_bits - 1))
            zero = float(zero)
        assert qx.dtype in (dt, torch.float32) and torch.as_tensor(s).dtype == dt      # the whole search stayed in x's dtype
        Wq = (wrap.L2.weight.data if split else wrap.module.weight.data).float()
        qw = torch.round(Wq / torch.from_numpy(wscale).reshape(-1, 1)).to(torch.int64)
        acc = qx.to(torch.int64) @ qw.T
        out = dict(y=y.float().numpy(), s_w=wscale, scale=np.float32(float(s)), zero=np.float32(zero),
                   acc=acc.numpy().astype(np.int32), qx=qx.float().numpy().astype(np.int8),
                   meta=np.array([K_in, K_pad, N, M, seed, int(had), int(split), int(bias), a_bits, int(sym), dtc], np.int64),
                   clip=np.float32(clip))
        if split:
            out["x0"] = xt[..., 0].float().numpy()
        gen_golden.save(f"wrapper_dynpt16_{tag}", **out)
        print(tag, "scale", float(s), "zero", zero, "max|y|", float(y.float().abs().max()), "levels", int(qx.min()), int(qx.max()))


if __name__ == "__main__":
    main()
