#!/usr/bin/env python3
"""Goldens for ASYMMETRIC weights (--w_asym): the REFERENCE's ActQuantWrapper.forward
(fake_quant/quant_utils.py:330-391) after an RTN pass with WeightQuantizer(sym=False)
(quant_utils.py:446-518), static symmetric int8 activations (and one dynamic per-token case), on CPU.
Writes tests/golden/wrapper_wasym_<case>.npz: outputs, scales, zero points and the integer accumulators
restated from the reference's own quantizers.  Build-container only."""
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

# tag: (K_in, K_pad, N, M, seed, had, bias, w_bits, w_mse, dynamic)
CASES = {
    "plain_3584": (3584, 3584, 48, 16, 1900, False, True, 4, False, False),
    "mse_1280": (1280, 1280, 40, 24, 1910, False, False, 4, True, False),
    "w8_2048": (2048, 2048, 32, 12, 1920, False, True, 8, False, False),
    "down_19968": (18944, 19968, 24, 6, 1930, True, False, 4, False, False),
    "dyn_3584": (3584, 3584, 32, 10, 1940, False, True, 4, False, True),
}


class Args:
    skip_names = []


def main():
    gen_golden._install_shims()
    torch.set_grad_enabled(False)
    from fake_quant import hadamard_utils as hu
    from fake_quant import quant_utils as qu
    from fake_quant import utils as ru
    assert qu.__file__.startswith(gen_golden.REF)
    for tag, (K_in, K_pad, N, M, seed, had, bias, w_bits, w_mse, dynamic) in CASES.items():
        lin = torch.nn.Linear(K_pad, N, bias=bias)
        lin.weight.data = torch.from_numpy(make_w(seed, (N, K_pad)))
        if bias:
            lin.bias.data = torch.from_numpy(make_w(seed + 1, (N,), std=0.1))
        wrap = qu.ActQuantWrapper(lin)
        if had:
            hadK, Kh = hu.get_hadK(K_pad)
            wrap.online_full_had, wrap.had_K, wrap.K = True, hadK, Kh
        if K_pad != K_in:
            wrap.register_forward_pre_hook(functools.partial(ru.revise_down_input, new_size=K_pad))
        wq = qu.WeightQuantizer()
        wq.configure(w_bits, perchannel=True, sym=False, mse=w_mse)
        Wd = wrap.module.weight.data
        wq.find_params(Wd)
        wrap.module.weight.data = wq.quantize(Wd)
        s_w = wq.scale.float().numpy().reshape(-1)
        z_w = wq.zero.float().numpy().reshape(-1)
        out = dict(s_w=s_w, z_w=z_w)
        if dynamic:
            wrap.quantizer.configure(bits=8, sym=True, clip_ratio=1.0)
        else:
            wrap.quantizer.configure(bits=8, sym=True, static=True, observer_type="minmax")
            args = Args()
            qu.model_open_calibrate(wrap, args)
            calib = [make_x(seed + 10 + i, (M, K_in)) for i in range(3)]
            for i, c in enumerate(calib):
                if i == len(calib) - 1:
                    qu.model_open_last_calibrate(wrap, args)
                wrap(torch.from_numpy(c.copy()))
            qu.model_close_calibrate(wrap, args)
            qu.model_quant(wrap, args)
            out["s_x"] = np.asarray(wrap.quantizer.quantizer.scale.numpy(), dtype=np.float32)
        x = make_x(seed + 20, (M, K_in))
        out["y"] = wrap(torch.from_numpy(x.copy())).numpy()
        # integer restatement
        xt = torch.from_numpy(x.copy())
        if K_pad != K_in:
            xt = torch.nn.functional.pad(xt, (0, K_pad - K_in))
        if had:
            xt = hu.matmul_hadU_cuda(xt, wrap.had_K, wrap.K)
        if dynamic:
            aq = qu.ActQuantizer()
            aq.configure(bits=8, sym=True, clip_ratio=1.0)
            aq.find_params(xt)
            qx, s_rows = aq.quantize(xt)
            out["s_rows"] = s_rows[:, 0].numpy().astype(np.float32)
            qx = qx.to(torch.int64)
        else:
            qx = wrap.quantizer.quantizer.quant(xt.float()).to(torch.int64)
        Wq = wrap.module.weight.data.float()
        q_w = torch.round(Wq / torch.from_numpy(s_w).reshape(-1, 1)) + torch.from_numpy(z_w).reshape(-1, 1)
        assert q_w.min() >= 0 and q_w.max() <= 2 ** w_bits - 1
        stored = (q_w - 2 ** (w_bits - 1)).to(torch.int64)
        out["acc"] = (qx @ stored.T).numpy().astype(np.int32)
        out["qw_head"] = stored[:, :64].numpy().astype(np.int8)
        out["qx_sum"] = qx.sum(dim=1).numpy()
        out["meta"] = np.array([K_in, K_pad, N, M, seed, int(had), int(bias), w_bits, int(w_mse), int(dynamic)], np.int64)
        gen_golden.save(f"wrapper_wasym_{tag}", **out)


if __name__ == "__main__":
    main()
