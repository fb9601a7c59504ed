#!/usr/bin/env python3
"""Goldens for group-wise WEIGHT scales (``--w_groupsize``; exam/quant_qwen2vl.py:327, consumed at
fake_quant/gptq/gptq_utils.py:263-273): the REFERENCE's GPTQ solver (``GPTQ.fasterquant(groupsize=g, actorder=False,
static_groups=False)``, gptq_utils.py:197-303) on a seeded Linear with a seeded Hessian, then the REFERENCE's
``ActQuantWrapper.forward`` (quant_utils.py:330-391) over the solved weights in three activation modes:

  static   int8 per-tensor static quantizer through the calibration protocol (MinmaxObserver, UniformQuantizer)
  dyn      dynamic per-token symmetric (the reference's default)
  agrp     dynamic group-wise symmetric with --a_groupsize == --w_groupsize (quant_utils.py:181-203)

Writes tests/golden/wrapper_wgrp_<case>.npz: the solved weights, EVERY group's scale (the reference's quantizer only keeps the
last; they are captured by recording ``find_params`` as the solver calls it), the wrapper's output, the activation scales and
int8 levels of the reference's own quantizers and the exact per-group integer accumulators restated from them.
Build-container only (imports /root/reference; nothing of it is copied)."""
import functools
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_golden  # noqa: E402
from golden_inputs import make_w, make_x  # noqa: E402

DT = {0: torch.float32, 1: torch.float16}
# tag: (K_in, K_pad, N, M, seed, had, bias, w_groupsize, act mode, dtype code, w_bits)
CASES = {
    "g128_static_1024_f32": (1024, 1024, 40, 16, 3100, False, True, 128, "static", 0, 4),
    "g128_static_had_1280_f16": (1280, 1280, 32, 12, 3110, True, True, 128, "static", 1, 4),
    "g64_static_512_f32": (512, 512, 24, 10, 3120, False, False, 64, "static", 0, 4),
    "g128_dyn_1024_f16": (1024, 1024, 32, 14, 3130, False, True, 128, "dyn", 1, 4),
    "g256_dyn_1536_f32_w8": (1536, 1536, 24, 9, 3140, False, False, 256, "dyn", 0, 8),
    "g128_agrp_1024_f32": (1024, 1024, 32, 12, 3150, False, True, 128, "agrp", 0, 4),
    "g128_agrp_had_1280_f16": (1280, 1280, 24, 8, 3160, True, False, 128, "agrp", 1, 4),
    "g128_static_pad_896_f32": (896, 1024, 24, 8, 3170, False, False, 128, "static", 0, 4),
    # --act_order (tags "ao_..."): the groups are runs of the solver's PERMUTED columns (gptq_utils.py:226-230, 263-273)
    "ao_g128_static_1024_f32": (1024, 1024, 40, 16, 3180, False, True, 128, "static", 0, 4),
    "ao_g64_dyn_512_f16": (512, 512, 24, 10, 3190, False, False, 64, "dyn", 1, 4),
    "ao_g128_static_had_1280_f16": (1280, 1280, 32, 12, 3200, True, True, 128, "static", 1, 4),
}


class Args:
    skip_names = []


def calib_inputs(seed, K_in, n=3, rows=40):
    return [torch.from_numpy(make_x(seed + 1 + i, (rows, K_in))) for i in range(n)]


def main():
    gen_golden._install_shims()
    unf = types.ModuleType("unfoldNd")
    unf.UnfoldNd = object
    sys.modules["unfoldNd"] = unf
    torch.cuda.synchronize = lambda *a, **k: None
    torch.set_grad_enabled(False)
    from fake_quant import hadamard_utils as hu
    from fake_quant import quant_utils as qu
    from fake_quant import utils as ru
    from fake_quant.gptq import gptq_utils as gu
    assert qu.__file__.startswith(gen_golden.REF) and gu.__file__.startswith(gen_golden.REF)
    for tag, (K_in, K_pad, N, M, seed, had, bias, g, mode, dtc, w_bits) in CASES.items():
        dt = DT[dtc]
        lin = torch.nn.Linear(K_pad, N, bias=bias)
        lin.weight.data = torch.from_numpy(make_w(seed, (N, K_pad))) * 4.0
        if bias:
            lin.bias.data = torch.from_numpy(make_w(seed + 1, (N,), std=0.1))
        # ---- the reference's GPTQ with column groups; the Hessian from seeded calibration inputs as the layer sees them
        solver = gu.GPTQ(lin)
        solver.quantizer = qu.WeightQuantizer()
        solver.quantizer.configure(w_bits, perchannel=True, sym=True, mse=False)
        for xc in calib_inputs(seed + 100, K_in):
            xin = torch.nn.functional.pad(xc, (0, K_pad - K_in)) if K_pad != K_in else xc
            if had:
                hadK, Kh = hu.get_hadK(K_pad)
                xin = hu.matmul_hadU_cuda(xin, hadK, Kh)
            solver.add_batch(xin.reshape(1, -1, K_pad), lin(xin))
        rec = []
        real_find = solver.quantizer.find_params

        def recording_find(x, _f=real_find, _q=solver.quantizer, _rec=rec):
            _f(x)
            _rec.append(_q.scale.reshape(-1).float().clone())
        solver.quantizer.find_params = recording_find
        actorder = tag.startswith("ao_")
        perm = torch.arange(K_pad)
        if actorder:                                              # the permutation the solver is about to take (gptq_utils.py:210-227)
            Hc = solver.H.clone()
            dead = torch.diag(Hc) == 0
            Hc[dead, dead] = 1
            perm = torch.argsort(torch.diag(Hc), descending=True)
        solver.fasterquant(percdamp=0.01, groupsize=g, actorder=actorder, static_groups=False)
        G = K_pad // g
        group_scales = torch.stack(rec[-G:], dim=1)               # the first recorded call is the whole-row warm-up of :208
        assert len(rec) in (G, G + 1) and torch.equal(group_scales[:, -1], solver.quantizer.scale.reshape(-1).float())
        Wq = lin.weight.data.float().clone()
        Wp = Wq[:, perm]                                          # the solver's column order: group j = columns perm[j g .. (j + 1) g - 1]
        lv = torch.round(Wp.reshape(N, G, g) / group_scales[:, :, None])
        assert float((lv * group_scales[:, :, None] - Wp.reshape(N, G, g)).abs().max()) < 1e-6 and float(lv.abs().max()) <= 2 ** (w_bits - 1)
        # ---- the reference's wrapper over the solved weights
        lin = lin.to(dt)
        wrap = qu.ActQuantWrapper(lin)
        if had:
            hadK, Kh = hu.get_hadK(K_pad)
            wrap.online_full_had, wrap.had_K, wrap.K = True, hadK, Kh
        if K_pad != K_in:
            wrap.register_forward_pre_hook(functools.partial(ru.revise_down_input, new_size=K_pad))
        out = {}
        if mode == "static":
            wrap.quantizer.configure(bits=8, sym=True, static=True, observer_type="minmax")
            qu.model_open_calibrate(wrap, Args())
            cal = [make_x(seed + 10 + i, (M, K_in)) for i in range(3)]
            for i, c in enumerate(cal):
                if i == len(cal) - 1:
                    qu.model_open_last_calibrate(wrap, Args())
                wrap(torch.from_numpy(c.copy()).to(dt))
            qu.model_close_calibrate(wrap, Args())
            qu.model_quant(wrap, Args())
            out["s_x"] = np.asarray(wrap.quantizer.quantizer.scale.numpy(), dtype=np.float32)
        elif mode == "dyn":
            wrap.quantizer.configure(bits=8, sym=True)
        else:
            wrap.quantizer.configure(bits=8, groupsize=g, sym=True, clip_ratio=1.0)
        shape = (1, M, K_in) if mode == "agrp" else (M, K_in)      # the reshape of quant_utils.py:183 needs 3 dims
        x = torch.from_numpy(make_x(seed + 20, shape)).to(dt)
        y = wrap(x.clone())
        # the reference's own activation quantizer on the tensor the Linear saw
        xt = x.clone()
        if K_pad != K_in:
            xt = torch.nn.functional.pad(xt, (0, K_pad - K_in))
        if had:
            xt = hu.matmul_hadU_cuda(xt, wrap.had_K, wrap.K)
        if mode == "static":
            qx = wrap.quantizer.quantizer.quant(xt.float()).reshape(M, K_pad).to(torch.int64)
        else:
            aq = qu.ActQuantizer()
            aq.configure(bits=8, groupsize=(g if mode == "agrp" else -1), sym=True, clip_ratio=1.0)
            aq.find_params(xt)
            q, scale = aq.quantize(xt)
            qx = q.reshape(M, K_pad).float().to(torch.int64)
            if mode == "agrp":
                out["s_x_groups"] = scale.reshape(M, G, g)[:, :, 0].float().numpy().astype(np.float32)
            else:
                out["s_x_rows"] = scale.reshape(M, K_pad)[:, 0].float().numpy().astype(np.float32)
        qx = qx[:, perm]                                          # activation levels in the solver's column order
        acc = torch.einsum("mgk,ngk->mgn", qx.reshape(M, G, g), lv.reshape(N, G, g).to(torch.int64))
        assert int(acc.abs().max()) < 2 ** 31
        if actorder:
            out["perm"] = perm.numpy().astype(np.int64)
        gen_golden.save(f"wrapper_wgrp_{tag}", y=y.float().numpy().reshape(M, N), W=Wq.numpy(),
                        group_scales=group_scales.numpy().astype(np.float32), qx=qx.numpy().astype(np.int8),
                        acc_groups=acc.numpy().astype(np.int32), mode=np.array(mode),
                        meta=np.array([K_in, K_pad, N, M, seed, int(had), int(bias), g, dtc, w_bits], np.int64), **out)
        print(tag, "max|y|", float(y.abs().max()), "group scale range", float(group_scales.min()), float(group_scales.max()))


if __name__ == "__main__":
    main()
