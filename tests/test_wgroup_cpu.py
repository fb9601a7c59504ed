"""Group-wise WEIGHT scales (--w_groupsize; reference exam/quant_qwen2vl.py:327 -> gptq/gptq_utils.py:263-273) on the CPU:
the oracle's restatement of the integer evaluation against the reference's own GPTQ + ActQuantWrapper.forward
(tests/golden/wrapper_wgrp_*.npz, tools/gen_golden_wgroup.py), and this repository's solver against the reference's weights
and per-group scales bit for bit."""
import glob
import os

import numpy as np
import pytest
import torch

import oracle
from golden_inputs import make_w, make_x

torch.set_grad_enabled(False)
DT = {0: torch.float32, 1: torch.float16}
TOL = {0: 1e-3, 1: 2e-3}          # of max |y|: the reference's half-precision forward rounds its output (and the GEMM inputs) to fp16


def cases(golden_dir):
    return sorted(glob.glob(os.path.join(golden_dir, "wrapper_wgrp_*.npz")))


def load(path):
    g = np.load(path)
    K_in, K_pad, N, M, seed, had, bias, gs, dtc, w_bits = [int(v) for v in g["meta"]]
    return g, dict(K_in=K_in, K_pad=K_pad, N=N, M=M, seed=seed, had=bool(had), bias=bool(bias), g=gs, dtc=dtc, w_bits=w_bits,
                   mode=str(g["mode"]))


def levels_of(g, c):
    """The levels in the SOLVER's column order (--act_order cases: columns gathered by ``perm``; ``qx`` is stored in that order)."""
    G = c["K_pad"] // c["g"]
    W = g["W"][:, g["perm"]] if "perm" in g else g["W"]
    lv = np.rint(W.reshape(c["N"], G, c["g"]) / g["group_scales"][:, :, None]).astype(np.int8)
    return lv.reshape(c["N"], c["K_pad"])


def test_fixture_set_is_complete(golden_dir):
    assert len(cases(golden_dir)) == 11 and sum("wrapper_wgrp_ao_" in p for p in cases(golden_dir)) == 3


def test_oracle_equals_the_reference_forward(golden_dir):
    for path in cases(golden_dir):
        g, c = load(path)
        lv = levels_of(g, c)
        bias = make_w(c["seed"] + 1, (c["N"],), std=0.1) if c["bias"] else None
        if bias is not None and c["dtc"] == 1:
            bias = torch.from_numpy(bias).half().float().numpy()
        kw = {}
        if c["mode"] == "static":
            kw["sx0"] = float(g["s_x"])
        elif c["mode"] == "dyn":
            kw["sx_rows"] = g["s_x_rows"]
        else:
            kw["s_xg"] = g["s_x_groups"]
        y, acc = oracle.gemm_wgroup(g["qx"], lv, g["group_scales"].T.copy(), c["g"], bias=bias, want_acc=True, **kw)
        np.testing.assert_array_equal(acc, g["acc_groups"], err_msg=path)                 # exact integers inside every group
        tol = TOL[c["dtc"]] * float(np.abs(g["y"]).max())
        np.testing.assert_allclose(y, g["y"], rtol=0, atol=tol, err_msg=path)


def test_this_repositorys_solver_keeps_every_groups_scale_and_equals_the_reference(golden_dir, had_table):
    from fake_quant import quant_utils as qu
    from fake_quant.gptq.gptq_utils import GPTQ
    for path in cases(golden_dir):
        g, c = load(path)
        lin = torch.nn.Linear(c["K_pad"], c["N"], bias=c["bias"])
        lin.weight.data = torch.from_numpy(make_w(c["seed"], (c["N"], c["K_pad"]))) * 4.0
        solver = GPTQ(lin)
        solver.quantizer = qu.WeightQuantizer()
        solver.quantizer.configure(c["w_bits"], perchannel=True, sym=True, mse=False)
        for i in range(3):
            xc = torch.from_numpy(make_x(c["seed"] + 100 + 1 + i, (40, c["K_in"])))
            xin = torch.nn.functional.pad(xc, (0, c["K_pad"] - c["K_in"])) if c["K_pad"] != c["K_in"] else xc
            if c["had"]:        # the rotated calibration input, by the oracle's restatement of matmul_hadU_cuda (pinned bit-exact)
                Kh = had_table["n2k"][c["K_pad"]]
                xin = torch.from_numpy(oracle.hadamard(xin.numpy(), c["K_pad"], Kh, had_table["mats"][Kh]))
            solver.add_batch(xin.reshape(1, -1, c["K_pad"]))
        actorder = "perm" in g
        solver.fasterquant(percdamp=0.01, groupsize=c["g"], actorder=actorder, static_groups=False)
        qz = solver.quantizer
        assert qz.groupsize == c["g"] and qz.group_permuted == actorder
        if actorder:                                   # the reference's permutation, kept with the scales
            np.testing.assert_array_equal(qz.group_perm.numpy(), g["perm"], err_msg=path)
        np.testing.assert_array_equal(qz.group_scales.numpy(), g["group_scales"], err_msg=path)
        np.testing.assert_array_equal(lin.weight.data.numpy(), g["W"], err_msg=path)


def test_backend_line_names_the_integer_path_or_the_reason():
    from fake_quant import quant_utils as qu
    lin = torch.nn.Linear(256, 16)
    wrap = qu.ActQuantWrapper(lin)
    assert "float" in wrap.backend() and "Backend:" in wrap.extra_repr()
    wrap.quantizer.configure(bits=8, sym=True)
    assert "no weight quantizer attached" in wrap.backend()
    wq = qu.WeightQuantizer()
    wq.configure(4, perchannel=True, sym=True)
    wq.find_params(lin.weight.data)
    qu.attach_weight_quantizer(wrap, "module", wq)
    assert wrap.backend() == "W4A8 integer (dynamic)"
    # group scales recorded: groups of 128 run, groups of 32 do not, asymmetric activations do not
    wq.groupsize, wq.group_permuted, wq.group_scales = 128, False, torch.ones(16, 2)
    assert wrap.backend() == "W4A8 integer (dynamic, weight groups of 128)"
    wq.groupsize, wq.group_scales = 32, torch.ones(16, 8)
    assert "weight group size 32" in wrap.backend()
    wq.groupsize, wq.group_scales = 128, torch.ones(16, 2)
    wrap.quantizer.configure(bits=8, sym=False)
    assert "asymmetric or per-tensor" in wrap.backend()
    wrap.quantizer.configure(bits=8, sym=True, groupsize=64)
    assert "differs" in wrap.backend()
    wrap.quantizer.configure(bits=8, sym=True, groupsize=128)
    assert wrap.backend().startswith("W4A8 integer")
    wrap.real_quant = False
    assert "switched off" in wrap.backend()


def test_group_wise_rtn_leaves_what_the_gptq_solver_leaves():
    """fake_quant.gptq.rtn.rtn_module(groupsize=g) -- the extension synthetic benchmarks use -- runs find_params + quantize on every
    group of g columns and records group_scales / group_zeros / groupsize like this repository's GPTQ; a Linear whose width is not
    a multiple of g (the split fc2's L2) keeps per-channel scales."""
    from fake_quant import quant_utils as qu
    from fake_quant.gptq.rtn import rtn_module

    torch.manual_seed(3)
    root = torch.nn.Module()
    root.a = torch.nn.Linear(256, 24, bias=False)
    root.b = torch.nn.Linear(255, 8, bias=False)
    w0 = root.a.weight.data.clone()
    qu.add_actquant(root)
    assert isinstance(root.a, qu.ActQuantWrapper)
    quantizers = {}
    rtn_module(root, "m", 4, True, False, [], quantizers, groupsize=64)
    qz = quantizers["m.a.module"]
    assert qz.groupsize == 64 and not qz.group_permuted and tuple(qz.group_scales.shape) == (24, 4)
    ref = qu.WeightQuantizer()
    ref.configure(4, perchannel=True, sym=True, mse=False)
    for gi in range(4):
        ref.find_params(w0[:, gi * 64:(gi + 1) * 64])
        assert torch.equal(ref.scale.reshape(-1).float(), qz.group_scales[:, gi])
        assert torch.equal(ref.quantize(w0[:, gi * 64:(gi + 1) * 64]), root.a.module.weight.data[:, gi * 64:(gi + 1) * 64])
    assert getattr(quantizers["m.b.module"], "group_scales", None) is None           # 255 columns: per-channel scales
    root.a.quantizer.configure(bits=8, sym=True)                                      # dynamic per-token activations: nothing to calibrate
    assert "weight groups of 64" in root.a.extra_repr(), root.a.extra_repr()
