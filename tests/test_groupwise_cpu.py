"""Group-wise dynamic activation quantizer (--a_groupsize): the oracle restatement (oracle/mq_oracle.c
orc_quant_group) against goldens captured from the reference's own ActQuantizer on fp32 / fp16 / bf16 tensors
(tools/gen_golden_groupwise.py), and the simulated wrapper of this repository against the reference's forward."""
import functools
import glob
import os

import numpy as np
import pytest
import torch

import oracle
from golden_inputs import make_w, make_x

DT = {0: torch.float32, 1: torch.float16, 2: torch.bfloat16}


def cases(golden_dir):
    return sorted(glob.glob(os.path.join(golden_dir, "wrapper_grp_*.npz")))


def load(path):
    g = np.load(path)
    K_in, K_pad, N, M, seed, had, bias, a_bits, gs, dtc = [int(v) for v in g["meta"]]
    return g, dict(K_in=K_in, K_pad=K_pad, N=N, M=M, seed=seed, had=bool(had), bias=bool(bias), bits=a_bits, g=gs, dtc=dtc)


def rotated_input(c, had_table):
    x = torch.from_numpy(make_x(c["seed"] + 20, (c["M"], c["K_in"]))).to(DT[c["dtc"]]).float().numpy()
    if c["had"]:
        K = had_table["n2k"][c["K_pad"]]
        x = oracle.hadamard(x, c["K_pad"], K, had_table["mats"][K], mid_round=c["dtc"], out_round=c["dtc"])
    return x


def test_there_are_goldens(golden_dir):
    assert len(cases(golden_dir)) == 6


def test_oracle_group_quantizer_equals_the_reference(golden_dir, had_table):
    for path in cases(golden_dir):
        g, c = load(path)
        q, s = oracle.quant_group(rotated_input(c, had_table), c["g"], c["bits"], float(g["clip"]), c["dtc"])
        np.testing.assert_array_equal(s, g["s_groups"], err_msg=path)
        np.testing.assert_array_equal(q, g["qx"], err_msg=path)
        # per-group integer accumulators of the restated levels against the reference's
        lv = np.rint(make_w_levels(g, c))
        acc = np.einsum("mgk,ngk->mgn", q.reshape(c["M"], -1, c["g"]).astype(np.int64),
                        lv.reshape(c["N"], -1, c["g"]).astype(np.int64))
        np.testing.assert_array_equal(acc, g["acc_groups"], err_msg=path)


def make_w_levels(g, c):
    """int4 levels of the RTN-quantized weight the generator used (weights are created in fp32, cast to the dtype)."""
    w = torch.from_numpy(make_w(c["seed"], (c["N"], c["K_pad"]))).to(DT[c["dtc"]]).float().numpy()
    s_w, lv = oracle.wquant_sym(w, bits=4)
    np.testing.assert_array_equal(s_w, g["s_w"])
    return lv


def test_integer_restatement_reproduces_the_reference_output(golden_dir):
    """y = (sum_g acc_g * s_g) * s_w + bias -- what mq_gemm_w4a8_groupscale evaluates -- against the reference's
    floating-point evaluation: within 1e-3 of the output range (the reference rounds s * q to x's dtype first)."""
    for path in cases(golden_dir):
        g, c = load(path)
        y = (g["acc_groups"].astype(np.float32) * g["s_groups"][:, :, None]).sum(axis=1) * g["s_w"][None, :]
        if c["bias"]:
            y = y + torch.from_numpy(make_w(c["seed"] + 1, (c["N"],), std=0.1)).to(DT[c["dtc"]]).float().numpy()[None, :]
        tol = 1e-3 * float(np.abs(g["y"]).max()) * (8 if c["dtc"] == 2 else (2 if c["dtc"] == 1 else 1))
        np.testing.assert_allclose(y, g["y"], rtol=0, atol=tol, err_msg=path)


def test_simulated_wrapper_equals_the_reference_forward(golden_dir):
    """This repository's fake_quant on the CPU (simulated path, opt-in) runs the same arithmetic as the reference."""
    from fake_quant import hadamard_utils as hu, quant_utils as qu, utils
    from fake_quant.gptq.rtn import rtn_module
    torch.set_grad_enabled(False)
    for path in cases(golden_dir):
        g, c = load(path)
        dt = DT[c["dtc"]]
        lin = torch.nn.Linear(c["K_pad"], c["N"], bias=c["bias"])
        lin.weight.data = torch.from_numpy(make_w(c["seed"], (c["N"], c["K_pad"])))
        if c["bias"]:
            lin.bias.data = torch.from_numpy(make_w(c["seed"] + 1, (c["N"],), std=0.1))
        wrap = qu.ActQuantWrapper(lin.to(dt))
        if c["had"]:
            hadK, Kh = hu.get_hadK(c["K_pad"])
            wrap.online_full_had, wrap.had_K, wrap.K = True, hadK, Kh
        if c["K_pad"] != c["K_in"]:
            wrap.register_forward_pre_hook(functools.partial(utils.revise_down_input, new_size=c["K_pad"]))
        rtn_module(wrap, "layer", 4, True, False, [], {})
        wrap.real_quant = False
        wrap.simulate_on_cpu = True
        wrap.quantizer.configure(bits=c["bits"], groupsize=c["g"], sym=True, clip_ratio=float(g["clip"]))
        x = torch.from_numpy(make_x(c["seed"] + 20, (1, c["M"], c["K_in"]))).to(dt)
        y = wrap(x).float().numpy().reshape(c["M"], c["N"])
        tol = {0: 2e-5, 1: 4e-3, 2: 3e-2}[c["dtc"]] * float(np.abs(g["y"]).max())
        np.testing.assert_allclose(y, g["y"], rtol=0, atol=tol, err_msg=path)
