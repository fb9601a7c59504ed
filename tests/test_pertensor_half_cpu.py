"""Dynamic PER-TENSOR activation quantizer on half-precision activations (act_per_tensor; reference
fake_quant/quant_utils.py:214-237 keeps range, scale, zero point, x / scale and the level sum in x's dtype): the oracle
restatement (orc_quant_tensor, mode = dtype) against goldens captured from the reference's own ActQuantizer on fp16 /
bf16 tensors (tools/gen_golden_pertensor_half.py), and this repository's simulated wrapper against the reference's
forward."""
import functools
import glob
import os

import numpy as np
import torch

import oracle
from golden_inputs import make_w, make_x

DT = {1: torch.float16, 2: torch.bfloat16}


def cases(golden_dir):
    return sorted(glob.glob(os.path.join(golden_dir, "wrapper_dynpt16_*.npz")))


def load(path):
    g = np.load(path)
    K_in, K_pad, N, M, seed, had, split, bias, a_bits, sym, dtc = [int(v) for v in g["meta"]]
    return g, dict(K_in=K_in, K_pad=K_pad, N=N, M=M, seed=seed, had=bool(had), split=bool(split), bias=bool(bias),
                   bits=a_bits, sym=bool(sym), dtc=dtc)


def rotated_input(c, had_table):
    x = torch.from_numpy(make_x(c["seed"] + 20, (c["M"], c["K_in"]))).to(DT[c["dtc"]]).float().numpy()
    if c["K_pad"] != c["K_in"]:
        x = np.pad(x, ((0, 0), (0, c["K_pad"] - c["K_in"])))
    if c["had"]:
        K = had_table["n2k"][c["K_pad"]]
        x = oracle.hadamard(x, c["K_pad"], K, had_table["mats"][K], mid_round=c["dtc"], out_round=c["dtc"])
    return x


def test_there_are_goldens(golden_dir):
    assert len(cases(golden_dir)) == 5


def test_oracle_per_tensor_quantizer_equals_the_reference_on_half_tensors(golden_dir, had_table):
    for path in cases(golden_dir):
        g, c = load(path)
        x = rotated_input(c, had_table)
        q, s, z, _ = oracle.quant_tensor(x, c["bits"], float(g["clip"]), asym=not c["sym"], skip_col0=c["split"], mode=c["dtc"])
        assert s == g["scale"] and z == g["zero"], (path, s, float(g["scale"]), z, float(g["zero"]))
        lv = q[:, 1:] if c["split"] else q
        np.testing.assert_array_equal(lv, g["qx"], err_msg=path)
        # and an fp32 evaluation of the same rule is NOT the reference's grid (why the kernels carry the dtype)
        q32, s32, _, _ = oracle.quant_tensor(x, c["bits"], float(g["clip"]), asym=not c["sym"], skip_col0=c["split"], mode=0)
        assert s32 != g["scale"] or not np.array_equal(q32[:, 1:] if c["split"] else q32, g["qx"]), path


def test_simulated_wrapper_equals_the_reference_forward(golden_dir):
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
        if c["split"]:
            wrap.split = True
            wrap.split_weights()
        if c["K_pad"] != c["K_in"]:
            wrap.register_forward_pre_hook(functools.partial(utils.revise_down_input, new_size=c["K_pad"]))
        rtn_module(wrap, "layer", 4, True, False, [], {})
        wrap.real_quant = False
        wrap.simulate_on_cpu = True
        wrap.quantizer.configure(bits=c["bits"], sym=c["sym"], clip_ratio=float(g["clip"]), act_per_tensor=True)
        x = torch.from_numpy(make_x(c["seed"] + 20, (c["M"], c["K_in"]))).to(dt)
        y = wrap(x).float().numpy()
        # half-precision Linear over K up to 19968 on both sides; the rotated cases also differ in the Hadamard's rounding
        # (the generator's shim multiplies by the dense matrix) and may flip a level
        tol = {1: 8e-3 if c["had"] else 4e-3, 2: 3e-2}[c["dtc"]] * float(np.abs(g["y"]).max())
        np.testing.assert_allclose(y, g["y"], rtol=0, atol=tol, err_msg=path)
