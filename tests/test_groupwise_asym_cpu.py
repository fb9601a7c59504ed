"""Group-wise dynamic activation quantizer with ASYMMETRIC levels (--a_groupsize + --a_asym): the oracle restatement
(oracle/mq_oracle.c orc_quant_group_asym) against goldens captured from the reference's own ActQuantizer on fp32 / fp16 / bf16
tensors (tools/gen_golden_groupwise_asym.py), and the integer evaluation the kernels use -- exact int32 sums inside a group,
y = (sum_g (s_g acc_g + shift_g wsum_g)) * s_w + bias -- against the reference's forward."""
import glob
import os

import numpy as np
import torch

import oracle
from golden_inputs import make_w, make_x

DT = {0: torch.float32, 1: torch.float16, 2: torch.bfloat16}


def cases(golden_dir):
    return sorted(glob.glob(os.path.join(golden_dir, "wrapper_grpa_*.npz")))


def load(path):
    g = np.load(path)
    K_in, K_pad, N, M, seed, had, bias, a_bits, gs, dtc = [int(v) for v in g["meta"]]
    return g, dict(K_in=K_in, K_pad=K_pad, N=N, M=M, seed=seed, had=bool(had), bias=bool(bias), bits=a_bits, g=gs, dtc=dtc)


def layer_input(c):
    """The generator's input: seeded activations with one all-positive and one all-zero group (torch tensor in the case's dtype)."""
    x = torch.from_numpy(make_x(c["seed"] + 20, (1, c["M"], c["K_in"]))).to(DT[c["dtc"]])
    if c["M"] > 2:
        w = min(c["g"], c["K_in"])
        x[0, 2, :w] = x[0, 2, :w].abs() + 0.5
        x[0, 1, :w] = 0
    return x


def rotated_input(c, had_table):
    x = layer_input(c).float().numpy().reshape(c["M"], c["K_in"])
    if c["had"]:
        K = had_table["n2k"][c["K_pad"]]
        x = oracle.hadamard(x, c["K_pad"], K, had_table["mats"][K], mid_round=c["dtc"], out_round=c["dtc"])
    return x


def test_there_are_goldens(golden_dir):
    assert len(cases(golden_dir)) == 6


def test_oracle_asymmetric_group_quantizer_equals_the_reference(golden_dir, had_table):
    for path in cases(golden_dir):
        g, c = load(path)
        q, s, z, _ = oracle.quant_group_asym(rotated_input(c, had_table), c["g"], c["bits"], float(g["clip"]), c["dtc"])
        np.testing.assert_array_equal(s, g["s_groups"], err_msg=path)
        np.testing.assert_array_equal(z, g["z_groups"], err_msg=path)
        np.testing.assert_array_equal(q, g["qx"], err_msg=path)


def test_integer_restatement_reproduces_the_reference_output(golden_dir):
    for path in cases(golden_dir):
        g, c = load(path)
        half = np.float32(1 << (c["bits"] - 1))
        shift = (g["s_groups"] * (half - g["z_groups"])).astype(np.float32)
        f = (g["acc_groups"].astype(np.float32) * g["s_groups"][:, :, None]).sum(axis=1)
        f = f + (shift[:, :, None] * g["wsum_groups"].astype(np.float32)[None, :, :]).sum(axis=1)
        y = f * g["s_w"][None, :]
        if c["bias"]:
            y = y + torch.from_numpy(make_w(c["seed"] + 1, (c["N"],), std=0.1)).to(DT[c["dtc"]]).float().numpy()[None, :]
        tol = 1e-3 * float(np.abs(g["y"]).max()) * (8 if c["dtc"] == 2 else (2 if c["dtc"] == 1 else 1))
        np.testing.assert_allclose(y, g["y"], rtol=0, atol=tol, err_msg=path)
