"""Group-wise dynamic activation scales on the INTEGER path (--a_groupsize; reference quant_utils.py:181-203):
mq_quantize_act_group_i8 against the oracle (pinned to the reference on the CPU, tests/test_groupwise_cpu.py) bit for
bit, mq_gemm_w4a8_groupscale against its arithmetic restated in numpy bit for bit, and the ActQuantWrapper running
both against the reference's own forward (goldens from tools/gen_golden_groupwise.py)."""
import functools
import glob
import os

import numpy as np
import pytest
import torch

import oracle
from golden_inputs import make_w, make_x
from test_groupwise_cpu import DT, cases, load

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
MODE = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}
torch.set_grad_enabled(False)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,K,g,bits,clip,tiled", [(1, 128, 128, 8, 1.0, True), (37, 3584, 128, 8, 1.0, True),
                                                    (19, 1280, 256, 8, 0.9, False), (50, 2048, 64, 6, 1.0, True),
                                                    (7, 4096, 1024, 8, 0.95, False), (33, 1920, 64, 8, 1.0, True),
                                                    (5, 512, 16, 4, 1.0, False)])
def test_group_quantizer_kernel_equals_the_oracle(dtype, M, K, g, bits, clip, tiled):
    from mquant_amd import ops
    x = torch.from_numpy(make_x(M + K + g, (M, K))).to(DEV).to(dtype)
    x[0, :g] = 0                                                    # an all-zero group: scale 1
    if M > 2:
        x[2, g:2 * g] = x[2, g:2 * g].abs()                         # an all-positive group (no zero inclusion)
    a, s = ops.quantize_act_group_i8(x, g, bits, clip, tiled=tiled)
    want_q, want_s = oracle.quant_group(x.float().cpu().numpy(), g, bits, clip, MODE[dtype])
    np.testing.assert_array_equal(s.cpu().numpy(), want_s)
    rows = a.to_rows() if tiled else a
    np.testing.assert_array_equal(rows[:, :K].cpu().numpy(), want_q)
    assert not rows[:, K:].any() and rows.shape[1] % 128 == 0
    assert float(s[0, 0]) == 1.0


@pytest.mark.parametrize("w_bits,out_dtype", [(4, torch.float16), (8, torch.float32), (4, torch.bfloat16)])
@pytest.mark.parametrize("M,N,K,g", [(37, 200, 1280, 128), (130, 96, 2048, 64), (16, 48, 3584, 256), (300, 264, 1920, 64)])
def test_groupscale_gemm_equals_its_arithmetic_restated(w_bits, out_dtype, M, N, K, g):
    from mquant_amd import ops
    rng = np.random.default_rng(M + N + K)
    K_pad = (K + 127) // 128 * 128
    a = np.zeros((M, K_pad), np.int8)
    a[:, :K] = rng.integers(-128, 128, size=(M, K), dtype=np.int8)
    lim = 8 if w_bits == 4 else 128
    w = rng.integers(-lim, lim, size=(N, K), dtype=np.int8)
    wp = np.zeros((N, K_pad), np.int8)
    wp[:, :K] = w
    G = K // g
    s_g = (rng.random((M, G), dtype=np.float32) * 0.2 + 0.01).astype(np.float32)
    s_w = (rng.random(N, dtype=np.float32) * 0.01 + 0.001).astype(np.float32)
    bias = rng.normal(size=N).astype(np.float32)
    at = ops.TiledAct.from_rows(torch.from_numpy(a).to(DEV))
    img = ops.prepack(torch.from_numpy(wp).to(DEV), w_bits)
    y = ops.gemm_w4a8_groupscale(at, img, w_bits, N, torch.from_numpy(s_g).to(DEV), g, torch.from_numpy(s_w).to(DEV),
                                 bias=torch.from_numpy(bias).to(DEV), out_dtype=out_dtype)
    acc = np.einsum("mgk,ngk->mgn", a[:, :K].reshape(M, G, g).astype(np.int64), w.reshape(N, G, g).astype(np.int64))
    f = np.zeros((M, N), np.float32)
    for gi in range(G):                                              # ascending groups, one fp32 rounding per operation
        f = (f + (acc[:, gi, :].astype(np.float32) * s_g[:, gi:gi + 1]).astype(np.float32)).astype(np.float32)
    want = ((f * s_w[None, :]).astype(np.float32) + bias[None, :]).astype(np.float32)
    np.testing.assert_array_equal(y.float().cpu().numpy(), oracle.round_to(want, MODE[out_dtype]))
    # row-major activations take the same kernel
    y2 = ops.gemm_w4a8_groupscale(torch.from_numpy(a).to(DEV), img, w_bits, N, torch.from_numpy(s_g).to(DEV), g,
                                  torch.from_numpy(s_w).to(DEV), bias=torch.from_numpy(bias).to(DEV), out_dtype=out_dtype)
    assert torch.equal(y, y2)


def test_wrapper_runs_the_group_wise_mode_on_the_integer_path(golden_dir):
    from fake_quant import hadamard_utils as hu, quant_utils as qu, utils
    from fake_quant.gptq.rtn import rtn_module
    from mquant_amd import ops
    paths = cases(golden_dir)
    assert len(paths) == 6
    for path in paths:
        g, c = load(path)
        dt = DT[c["dtc"]]
        lin = torch.nn.Linear(c["K_pad"], c["N"], bias=c["bias"])
        lin.weight.data = torch.from_numpy(make_w(c["seed"], (c["N"], c["K_pad"])))
        if c["bias"]:
            lin.bias.data = torch.from_numpy(make_w(c["seed"] + 1, (c["N"],), std=0.1))
        wrap = qu.ActQuantWrapper(lin.to(dt).to(DEV))
        if c["had"]:
            hadK, Kh = hu.get_hadK(c["K_pad"])
            wrap.online_full_had, wrap.had_K, wrap.K = True, hadK, Kh
        if c["K_pad"] != c["K_in"]:
            wrap.register_forward_pre_hook(functools.partial(utils.revise_down_input, new_size=c["K_pad"]))
        rtn_module(wrap, "layer", 4, True, False, [], {})
        wrap.quantizer.configure(bits=c["bits"], groupsize=c["g"], sym=True, clip_ratio=float(g["clip"]))
        x = torch.from_numpy(make_x(c["seed"] + 20, (1, c["M"], c["K_in"]))).to(dt).to(DEV)
        assert wrap._real_ready(x), path
        y = wrap(x)
        real = wrap._real
        assert real is not None and real.dynamic["groupsize"] == c["g"]
        assert y.shape == (1, c["M"], c["N"]) and y.dtype == dt
        tol = {0: 1e-3, 1: 2e-3, 2: 8e-3}[c["dtc"]] * float(np.abs(g["y"]).max())
        np.testing.assert_allclose(y.float().cpu().numpy().reshape(c["M"], c["N"]), g["y"], rtol=0, atol=tol, err_msg=path)
        # the levels and scales the kernels produced are the reference quantizer's, bit for bit
        rows = x.reshape(c["M"], c["K_in"])
        xr = ops.hadamard(rows, real.had.n, real.had.K, real.had.bits) if c["had"] else rows
        a, s = ops.quantize_act_group_i8(xr, c["g"], c["bits"], float(g["clip"]))
        np.testing.assert_array_equal(s.cpu().numpy(), g["s_groups"], err_msg=path)
        np.testing.assert_array_equal(a[:, :c["K_pad"]].cpu().numpy(), g["qx"], err_msg=path)


def test_group_modes_outside_the_kernels_stay_simulated():
    from fake_quant import quant_utils as qu
    from fake_quant.gptq.rtn import rtn_module
    lin = torch.nn.Linear(384, 32).to(DEV).half()
    x = torch.from_numpy(make_x(1, (1, 8, 384))).to(DEV).half()
    for kw in (dict(bits=8, sym=True, groupsize=32),         # a group smaller than a k-tile of the GEMM
               dict(bits=8, sym=False, groupsize=32),
               dict(bits=8, sym=True, groupsize=256)):       # 384 is not a whole number of groups
        wrap = qu.ActQuantWrapper(lin)
        rtn_module(wrap, "l", 4, True, False, [], {})
        wrap.quantizer.configure(**kw)
        assert not wrap._real_ready(x), kw


# ---- asymmetric levels (--a_groupsize + --a_asym), round 4 ------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,K,g,bits,clip,tiled", [(1, 128, 128, 8, 1.0, True), (37, 3584, 128, 8, 1.0, True),
                                                    (19, 1280, 256, 8, 0.9, False), (50, 2048, 64, 6, 1.0, True),
                                                    (7, 4096, 1024, 8, 0.95, False), (5, 512, 16, 4, 1.0, False)])
def test_asymmetric_group_quantizer_kernel_equals_the_oracle(dtype, M, K, g, bits, clip, tiled):
    from mquant_amd import ops
    x = torch.from_numpy(make_x(M + K + g + 1, (M, K))).to(DEV).to(dtype)
    x[0, :g] = 0                                                    # an all-zero group: range (-1, +1)
    if M > 2:
        x[2, g:2 * g] = x[2, g:2 * g].abs() + 0.5                   # an all-positive group: zero point below 0
    a, s, z, sh = ops.quantize_act_group_asym_i8(x, g, bits, clip, tiled=tiled)
    want_q, want_s, want_z, want_sh = oracle.quant_group_asym(x.float().cpu().numpy(), g, bits, clip, MODE[dtype])
    np.testing.assert_array_equal(s.cpu().numpy(), want_s)
    np.testing.assert_array_equal(z.cpu().numpy(), want_z)
    np.testing.assert_array_equal(sh.cpu().numpy(), want_sh)
    rows = a.to_rows() if tiled else a
    np.testing.assert_array_equal(rows[:, :K].cpu().numpy(), want_q)
    assert not rows[:, K:].any() and rows.shape[1] % 128 == 0


@pytest.mark.parametrize("w_bits,out_dtype", [(4, torch.float16), (8, torch.float32), (4, torch.bfloat16)])
@pytest.mark.parametrize("M,N,K,g", [(37, 200, 1280, 128), (130, 96, 2048, 64), (16, 48, 3584, 256), (300, 264, 1920, 64),
                                     # K % 128 == 64 with groups of 64: K_pad ends in a zero-padded k-tile that belongs to no
                                     # group -- its constant term must not be added again for the last group (advisor r4)
                                     (33, 72, 192, 64), (140, 200, 1984, 64)])
def test_asymmetric_groupscale_gemm_equals_its_arithmetic_restated(w_bits, out_dtype, M, N, K, g):
    from mquant_amd import ops
    rng = np.random.default_rng(M + N + K + 7)
    K_pad = (K + 127) // 128 * 128
    a = np.zeros((M, K_pad), np.int8)
    a[:, :K] = rng.integers(-128, 128, size=(M, K), dtype=np.int8)
    lim = 8 if w_bits == 4 else 128
    w = rng.integers(-lim, lim, size=(N, K), dtype=np.int8)
    wp = np.zeros((N, K_pad), np.int8)
    wp[:, :K] = w
    G = K // g
    s_g = (rng.random((M, G), dtype=np.float32) * 0.2 + 0.01).astype(np.float32)
    sh_g = rng.normal(size=(M, G)).astype(np.float32)
    s_w = (rng.random(N, dtype=np.float32) * 0.01 + 0.001).astype(np.float32)
    bias = rng.normal(size=N).astype(np.float32)
    wsum = w.reshape(N, G, g).astype(np.int64).sum(axis=2).T.astype(np.float32).copy()          # [G][N]
    at = ops.TiledAct.from_rows(torch.from_numpy(a).to(DEV))
    img = ops.prepack(torch.from_numpy(wp).to(DEV), w_bits)
    y = ops.gemm_w4a8_groupscale_asym(at, img, w_bits, N, torch.from_numpy(s_g).to(DEV), torch.from_numpy(sh_g).to(DEV),
                                      torch.from_numpy(wsum).to(DEV), g, torch.from_numpy(s_w).to(DEV),
                                      bias=torch.from_numpy(bias).to(DEV), out_dtype=out_dtype)
    acc = np.einsum("mgk,ngk->mgn", a[:, :K].reshape(M, G, g).astype(np.int64), w.reshape(N, G, g).astype(np.int64))
    f = np.zeros((M, N), np.float32)
    for gi in range(G):                                              # ascending groups: + s_g acc_g, then + shift_g wsum_g, each rounded
        f = (f + (acc[:, gi, :].astype(np.float32) * s_g[:, gi:gi + 1]).astype(np.float32)).astype(np.float32)
        f = (f + (sh_g[:, gi:gi + 1] * wsum[gi][None, :]).astype(np.float32)).astype(np.float32)
    want = ((f * s_w[None, :]).astype(np.float32) + bias[None, :]).astype(np.float32)
    np.testing.assert_array_equal(y.float().cpu().numpy(), oracle.round_to(want, MODE[out_dtype]))


def test_wrapper_runs_the_asymmetric_group_wise_mode_on_the_integer_path(golden_dir):
    from fake_quant import hadamard_utils as hu, quant_utils as qu, utils
    from fake_quant.gptq.rtn import rtn_module
    from mquant_amd import ops
    from test_groupwise_asym_cpu import cases as acases, layer_input, load as aload
    paths = acases(golden_dir)
    assert len(paths) == 6
    for path in paths:
        g, c = aload(path)
        dt = DT[c["dtc"]]
        lin = torch.nn.Linear(c["K_pad"], c["N"], bias=c["bias"])
        lin.weight.data = torch.from_numpy(make_w(c["seed"], (c["N"], c["K_pad"])))
        if c["bias"]:
            lin.bias.data = torch.from_numpy(make_w(c["seed"] + 1, (c["N"],), std=0.1))
        wrap = qu.ActQuantWrapper(lin.to(dt).to(DEV))
        if c["had"]:
            hadK, Kh = hu.get_hadK(c["K_pad"])
            wrap.online_full_had, wrap.had_K, wrap.K = True, hadK, Kh
        if c["K_pad"] != c["K_in"]:
            wrap.register_forward_pre_hook(functools.partial(utils.revise_down_input, new_size=c["K_pad"]))
        rtn_module(wrap, "layer", 4, True, False, [], {})
        wrap.quantizer.configure(bits=c["bits"], groupsize=c["g"], sym=False, clip_ratio=float(g["clip"]))
        x = layer_input(c).to(DEV)
        assert wrap._real_ready(x), path
        y = wrap(x)
        real = wrap._real
        assert real is not None and real.dynamic["groupsize"] == c["g"] and real.wsum_groups is not None
        np.testing.assert_array_equal(real.wsum_groups.cpu().numpy(), g["wsum_groups"].astype(np.float32), err_msg=path)
        tol = {0: 1e-3, 1: 2e-3, 2: 8e-3}[c["dtc"]] * float(np.abs(g["y"]).max())
        np.testing.assert_allclose(y.float().cpu().numpy().reshape(c["M"], c["N"]), g["y"], rtol=0, atol=tol, err_msg=path)
        rows = x.reshape(c["M"], c["K_in"])
        rows = torch.nn.functional.pad(rows, (0, c["K_pad"] - c["K_in"])) if c["K_pad"] != c["K_in"] else rows
        xr = ops.hadamard(rows, real.had.n, real.had.K, real.had.bits) if c["had"] else rows
        a, s, z, _ = ops.quantize_act_group_asym_i8(xr, c["g"], c["bits"], float(g["clip"]))
        np.testing.assert_array_equal(s.cpu().numpy(), g["s_groups"], err_msg=path)
        np.testing.assert_array_equal(z.cpu().numpy(), g["z_groups"], err_msg=path)
        np.testing.assert_array_equal(a[:, :c["K_pad"]].cpu().numpy(), g["qx"], err_msg=path)


@pytest.mark.parametrize("K", [192, 1984])
def test_wrapper_asymmetric_groups_of_64_with_a_padded_last_k_tile(K):
    """K % 128 == 64 with groups of 64 (advisor finding r4): the integer path pads K to a multiple of 128, the pad tile belongs
    to no group.  The wrapper's integer result has to stay on the simulated evaluation of the same wrapper."""
    from fake_quant import quant_utils as qu
    from fake_quant.gptq.rtn import rtn_module
    M, N = 45, 136
    lin = torch.nn.Linear(K, N, bias=True)
    lin.weight.data = torch.from_numpy(make_w(K, (N, K)))
    lin.bias.data = torch.from_numpy(make_w(K + 1, (N,), std=0.1))
    wrap = qu.ActQuantWrapper(lin.to(DEV))
    rtn_module(wrap, "layer", 4, True, False, [], {})
    wrap.quantizer.configure(bits=8, groupsize=64, sym=False, clip_ratio=1.0)
    x = (torch.from_numpy(make_x(K + 3, (M, K))) + 0.7).to(DEV)        # off-centre: the zero points matter
    assert wrap._real_ready(x)
    y = wrap(x)
    assert wrap._real is not None and wrap._real.wsum_groups is not None
    wrap.real_quant = False
    y_sim = wrap(x.clone())
    tol = 1e-3 * float(y_sim.abs().max())
    np.testing.assert_allclose(y.cpu().numpy(), y_sim.cpu().numpy(), rtol=0, atol=tol)
