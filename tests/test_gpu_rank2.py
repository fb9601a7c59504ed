"""Two rank-1 epilogue terms on the GPU (mq_gemm_w4a8_rank2_ws): every kernel family against the oracle bit for bit, and the
ActQuantWrapper in the flag combinations that need both slots (--w_asym + --visual_split, --w_asym + --a_asym, --a_asym +
--visual_split) against the REFERENCE's own forward (tests/golden/wrapper_rank2_*.npz, tools/gen_golden_rank2.py): the integer
path must be the one that runs, scales / zero points / int32 accumulators bit for bit, outputs within 1e-3."""
import functools
import os

import numpy as np
import pytest
import torch

import oracle
from golden_inputs import make_w, make_x
from test_rank2_cpu import ACT, rank1_terms, rank2_cases

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
MODE = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}
torch.set_grad_enabled(False)


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    return t if dtype is None else t.to(dtype)


@pytest.mark.parametrize("tile", [-1, 40, 41, 42, 43, 44, 45, 46, 47, 51, 53, 54, 14, 20, 15, 3, 26, 10])
@pytest.mark.parametrize("rowscale", [False, True])
def test_rank2_epilogue_equals_the_oracle_on_every_kernel_family(tile, rowscale):
    from mquant_amd import ops
    rng = np.random.default_rng(100 + tile)
    ops.splitk_workspace(torch.device(DEV), 64 << 20)
    try:
        for (M, N, K, splits, out_dtype) in ((300, 520, 1408, 1, torch.float16), (77, 264, 640, 2, torch.float32), (515, 136, 2048, 1, torch.bfloat16)):
            a = rng.integers(-128, 128, size=(M, K), dtype=np.int8)
            w = rng.integers(-8, 8, size=(N, K), dtype=np.int8)
            s_w = rng.uniform(0.001, 0.01, size=N).astype(np.float32)
            bias = rng.normal(size=N).astype(np.float32)
            x0, x1 = rng.normal(size=M).astype(np.float32), rng.normal(size=M).astype(np.float32)
            w0, w1 = rng.normal(size=N).astype(np.float32), rng.normal(size=N).astype(np.float32)
            s_rows = rng.uniform(0.01, 0.05, size=M).astype(np.float32)
            sel = (rng.random(M) < 0.4).astype(np.uint8)
            acc = oracle.gemm_i32(a, w)
            if rowscale:
                want = oracle.epilogue(acc, s_rows, s_w, bias=bias, x0=x0, w0=w0, x1=x1, w1=w1)
            else:
                want = oracle.epilogue(acc, np.float32(0.02), s_w, bias=bias, sx1=np.float32(0.05), row_sel=sel, x0=x0, w0=w0, x1=x1, w1=w1)
            want = oracle.round_to(want, MODE[out_dtype])
            at = ops.TiledAct.from_rows(dev(a))
            img = ops.prepack(dev(w), 4)
            ops.gemm_debug_force(tile, splits)
            kw = dict(s_x_rows=dev(s_rows)) if rowscale else dict(s_x0=0.02, s_x1=0.05, row_sel=dev(sel))
            y = ops.gemm_w4a8_rank2(at, img, 4, N, dev(s_w), dev(x0), dev(w0), dev(x1), dev(w1), bias=dev(bias), out_dtype=out_dtype, **kw)
            np.testing.assert_array_equal(y.float().cpu().numpy(), want, err_msg=f"tile {tile} {M}x{N}x{K} splits {splits}")
    finally:
        ops.gemm_debug_force(-1, 0)


@pytest.mark.parametrize("path", rank2_cases(os.path.join(os.path.dirname(__file__), "golden")))
def test_wrapper_runs_the_two_slot_combinations_on_the_integer_path(path):
    from fake_quant import hadamard_utils as hu, quant_utils as qu, utils
    from mquant_amd import ops
    g = np.load(path)
    K_in, K_pad, N, M, seed, had, split, bias, w_sym, act = [int(v) for v in g["meta"]]
    act = ACT[act]
    lin = torch.nn.Linear(K_pad, N, bias=bool(bias))
    lin.weight.data = torch.from_numpy(make_w(seed, (N, K_pad)))
    if bias:
        lin.bias.data = torch.from_numpy(make_w(seed + 1, (N,), std=0.1))
    wrap = qu.ActQuantWrapper(lin.to(DEV))
    if had:
        hadK, Kh = hu.get_hadK(K_pad)
        wrap.online_full_had, wrap.had_K, wrap.K = True, hadK, Kh
    if split:
        wrap.split = True
        wrap.split_weights()
    if K_pad != K_in:
        wrap.register_forward_pre_hook(functools.partial(utils.revise_down_input, new_size=K_pad))
    # the weight pass: the repository's RTN driver (attaches the WeightQuantizer to the wrapper)
    from fake_quant.gptq.rtn import rtn_module
    quantizers = {}
    rtn_module(wrap, "layer", 4, bool(w_sym), False, [], quantizers)
    wq = quantizers["layer.L2" if split else "layer.module"]
    np.testing.assert_array_equal(wq.scale.float().cpu().numpy().reshape(-1), g["s_w"])
    if not w_sym:
        np.testing.assert_array_equal(wq.zero.float().cpu().numpy().reshape(-1), g["z_w"])

    class Args:
        skip_names = []
    if act == "static":
        wrap.quantizer.configure(bits=8, sym=True, static=True, observer_type="minmax")
        qu.calib_layer(wrap, [dev(make_x(seed + 10 + i, (M, K_in))) for i in range(3)], Args())
        np.testing.assert_array_equal(np.asarray(wrap.quantizer.quantizer.scale.cpu().numpy(), np.float32), g["s_x"])
    else:
        wrap.quantizer.configure(bits=8, sym=act == "dyn_sym", clip_ratio=1.0, act_per_tensor=act == "pt_asym")
    x = dev(make_x(seed + 20, (M, K_in)))
    assert wrap._real_ready(x), "this combination must run the real kernels"
    y = wrap(x)
    real = wrap._real
    assert real is not None
    n_terms = int(real.split) + int(real.w_shift is not None) + int(real.w_colsum is not None)
    assert n_terms == int(bool(split)) + int(not w_sym) + int(act in ("dyn_asym", "pt_asym")) and n_terms in (2, 3), path
    np.testing.assert_allclose(y.cpu().numpy(), g["y"], rtol=0, atol=1e-3)
    # the integers of the kernels that ran
    xp = torch.nn.functional.pad(x, (0, K_pad - K_in)) if K_pad != K_in else x
    xr = ops.hadamard(xp, real.had.n, real.had.K, real.had.bits) if had else xp
    if act == "static":
        a, _ = real.quantize(xp)
        lv = a.to_rows()[:, 1 if split else 0:K_pad]
    else:
        xq = xr[:, 1:] if split else xr
        if act == "dyn_sym":
            lvq, s_rows, _ = ops.quantize_act_dyn_i8(xq)
        elif act == "dyn_asym":
            lvq, s_rows, zero, _ = ops.quantize_act_dyn_asym_i8(xq)
        else:
            lvq, s_rows, zero, _, _ = ops.quantize_act_tensor_i8(xq, asym=True)
        np.testing.assert_array_equal(s_rows.cpu().numpy(), g["s_rows"])
        if act != "dyn_sym":
            np.testing.assert_array_equal(zero.cpu().numpy(), g["zero"])
        lv = lvq[:, :xq.shape[1]]
    np.testing.assert_array_equal(lv[:, :64].cpu().numpy(), g["qx_head"])
    np.testing.assert_array_equal(lv.to(torch.int64).sum(dim=1).cpu().numpy(), g["qx_sum"])


@pytest.mark.parametrize("out_dtype", [torch.float16, torch.bfloat16, torch.float32])
def test_third_rank1_term_behind_an_fp32_gemm_equals_the_oracle(out_dtype):
    """mq_rank1_add_cast: cast(y32 + x2[m] * w2[n]) continues the epilogue's fp32 sum and rounds once."""
    from mquant_amd import ops
    rng = np.random.default_rng(7)
    M, N, K = 130, 264, 640
    a = rng.integers(-128, 128, size=(M, K), dtype=np.int8)
    w = rng.integers(-8, 8, size=(N, K), dtype=np.int8)
    s_w = rng.uniform(0.001, 0.01, size=N).astype(np.float32)
    bias = rng.normal(size=N).astype(np.float32)
    xs = [rng.normal(size=M).astype(np.float32) for _ in range(3)]
    ws = [rng.normal(size=N).astype(np.float32) for _ in range(3)]
    s_rows = rng.uniform(0.01, 0.05, size=M).astype(np.float32)
    want = oracle.epilogue(oracle.gemm_i32(a, w), s_rows, s_w, bias=bias, x0=xs[0], w0=ws[0], x1=xs[1], w1=ws[1])
    want = (want + (xs[2].reshape(-1, 1) * ws[2][None, :]).astype(np.float32)).astype(np.float32)
    at = ops.TiledAct.from_rows(dev(a))
    img = ops.prepack(dev(w), 4)
    y32 = ops.gemm_w4a8_rank2(at, img, 4, N, dev(s_w), dev(xs[0]), dev(ws[0]), dev(xs[1]), dev(ws[1]), s_x_rows=dev(s_rows), bias=dev(bias),
                              out_dtype=torch.float32)
    y = ops.rank1_add_cast(y32, dev(xs[2]), dev(ws[2]), out_dtype)
    np.testing.assert_array_equal(y.float().cpu().numpy(), oracle.round_to(want, MODE[out_dtype]))
