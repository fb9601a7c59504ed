"""GPU drop-in parity: the repo's ``fake_quant.quant_utils.ActQuantWrapper`` (real W4A8 kernels
underneath) against outputs of the reference's own ``ActQuantWrapper.forward`` captured by
tools/gen_golden.py (tests/golden/wrapper_*.npz).

The flow is the reference's: build the Linear -> wrap -> (online Hadamard / split / pad hook)
-> RTN weight pass -> configure static int8 -> calibration protocol -> quantized forward.
Bar: calibrated scale identical, int32 accumulators bit-exact, fp outputs within 1e-3
(BASELINE.json north_star) of the reference evaluated in fp32.
"""
import functools
import os

import numpy as np
import pytest
import torch

import oracle
from golden_inputs import make_w, make_x

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
torch.set_grad_enabled(False)

CASES = ["plain_3584", "plain_4096_mse", "plain_w8", "had_5120_split", "had_5120", "had_5120_fp32had",
         "had_11008", "had_8192", "down_19968", "down_19968_split", "had_14336"]


class Args:
    skip_names = []


def build_wrapper(g, case, dtype=torch.float32):
    from fake_quant import hadamard_utils as hu, quant_utils as qu, utils
    from fake_quant.gptq.rtn import rtn_module
    K_in, K_pad, N, M, seed, had, split, w_bits, w_mse, bias = [int(v) for v in g["meta"]]
    lin = torch.nn.Linear(K_pad, N, bias=bool(bias))
    lin.weight.data = torch.from_numpy(make_w(seed, (N, K_pad)))
    if bias:
        lin.bias.data = torch.from_numpy(make_w(seed + 1, (N,), std=0.1))
    lin = lin.to(device=DEV, dtype=dtype)
    wrap = qu.ActQuantWrapper(lin)
    if had:
        hadK, Kh = hu.get_hadK(K_pad)
        wrap.online_full_had, wrap.had_K, wrap.K = True, hadK, Kh
        wrap.fp32_had = case.endswith("fp32had")
    if split:
        wrap.split = True
        wrap.split_weights()
    if K_pad != K_in:
        wrap.register_forward_pre_hook(functools.partial(utils.revise_down_input, new_size=K_pad))
    quantizers = {}
    rtn_module(wrap, "layer", w_bits, True, bool(w_mse), [], quantizers)
    wrap.quantizer.configure(bits=8, sym=True, static=True, observer_type="minmax")
    return wrap, quantizers, (K_in, K_pad, N, M, seed, split)


@pytest.mark.parametrize("case", CASES)
def test_wrapper_matches_reference_forward(golden_dir, case):
    from fake_quant import quant_utils as qu
    g = np.load(os.path.join(golden_dir, f"wrapper_{case}.npz"))
    wrap, quantizers, (K_in, K_pad, N, M, seed, split) = build_wrapper(g, case)
    key = "layer.L2" if split else "layer.module"
    np.testing.assert_array_equal(quantizers[key].scale.float().numpy().reshape(-1), g["s_w"])
    batches = [torch.from_numpy(make_x(seed + 10 + i, (M, K_in))).to(DEV) for i in range(3)]
    qu.calib_layer(wrap, batches, Args())
    # calibrated per-tensor scale: identical to the reference's
    np.testing.assert_array_equal(np.asarray(wrap.quantizer.quantizer.scale.cpu().numpy(), np.float32), g["s_x"])
    x = torch.from_numpy(make_x(seed + 20, (M, K_in))).to(DEV)
    assert wrap._real_ready(x), "the real W4A8 backend must be the one that runs"
    y = wrap(x)
    assert wrap._real is not None
    np.testing.assert_allclose(y.cpu().numpy(), g["y"], rtol=0, atol=1e-3)
    # integer accumulators of the very kernels that ran
    real = wrap._real
    a, _ = real.quantize(x)
    from mquant_amd import ops
    acc = ops.gemm_w4a8_i32(a, real.w_img, real.w_bits, real.N).cpu().numpy()
    np.testing.assert_array_equal(acc, g["acc"])
    a_rows = a.to_rows() if isinstance(a, ops.TiledAct) else a       # the engine keeps activations tiled
    np.testing.assert_array_equal(a_rows.cpu().numpy()[:, 1 if split else 0:65 if split else 64], g["qx_head"])


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_wrapper_half_precision_matches_oracle(golden_dir, had_table, dtype):
    """fp16/bf16 (the deployment dtypes): wrapper == oracle bit for bit."""
    from fake_quant import quant_utils as qu
    case = "had_5120_split"
    g = np.load(os.path.join(golden_dir, f"wrapper_{case}.npz"))
    wrap, quantizers, (K_in, K_pad, N, M, seed, split) = build_wrapper(g, case, dtype)
    batches = [torch.from_numpy(make_x(seed + 10 + i, (M, K_in))).to(device=DEV, dtype=dtype) for i in range(3)]
    qu.calib_layer(wrap, batches, Args())
    x = torch.from_numpy(make_x(seed + 20, (M, K_in))).to(device=DEV, dtype=dtype)
    y = wrap(x)
    mode = 1 if dtype == torch.float16 else 2
    s_x = np.float32(float(wrap.quantizer.quantizer.scale))
    rot = oracle.hadamard(x.float().cpu().numpy(), K_pad, 40, had_table["mats"][40], mid_round=mode, out_round=mode)
    q = oracle.quant_static(rot, s_x)
    q[:, 0] = 0
    W = wrap.L2.weight.data.float().cpu().numpy()
    s_w = quantizers["layer.L2"].scale.float().numpy().reshape(-1)
    lv = np.concatenate([np.zeros((N, 1), np.int8), np.rint(W / s_w[:, None]).astype(np.int8)], axis=1)
    acc = oracle.gemm_i32(q, lv)
    ref = oracle.epilogue(acc, s_x, s_w, bias=wrap.L2.bias.data.float().cpu().numpy(),
                          x0=rot[:, 0], w0=wrap.L1.weight.data.float().cpu().numpy().reshape(-1))
    np.testing.assert_array_equal(y.float().cpu().numpy(), oracle.round_to(ref, mode))


def test_calibration_runs_the_observer_kernels_on_gpu(golden_dir):
    from fake_quant import quant_utils as qu
    g = np.load(os.path.join(golden_dir, "wrapper_plain_3584.npz"))
    wrap, _, (K_in, K_pad, N, M, seed, split) = build_wrapper(g, "plain_3584")
    qu.model_open_calibrate(wrap, Args())
    x = torch.from_numpy(make_x(seed + 10, (M, K_in))).to(DEV)
    y = wrap(x)
    np.testing.assert_allclose(y.cpu().numpy(), g["y_calib0"], rtol=0, atol=2e-5)   # unquantized flow
    ob = wrap.quantizer.observer
    assert float(ob.max_val) == float(x.max().clamp(min=0)) and float(ob.min_val) == float(x.min().clamp(max=0))


def test_w16_static_activation_path_uses_fused_fakequant(golden_dir):
    """Weights left in floating point: activation fake-quant (one fused kernel) + F.linear."""
    from fake_quant import quant_utils as qu
    lin = torch.nn.Linear(256, 32, bias=True).to(DEV).half()
    wrap = qu.ActQuantWrapper(lin)
    wrap.quantizer.configure(bits=8, sym=True, static=True)
    xs = [torch.from_numpy(make_x(i, (16, 256))).to(DEV).half() for i in range(2)]
    qu.calib_layer(wrap, xs, Args())
    x = torch.from_numpy(make_x(9, (16, 256))).to(DEV).half()
    assert not wrap._real_ready(x)
    y = wrap(x)
    s = np.float32(float(wrap.quantizer.quantizer.scale))
    xq = oracle.round_to(oracle.dequant_static(oracle.quant_static(x.float().cpu().numpy(), s), s), 1)
    ref = torch.nn.functional.linear(torch.from_numpy(xq).to(DEV).half(), lin.weight, lin.bias)
    torch.testing.assert_close(y, ref, rtol=0, atol=0)


def test_msq_two_scale_sets_selected_by_token_type_mask(had_table):
    """Modality-specific static quantization end to end through the wrapper."""
    from fake_quant import quant_utils as qu
    from fake_quant.gptq.rtn import rtn_module
    M, K, N = 48, 512, 64
    lin = torch.nn.Linear(K, N, bias=False)
    lin.weight.data = torch.from_numpy(make_w(3, (N, K)))
    wrap = qu.ActQuantWrapper(lin.to(DEV).half())
    quantizers = {}
    rtn_module(wrap, "l", 4, True, False, [], quantizers)
    wrap.quantizer.configure(bits=8, sym=True, static=True, msq=True)
    mask = torch.tensor([0] * 16 + [1] * 32, device=DEV)
    xs = []
    for i in range(2):
        x = make_x(40 + i, (M, K))
        x[:16] *= 8.0
        xs.append(torch.from_numpy(x).to(DEV).half())
    with qu.token_type_mask(mask):
        qu.calib_layer(wrap, xs, Args())
        x = xs[0]
        y = wrap(x)
    s0 = np.float32(float(wrap.quantizer.quantizer.scale))
    s1 = np.float32(float(wrap.quantizer.quantizer_text.scale))
    assert s0 > 4 * s1
    sel = mask.cpu().numpy().astype(np.uint8)
    q = oracle.quant_static(x.float().cpu().numpy(), s0, scale1=s1, row_sel=sel)
    W = wrap.module.weight.data.float().cpu().numpy()
    s_w = quantizers["l.module"].scale.float().numpy().reshape(-1)
    acc = oracle.gemm_i32(q, np.rint(W / s_w[:, None]).astype(np.int8))
    ref = oracle.round_to(oracle.epilogue(acc, s0, s_w, sx1=s1, row_sel=sel), 1)
    np.testing.assert_array_equal(y.float().cpu().numpy(), ref)


def test_conv3d_patch_embed_runs_as_gemm():
    """Qwen2-VL patch_embed.proj is a wrapped Conv3d whose kernel covers the whole patch."""
    from fake_quant import quant_utils as qu
    from fake_quant.gptq.rtn import rtn_wrapped_conv
    conv = torch.nn.Conv3d(3, 64, kernel_size=(2, 14, 14), stride=(2, 14, 14), bias=False)
    conv.weight.data = torch.from_numpy(make_w(5, tuple(conv.weight.shape)))
    wrap = qu.ActQuantWrapper(conv.to(DEV).half())
    quantizers = {}
    rtn_wrapped_conv(wrap, "pe", 4, True, False, quantizers)
    wrap.quantizer.configure(bits=8, sym=True, static=True)
    xs = [torch.from_numpy(make_x(60 + i, (32, 3, 2, 14, 14))).to(DEV).half() for i in range(2)]
    qu.calib_layer(wrap, xs, Args())
    x = xs[1]
    assert wrap._real_ready(x)
    y = wrap(x)
    assert y.shape == (32, 64, 1, 1, 1)
    s = np.float32(float(wrap.quantizer.quantizer.scale))
    q = oracle.quant_static(x.float().cpu().numpy().reshape(32, -1), s)
    W = wrap.module.weight.data.float().cpu().numpy().reshape(64, -1)
    s_w = quantizers["pe"].scale.float().numpy().reshape(-1)
    ref = oracle.round_to(oracle.epilogue(oracle.gemm_i32(q, np.rint(W / s_w[:, None]).astype(np.int8)), s, s_w), 1)
    np.testing.assert_array_equal(y.float().cpu().numpy().reshape(32, 64), ref)


# ---------------------------------------------------------------------------- asymmetric weights
from test_oracle_golden import WASYM_CASES  # noqa: E402


def test_rowsum_kernel_matches_the_integer_sum():
    from mquant_amd import ops
    rng = np.random.default_rng(3)
    for M, K in ((1, 128), (37, 1280), (768, 19968)):
        a = rng.integers(-128, 128, size=(M, K), dtype=np.int8)
        sel = (rng.random(M) < 0.5).astype(np.uint8)
        s_rows = rng.uniform(0.001, 0.1, size=M).astype(np.float32)
        want_static = np.where(sel != 0, np.float32(0.05), np.float32(0.02)) * a.astype(np.int64).sum(axis=1).astype(np.float32)
        want_rows = s_rows * a.astype(np.int64).sum(axis=1).astype(np.float32)
        for act in (torch.from_numpy(a).to(DEV), ops.TiledAct.from_rows(torch.from_numpy(a).to(DEV))):
            got = ops.act_rowsum_scaled(act, 0.02, 0.05, torch.from_numpy(sel).to(DEV))
            np.testing.assert_array_equal(got.cpu().numpy(), want_static)
            got = ops.act_rowsum_scaled(act, s_x_rows=torch.from_numpy(s_rows).to(DEV))
            np.testing.assert_array_equal(got.cpu().numpy(), want_rows)


@pytest.mark.parametrize("case", WASYM_CASES)
def test_wrapper_with_asymmetric_weights_matches_reference_forward(golden_dir, case):
    """--w_asym on the real-integer path: weights stored minus 2^(bits-1), zero points restored by the
    rank-1 epilogue term (mq_act_rowsum_scaled x s_w (2^(bits-1) - z_w))."""
    import functools
    from fake_quant import hadamard_utils as hu, quant_utils as qu, utils
    from fake_quant.gptq.rtn import rtn_module
    from golden_inputs import make_w, make_x
    from mquant_amd import ops
    g = np.load(os.path.join(golden_dir, f"wrapper_wasym_{case}.npz"))
    K_in, K_pad, N, M, seed, had, bias, w_bits, w_mse, dynamic = [int(v) for v in g["meta"]]
    lin = torch.nn.Linear(K_pad, N, bias=bool(bias))
    lin.weight.data = torch.from_numpy(make_w(seed, (N, K_pad)))
    if bias:
        lin.bias.data = torch.from_numpy(make_w(seed + 1, (N,), std=0.1))
    wrap = qu.ActQuantWrapper(lin.to(DEV))
    if had:
        hadK, Kh = hu.get_hadK(K_pad)
        wrap.online_full_had, wrap.had_K, wrap.K = True, hadK, Kh
    if K_pad != K_in:
        wrap.register_forward_pre_hook(functools.partial(utils.revise_down_input, new_size=K_pad))
    quantizers = {}
    rtn_module(wrap, "layer", w_bits, False, bool(w_mse), [], quantizers)
    wq = wrap.weight_quantizers["module"]
    np.testing.assert_array_equal(wq.scale.reshape(-1).cpu().numpy(), g["s_w"])
    np.testing.assert_array_equal(wq.zero.reshape(-1).cpu().numpy(), g["z_w"])
    if dynamic:
        wrap.quantizer.configure(bits=8, sym=True, clip_ratio=1.0)
    else:
        wrap.quantizer.configure(bits=8, sym=True, static=True, observer_type="minmax")

        class A:
            skip_names = []
        calib = [torch.from_numpy(make_x(seed + 10 + i, (M, K_in))).to(DEV) for i in range(3)]
        qu.calib_layer(wrap, calib, A())
        np.testing.assert_array_equal(np.float32(wrap.quantizer.quantizer.scale.cpu().numpy()), g["s_x"])
    x = torch.from_numpy(make_x(seed + 20, (M, K_in))).to(DEV)
    assert wrap._real_ready(x), "asymmetric weights must run the real kernels"
    y = wrap(x)
    real = wrap._real
    assert real is not None and real.w_shift is not None
    np.testing.assert_allclose(y.cpu().numpy(), g["y"], rtol=0, atol=1e-3)
    half = 1 << (w_bits - 1)
    np.testing.assert_array_equal(real.w_shift.cpu().numpy(), g["s_w"] * (np.float32(half) - g["z_w"]))
    # the integers of the kernels that ran
    xr = x if K_pad == K_in else torch.nn.functional.pad(x, (0, K_pad - K_in))
    xr = ops.hadamard(xr, real.had.n, real.had.K, real.had.bits) if had else xr
    if dynamic:
        a, _, _ = ops.quantize_act_dyn_i8(xr, 8, 1.0)
    else:
        a, _ = ops.quantize_act_i8(xr, float(g["s_x"]))
    np.testing.assert_array_equal(ops.gemm_w4a8_i32(a, real.w_img, w_bits, N).cpu().numpy(), g["acc"])
    # combinations that need the rank-1 slot twice run the integer kernels as well (tests/test_gpu_rank2.py)
    wrap2 = qu.ActQuantWrapper(torch.nn.Linear(256, 32).to(DEV))
    rtn_module(wrap2, "layer", 4, False, False, [], {})
    wrap2.quantizer.configure(bits=8, sym=False)
    assert wrap2._real_ready(torch.zeros(4, 256, device=DEV))
