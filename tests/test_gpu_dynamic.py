"""Dynamic per-token activation mode on the real-integer path: mq_quantize_act_dyn_i8 +
mq_gemm_w4a8_rowscale_ws against the oracle (bit-exact) and the wrapper against the reference's
own forward in that mode (tests/golden/wrapper_dyn_*.npz)."""
import functools
import os

import numpy as np
import pytest
import torch

import oracle
from golden_inputs import make_w, make_x
from test_oracle_golden import ASYM_DYN_CASES, DYN_CASES, PT_DYN_CASES

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
torch.set_grad_enabled(False)


@pytest.mark.parametrize("M,K,dtype,bits,clip,skip", [(768, 3584, torch.float16, 8, 1.0, False), (33, 1000, torch.float16, 8, 0.9, True),
                                                       (5, 19968, torch.bfloat16, 8, 1.0, False), (17, 260, torch.float32, 4, 1.0, False),
                                                       (3, 32768, torch.float16, 8, 1.0, False)])
def test_kernel_matches_oracle(M, K, dtype, bits, clip, skip):
    from mquant_amd import ops
    x = torch.from_numpy(make_x(M * 3 + K, (M, K))).to(device=DEV, dtype=dtype)
    x[1] = 0                                                       # all-zero row: scale 1
    q, s, x0 = ops.quantize_act_dyn_i8(x, bits, clip, skip_col0=skip)
    q_ref, s_ref = oracle.quant_dyn(x.float().cpu().numpy(), bits=bits, clip=clip, skip_col0=skip)
    np.testing.assert_array_equal(s.cpu().numpy(), s_ref)
    np.testing.assert_array_equal(q.cpu().numpy()[:, :K], q_ref)
    assert not q[:, K:].any() and float(s[1]) == 1.0
    if skip:
        np.testing.assert_array_equal(x0.cpu().numpy(), x[:, 0].float().cpu().numpy())


def test_rowscale_gemm_matches_oracle_including_splitk():
    from mquant_amd import ops
    for M, K, N in ((64, 512, 96), (48, 19968, 256)):
        x = torch.from_numpy(make_x(K, (M, K))).to(DEV).half()
        w = torch.from_numpy(make_w(N, (N, K))).to(DEV).half()
        s_w, levels, _, _ = ops.wquant_sym(w, 4)
        a, s_rows, _ = ops.quantize_act_dyn_i8(x)
        bias = torch.linspace(-1, 1, N, device=DEV)
        y = ops.gemm_w4a8_rowscale(a, ops.prepack(levels, 4), 4, N, s_rows, s_w, bias=bias)
        acc = oracle.gemm_i32(a.cpu().numpy()[:, :K], levels.cpu().numpy())
        want = oracle.round_to(oracle.epilogue(acc, s_rows.cpu().numpy(), s_w.cpu().numpy(), bias=bias.cpu().numpy()), 1)
        np.testing.assert_array_equal(y.float().cpu().numpy(), want)


@pytest.mark.parametrize("case", DYN_CASES)
def test_wrapper_dynamic_mode_matches_reference_forward(golden_dir, case):
    from fake_quant import hadamard_utils as hu, quant_utils as qu, utils
    from fake_quant.gptq.rtn import rtn_module
    from mquant_amd import ops
    g = np.load(os.path.join(golden_dir, f"wrapper_dyn_{case}.npz"))
    K_in, K_pad, N, M, seed, had, split, bias, a_bits = [int(v) for v in g["meta"]]
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
    quantizers = {}
    rtn_module(wrap, "layer", 4, True, False, [], quantizers)
    wrap.quantizer.configure(bits=a_bits, sym=True, clip_ratio=float(g["clip"]))
    x = torch.from_numpy(make_x(seed + 20, (M, K_in))).to(DEV)
    assert wrap._real_ready(x), "the dynamic mode must run the real kernels too"
    y = wrap(x)
    assert wrap._real is not None and wrap._real.dynamic is not None
    np.testing.assert_allclose(y.cpu().numpy(), g["y"], rtol=0, atol=1e-3)
    # the integers of the kernels that ran
    real = wrap._real
    xr = ops.hadamard(x, real.had.n, real.had.K, real.had.bits) if had else x
    a, s_rows, _ = ops.quantize_act_dyn_i8(xr, a_bits, float(g["clip"]), skip_col0=bool(split))
    np.testing.assert_array_equal(s_rows.cpu().numpy(), g["s_rows"])
    a_rows = a.to_rows() if isinstance(a, ops.TiledAct) else a       # the engine keeps activations tiled
    np.testing.assert_array_equal(a_rows.cpu().numpy()[:, 1 if split else 0:65 if split else 64], g["qx_head"])
    np.testing.assert_array_equal(ops.gemm_w4a8_i32(a, real.w_img, 4, N).cpu().numpy(), g["acc"])


@pytest.mark.parametrize("M,K,dtype,bits,clip", [(768, 3584, torch.float16, 8, 1.0), (33, 1000, torch.float16, 8, 0.9),
                                                 (5, 19968, torch.bfloat16, 8, 1.0), (17, 260, torch.float32, 4, 1.0),
                                                 (3, 32768, torch.float16, 6, 1.0)])
@pytest.mark.parametrize("tiled", [False, True])
def test_asymmetric_kernel_matches_oracle(M, K, dtype, bits, clip, tiled):
    from mquant_amd import ops
    x = torch.from_numpy(make_x(M * 5 + K, (M, K))).to(device=DEV, dtype=dtype)
    x[0] = x[0].abs()                                              # one-sided row: xmin clamps to 0, zero point 0
    if M > 2:
        x[1] = 0                                                   # all-zero row: range (-1, +1)
        x[2] = -x[2].abs()
    q, s, z, sh = ops.quantize_act_dyn_asym_i8(x, bits, clip, tiled=tiled)
    q_ref, s_ref, z_ref, sh_ref = oracle.quant_dyn_asym(x.float().cpu().numpy(), bits=bits, clip=clip)
    np.testing.assert_array_equal(s.cpu().numpy(), s_ref)
    np.testing.assert_array_equal(z.cpu().numpy(), z_ref)
    np.testing.assert_array_equal(sh.cpu().numpy(), sh_ref)
    rows = q.to_rows() if isinstance(q, ops.TiledAct) else q
    np.testing.assert_array_equal(rows.cpu().numpy()[:, :K], q_ref)
    assert not rows[:, K:].any() and float(z[0]) == 0.0
    if M > 2:
        assert float(s[1]) == np.float32(2.0) / np.float32((1 << bits) - 1)


@pytest.mark.parametrize("case", ASYM_DYN_CASES)
def test_wrapper_asymmetric_dynamic_mode_matches_reference_forward(golden_dir, case):
    from fake_quant import hadamard_utils as hu, quant_utils as qu, utils
    from fake_quant.gptq.rtn import rtn_module
    from mquant_amd import ops
    g = np.load(os.path.join(golden_dir, f"wrapper_dyn_{case}.npz"))
    K_in, K_pad, N, M, seed, had, split, bias, a_bits = [int(v) for v in g["meta"]]
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
    rtn_module(wrap, "layer", 4, True, False, [], {})
    wrap.quantizer.configure(bits=a_bits, sym=False, clip_ratio=float(g["clip"]))
    x = torch.from_numpy(make_x(seed + 20, (M, K_in))).to(DEV)
    assert wrap._real_ready(x), "the asymmetric dynamic mode must run the real kernels"
    y = wrap(x)
    real = wrap._real
    assert real is not None and real.dynamic is not None and real.w_colsum is not None
    np.testing.assert_allclose(y.cpu().numpy(), g["y"], rtol=0, atol=1e-3)
    xr = ops.hadamard(x, real.had.n, real.had.K, real.had.bits) if had else x
    a, s_rows, zero, _ = ops.quantize_act_dyn_asym_i8(xr, a_bits, float(g["clip"]))
    np.testing.assert_array_equal(s_rows.cpu().numpy(), g["s_rows"])
    np.testing.assert_array_equal(zero.cpu().numpy(), g["zero"])
    np.testing.assert_array_equal(a.cpu().numpy()[:, :64], g["qx_head"])
    np.testing.assert_array_equal(ops.gemm_w4a8_i32(a, real.w_img, 4, N).cpu().numpy(), g["acc"])
    # a split wrapper runs the integer kernels in this mode too (two rank-1 epilogue slots, tests/test_gpu_rank2.py); split
    # column + asymmetric weights + asymmetric activations need three: since round 5 the third is added behind an fp32 GEMM output
    # (mq_rank1_add_cast; goldens wrapper_rank2_all3_*), and the result stays on the simulated evaluation of the same wrapper
    wrap2 = qu.ActQuantWrapper(torch.nn.Linear(256, 32).to(DEV))
    wrap2.split = True
    wrap2.split_weights()
    rtn_module(wrap2, "layer", 4, True, False, [], {})
    wrap2.quantizer.configure(bits=8, sym=False)
    assert wrap2._real_ready(torch.zeros(4, 256, device=DEV))
    wrap3 = qu.ActQuantWrapper(torch.nn.Linear(256, 32).to(DEV))
    wrap3.split = True
    wrap3.split_weights()
    rtn_module(wrap3, "layer", 4, False, False, [], {})
    wrap3.quantizer.configure(bits=8, sym=False)
    assert wrap3._real_ready(torch.zeros(4, 256, device=DEV)) and wrap3.backend().startswith("W4A8 integer")
    x3 = torch.from_numpy(make_x(3, (4, 256))).to(DEV)
    y3 = wrap3(x3)
    assert wrap3._real is not None and wrap3._real.n_terms == 3
    wrap3.real_quant = False
    y3_sim = wrap3(x3.clone())
    np.testing.assert_allclose(y3.cpu().numpy(), y3_sim.cpu().numpy(), rtol=0, atol=1e-3 * float(y3_sim.abs().max()))


@pytest.mark.parametrize("M,K,dtype,bits,clip,asym,skip", [(768, 3584, torch.float16, 8, 1.0, False, False), (33, 1000, torch.float16, 8, 0.9, True, False),
                                                           (5, 19968, torch.bfloat16, 8, 1.0, False, True), (17, 260, torch.float32, 4, 1.0, True, False),
                                                           (9, 512, torch.float16, 8, 1.0, True, False)])
def test_per_tensor_kernel_matches_oracle(M, K, dtype, bits, clip, asym, skip):
    from mquant_amd import ops
    x = torch.from_numpy(make_x(M * 7 + K, (M, K))).to(device=DEV, dtype=dtype)
    if M == 9:
        x = x.abs()                                                 # xmin == 0 alone: the per-tensor rule makes it -1
    q, s, z, sh, x0 = ops.quantize_act_tensor_i8(x, bits, clip, asym=asym, skip_col0=skip)
    mode = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}[dtype]      # the reference evaluates this mode in x's dtype
    q_ref, s_ref, z_ref, sh_ref = oracle.quant_tensor(x.float().cpu().numpy(), bits=bits, clip=clip, asym=asym, skip_col0=skip, mode=mode)
    np.testing.assert_array_equal(s.cpu().numpy(), np.full(M, s_ref, np.float32))
    np.testing.assert_array_equal(q.cpu().numpy()[:, :K], q_ref)
    if asym:
        np.testing.assert_array_equal(z.cpu().numpy(), np.full(M, z_ref, np.float32))
        np.testing.assert_array_equal(sh.cpu().numpy(), np.full(M, sh_ref, np.float32))
    if skip:
        np.testing.assert_array_equal(x0.cpu().numpy(), x[:, 0].float().cpu().numpy())


@pytest.mark.parametrize("case", PT_DYN_CASES)
def test_wrapper_per_tensor_dynamic_mode_matches_reference_forward(golden_dir, case):
    from fake_quant import hadamard_utils as hu, quant_utils as qu, utils
    from fake_quant.gptq.rtn import rtn_module
    from mquant_amd import ops
    g = np.load(os.path.join(golden_dir, f"wrapper_dyn_{case}.npz"))
    K_in, K_pad, N, M, seed, had, split, bias, a_bits = [int(v) for v in g["meta"]]
    asym = "zero" in g.files
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
    rtn_module(wrap, "layer", 4, True, False, [], {})
    wrap.quantizer.configure(bits=a_bits, sym=not asym, clip_ratio=float(g["clip"]), act_per_tensor=True)
    x = torch.from_numpy(make_x(seed + 20, (M, K_in))).to(DEV)
    assert wrap._real_ready(x), "the per-tensor dynamic mode must run the real kernels"
    y = wrap(x)
    real = wrap._real
    assert real is not None and real.dynamic["per_tensor"]
    np.testing.assert_allclose(y.cpu().numpy(), g["y"], rtol=0, atol=1e-3)
    xr = ops.hadamard(x, real.had.n, real.had.K, real.had.bits) if had else x
    a, s_rows, zero, _, _ = ops.quantize_act_tensor_i8(xr, a_bits, float(g["clip"]), asym=asym, skip_col0=bool(split))
    np.testing.assert_array_equal(s_rows.cpu().numpy(), g["s_rows"])
    if asym:
        np.testing.assert_array_equal(zero.cpu().numpy(), g["zero"])
    np.testing.assert_array_equal(a.cpu().numpy()[:, 1 if split else 0:65 if split else 64], g["qx_head"])
    np.testing.assert_array_equal(ops.gemm_w4a8_i32(a, real.w_img, 4, N).cpu().numpy(), g["acc"])


def test_modes_the_kernels_do_not_cover_stay_on_the_simulated_path():
    from fake_quant import quant_utils as qu
    from fake_quant.gptq.rtn import rtn_module
    lin = torch.nn.Linear(256, 32).to(DEV).half()
    x = torch.from_numpy(make_x(1, (8, 256))).to(DEV).half()
    # (symmetric AND asymmetric group-wise scales with groups of 64 / 128 / 256 ... run the integer kernels: tests/test_gpu_groupwise.py;
    #  per-tensor ranges on half activations do too, in x's dtype like the reference: the test below)
    for kw in (dict(bits=8, sym=False, groupsize=32), dict(bits=8, sym=True, groupsize=32), dict(bits=16)):
        wrap = qu.ActQuantWrapper(lin)
        rtn_module(wrap, "l", 4, True, False, [], {})
        wrap.quantizer.configure(**kw)
        assert not wrap._real_ready(x)
        assert torch.isfinite(wrap(x)).all()


def test_empty_inputs_are_accepted_everywhere():
    """M == 0 / N == 0: every entry point returns without launching (reference: empty tensors pass)."""
    from mquant_amd import ops
    e16 = torch.empty((0, 256), device=DEV, dtype=torch.float16)
    assert ops.quantize_act_dyn_i8(e16)[0].shape == (0, 256)
    assert ops.rmsn_quantize_i8(e16, 256, 1e-6, 0.1)[0].shape == (0, 256)
    assert ops.quantize_act_i8(e16, 0.1)[0].shape == (0, 256)
    assert ops.hadamard_quant_i8(e16, 256, 1, None, 0.1)[0].shape == (0, 256)
    assert ops.act_hadamard_quant_i8(e16, e16, ops.ACT_SILU_MUL, 256, 1, None, 0.1)[0].shape == (0, 256)
    s, lv, pk, _ = ops.wquant_sym(torch.empty((0, 64), device=DEV), 4, want_packed=True)
    assert s.numel() == 0 and lv.shape == (0, 64) and pk.shape == (0, 32)
    w_img = ops.prepack(torch.zeros((32, 128), dtype=torch.int8, device=DEV), 4)
    a = torch.empty((0, 128), dtype=torch.int8, device=DEV)
    assert ops.gemm_w4a8(a, w_img, 4, 32, 0.1, torch.ones(32, device=DEV)).shape == (0, 32)
    assert ops.gemm_w4a8_rowscale(a, w_img, 4, 32, torch.empty((0,), device=DEV), torch.ones(32, device=DEV)).shape == (0, 32)
    W = torch.empty((0, 16), device=DEV)
    ops.gptq_block(W, 0, 16, torch.eye(16, device=DEV), torch.empty((0,), device=DEV), 4, torch.empty((0, 16), device=DEV),
                   torch.empty((0, 16), device=DEV))
    torch.cuda.synchronize()


PT16_CASES = ["sym_3584_f16", "sym_had_5120_split_f16", "asym_1280_bf16", "asym_down_19968_f16", "sym_2048_bf16_clip"]


@pytest.mark.parametrize("case", PT16_CASES)
def test_wrapper_per_tensor_mode_on_half_activations_runs_the_reference_grid(golden_dir, case):
    """act_per_tensor on fp16 / bf16 activations: the reference evaluates range, scale, zero point, x / scale and the level
    sum in x's dtype (quant_utils.py:214-231) and so do the kernels -- scale, zero point, int8 levels and integer
    accumulators equal the reference's own (tools/gen_golden_pertensor_half.py); outputs within half-precision noise of its
    floating-point evaluation (it rounds s * (q - z) to x's dtype per element and multiplies in half precision)."""
    from fake_quant import hadamard_utils as hu, quant_utils as qu, utils
    from fake_quant.gptq.rtn import rtn_module
    from mquant_amd import ops
    g = np.load(os.path.join(golden_dir, f"wrapper_dynpt16_{case}.npz"))
    K_in, K_pad, N, M, seed, had, split, bias, a_bits, sym, dtc = [int(v) for v in g["meta"]]
    dt = {1: torch.float16, 2: torch.bfloat16}[dtc]
    lin = torch.nn.Linear(K_pad, N, bias=bool(bias))
    lin.weight.data = torch.from_numpy(make_w(seed, (N, K_pad)))
    if bias:
        lin.bias.data = torch.from_numpy(make_w(seed + 1, (N,), std=0.1))
    wrap = qu.ActQuantWrapper(lin.to(dt).to(DEV))
    if had:
        hadK, Kh = hu.get_hadK(K_pad)
        wrap.online_full_had, wrap.had_K, wrap.K = True, hadK, Kh
    if split:
        wrap.split = True
        wrap.split_weights()
    if K_pad != K_in:
        wrap.register_forward_pre_hook(functools.partial(utils.revise_down_input, new_size=K_pad))
    rtn_module(wrap, "layer", 4, True, False, [], {})
    wrap.quantizer.configure(bits=a_bits, sym=bool(sym), clip_ratio=float(g["clip"]), act_per_tensor=True)
    x = torch.from_numpy(make_x(seed + 20, (M, K_in))).to(dt).to(DEV)
    assert wrap._real_ready(x), "the per-tensor dynamic mode on half activations must run the real kernels"
    y = wrap(x)
    real = wrap._real
    assert real is not None and real.dynamic["per_tensor"] and y.dtype == dt
    tol = {1: 8e-3 if had else 4e-3, 2: 3e-2}[dtc] * float(np.abs(g["y"]).max())
    np.testing.assert_allclose(y.float().cpu().numpy(), g["y"], rtol=0, atol=tol)
    xr = ops.hadamard(x, real.had.n, real.had.K, real.had.bits) if had else x
    a, s_rows, zero, _, _ = ops.quantize_act_tensor_i8(xr, a_bits, float(g["clip"]), asym=not sym, skip_col0=bool(split))
    np.testing.assert_array_equal(s_rows.cpu().numpy(), np.full(M, g["scale"], np.float32))
    if not sym:
        np.testing.assert_array_equal(zero.cpu().numpy(), np.full(M, g["zero"], np.float32))
    lv = a.cpu().numpy()[:, 1 if split else 0:K_pad]
    np.testing.assert_array_equal(lv, g["qx"])
    np.testing.assert_array_equal(ops.gemm_w4a8_i32(a, real.w_img, 4, N).cpu().numpy(), g["acc"])


@pytest.mark.parametrize("w_sym", [True, False])
@pytest.mark.parametrize("act", ["static", "dyn_sym", "dyn_asym"])
def test_per_tensor_weight_quantizers_take_the_integer_path(w_sym, act):
    """WeightQuantizer(perchannel=False) repeats its one (scale, zero point) per output channel (reference quant_utils.py:507-509), so
    the per-channel kernels serve it unchanged -- also with asymmetric levels (round 5; it used to simulate)."""
    from fake_quant import quant_utils as qu
    K, N, M = 640, 72, 33
    lin = torch.nn.Linear(K, N, bias=True)
    lin.weight.data = torch.from_numpy(make_w(11, (N, K)))
    wrap = qu.ActQuantWrapper(lin.to(DEV))
    wq = qu.WeightQuantizer()
    wq.configure(4, perchannel=False, sym=w_sym, mse=False)
    wq.find_params(wrap.module.weight.data)
    wrap.module.weight.data = wq.quantize(wrap.module.weight.data)
    qu.attach_weight_quantizer(wrap, "module", wq)
    assert wq.scale.numel() == N and float(wq.scale.min()) == float(wq.scale.max())
    if act == "static":
        wrap.quantizer.configure(bits=8, sym=True, static=True, observer_type="minmax")
        qu.calib_layer(wrap, [torch.from_numpy(make_x(20 + i, (M, K))).to(DEV) for i in range(2)])
    else:
        wrap.quantizer.configure(bits=8, sym=act == "dyn_sym")
    x = torch.from_numpy(make_x(30, (M, K))).to(DEV)
    assert wrap._real_ready(x), wrap.backend()
    y = wrap(x)
    assert wrap._real is not None and (wrap._real.w_shift is None) == w_sym
    wrap.real_quant = False
    y_sim = wrap(x.clone())
    np.testing.assert_allclose(y.cpu().numpy(), y_sim.cpu().numpy(), rtol=0, atol=1e-3 * float(y_sim.abs().max()))
