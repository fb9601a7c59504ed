"""NON-DEFAULT fast stage of the online Hadamard kernels (per-call flag ``MQ_HAD_FAST`` / ``fast=True`` in ops; a field of
the layer's ``HadamardSpec`` in the engine -- nothing process-wide): the K x K stage on the half-precision matrix core.  Same exact +-1 products as the exact mode, another fp32
accumulation order: the rotated activations may differ from the exact mode (which is pinned bit for bit to the
reference goldens) by ONE unit in the last place of x's dtype on a small fraction of the elements, the int8
levels by one step on a smaller fraction still.  The tolerances below state exactly that.  Everything the fast
mode does not cover must fall back to the exact kernel, bit for bit."""
import numpy as np
import pytest
import torch

from golden_inputs import make_x

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

# (K, n) pairs: every special factor the reference knows with co-factors 64 / 128 / 512, incl. the BASELINE shapes
SHAPES = [(156, 19968), (40, 5120), (172, 11008), (28, 14336), (60, 30720), (12, 768), (36, 2304),
          (52, 6656), (108, 6912), (140, 8960)]


def _ulp(y: torch.Tensor, dt) -> torch.Tensor:
    """one unit in the last place of dtype dt at |y| (normal range)"""
    mant = 10 if dt == torch.float16 else 7
    return torch.pow(2.0, torch.floor(torch.log2(y.abs().clamp_min(1e-30))) - mant)


@pytest.mark.parametrize("K,n", SHAPES)
@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_fast_rotation_is_within_one_ulp_of_the_exact_mode(K, n, dt):
    from fake_quant import hadamard_utils as hu
    from mquant_amd import ops
    assert hu.get_hadK(n)[1] == K
    bits = hu.had_sign_bits(K, DEV)
    M, n_in = 5, n - (n // 16 if K in (156, 60) else 0)             # ragged input width for the padded shapes
    x = torch.from_numpy(make_x(K + n, (M, n_in))).to(DEV).to(dt)
    exact = ops.hadamard(x, n, K, bits).float()
    fast = ops.hadamard(x, n, K, bits, fast=True).float()
    assert torch.isfinite(fast).all()
    diff = (fast - exact).abs()
    assert bool((diff <= _ulp(exact, dt) * 1.001).all()), float((diff / _ulp(exact, dt)).max())
    assert float((diff > 0).float().mean()) < 0.02                   # and only on a small fraction of the elements


@pytest.mark.parametrize("K,n,tiled,split,msq", [(156, 19968, True, False, True), (40, 5120, True, True, False),
                                                 (172, 11008, False, False, False), (28, 14336, True, False, True),
                                                 (60, 30720, False, True, True)])
def test_fast_quantized_levels_against_the_exact_mode(K, n, tiled, split, msq):
    from fake_quant import hadamard_utils as hu
    from mquant_amd import ops
    bits = hu.had_sign_bits(K, DEV)
    M = 37
    n_in = n - 1024 if K in (156, 60) else n
    x = torch.from_numpy(make_x(3 * K + 1, (M, n_in))).to(DEV).half()
    rot = ops.hadamard(x, n, K, bits).float()
    s0 = float(rot.abs().max()) / 127.0
    s1 = 0.7 * s0 if msq else None
    sel = (torch.arange(M, device=DEV) % 3 == 0).to(torch.uint8) if msq else None
    outs = {}
    for fast in (False, True):
        q, x0 = ops.hadamard_quant_i8(x, n, K, bits, s0, s1, row_sel=sel, skip_col0=split, tiled=tiled, fast=fast)
        rows = q.to_rows() if tiled else q
        outs[fast] = (rows.cpu().numpy().astype(np.int16), None if x0 is None else x0.cpu().numpy())
    (qe, x0e), (qf, x0f) = outs[False], outs[True]
    assert qe.shape == qf.shape and qe.shape[1] % 128 == 0
    assert not qf[:, n:].any()                                       # zero fill up to K_pad
    d = np.abs(qe[:, :n] - qf[:, :n])
    assert d.max() <= 1 and (d > 0).mean() < 2e-3, (d.max(), (d > 0).mean())
    if split:
        assert not qf[:, 0].any()
        np.testing.assert_allclose(x0f, x0e, rtol=2 ** -10, atol=0)   # column 0 stays in floating point: <= 1 ulp(fp16)


def test_fast_mode_with_the_fused_activation_prologue():
    from fake_quant import hadamard_utils as hu
    from mquant_amd import ops
    K, n, n_in, M = 156, 19968, 18944, 9
    bits = hu.had_sign_bits(K, DEV)
    g = torch.from_numpy(make_x(5, (M, n_in))).to(DEV).half()
    u = torch.from_numpy(make_x(6, (M, n_in))).to(DEV).half()
    h = torch.nn.functional.silu(g) * u
    ref, _ = ops.hadamard_quant_i8(h, n, K, bits, 0.02, tiled=True, fast=True)
    got, _ = ops.act_hadamard_quant_i8(g, u, ops.ACT_SILU_MUL, n, K, bits, 0.02, tiled=True, fast=True)
    assert torch.equal(ref.to_rows(), got.to_rows())                   # same mode on both sides: identical bits


def test_what_the_fast_mode_does_not_cover_runs_the_exact_kernel_bit_for_bit():
    from fake_quant import hadamard_utils as hu
    from mquant_amd import ops
    cases = [(156, 19968, torch.float32, False),      # fp32 activations are not half numbers
             (40, 5120, torch.float16, True),         # --fp32_had keeps fp32 between the stages
             (1, 8192, torch.float16, False),         # pure power of two: no K x K stage at all
             (12, 384, torch.float16, False)]         # co-factor 32 < 64
    for K, n, dt, fp32_had in cases:
        bits = hu.had_sign_bits(K, DEV) if K > 1 else None
        x = torch.from_numpy(make_x(K + 7, (6, n))).to(DEV).to(dt)
        got = ops.hadamard(x, n, K, bits, fp32_had, fast=True)
        want = ops.hadamard(x, n, K, bits, fp32_had)
        assert torch.equal(got, want), (K, n, dt)


def test_there_is_no_process_wide_mode():
    """The fast stage is a flag of the call (MQ_HAD_FAST) / a field of the layer's HadamardSpec: the C ABI has no setter."""
    from mquant_amd import _lib, engine
    lib = _lib.load()
    assert not hasattr(lib, "mq_hadamard_set_mode") and "mq_hadamard_set_mode" not in _lib.SIGNATURES
    assert engine.HadamardSpec(128, 1, None).fast is False


# ---- the verdict against the REFERENCE: every golden the reference's own forward produced on HALF tensors through an online
# Hadamard (the fast stage only exists for half activations; on fp32 tensors -- the eleven static wrapper_* goldens and the
# fp32 dynamic ones -- the flag takes the exact kernel, previous test).  Same fixtures, same assertions as the exact-mode tests
# (tests/test_gpu_dynamic.py, tests/test_gpu_groupwise.py): scales, zero points, int8 levels and int32 accumulators bit for
# bit, outputs within the same tolerance -- with the engine's rotation flagged fast.
def _build_half_wrapper(meta, dt, had, split, bias, configure):
    import functools
    from fake_quant import hadamard_utils as hu, quant_utils as qu, utils
    from fake_quant.gptq.rtn import rtn_module
    from golden_inputs import make_w
    K_in, K_pad, N, seed = meta
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
    configure(wrap.quantizer)
    return wrap


@pytest.mark.parametrize("case", ["sym_had_5120_split_f16", "asym_down_19968_f16"])
def test_fast_stage_against_the_reference_goldens_per_tensor_ranges(golden_dir, case):
    import os
    from mquant_amd import ops
    g = np.load(os.path.join(golden_dir, f"wrapper_dynpt16_{case}.npz"))
    K_in, K_pad, N, M, seed, had, split, bias, a_bits, sym, dtc = [int(v) for v in g["meta"]]
    assert had and dtc == 1
    dt = torch.float16
    wrap = _build_half_wrapper((K_in, K_pad, N, seed), dt, had, split, bias,
                               lambda q: q.configure(bits=a_bits, sym=bool(sym), clip_ratio=float(g["clip"]), act_per_tensor=True))
    x = torch.from_numpy(make_x(seed + 20, (M, K_in))).to(dt).to(DEV)
    y_exact = wrap(x)
    real = wrap._real
    assert real is not None and real.had is not None and real.had.fast is False
    real.had.fast = True                                           # this layer only
    y = wrap(x)
    tol = 8e-3 * float(np.abs(g["y"]).max())
    np.testing.assert_allclose(y.float().cpu().numpy(), g["y"], rtol=0, atol=tol)
    xr = ops.hadamard(x, real.had.n, real.had.K, real.had.bits, fast=True)
    a, s_rows, zero, _, _ = ops.quantize_act_tensor_i8(xr, a_bits, float(g["clip"]), asym=not sym, skip_col0=bool(split))
    np.testing.assert_array_equal(s_rows.cpu().numpy(), np.full(M, g["scale"], np.float32))
    if not sym:
        np.testing.assert_array_equal(zero.cpu().numpy(), np.full(M, g["zero"], np.float32))
    lv = a.cpu().numpy()[:, 1 if split else 0:K_pad]
    flips = int((lv != g["qx"]).sum())
    print(f"wrapper_dynpt16_{case}: fast stage, {flips} of {lv.size} int8 levels differ from the reference's; "
          f"max |y - y_ref| = {float(np.abs(y.float().cpu().numpy() - g['y']).max()):.3e} (exact kernel: "
          f"{float(np.abs(y_exact.float().cpu().numpy() - g['y']).max()):.3e}), |y_ref| max {float(np.abs(g['y']).max()):.3e}")
    np.testing.assert_array_equal(lv, g["qx"])
    np.testing.assert_array_equal(ops.gemm_w4a8_i32(a, real.w_img, 4, N).cpu().numpy(), g["acc"])


def test_fast_stage_against_the_reference_goldens_group_wise(golden_dir):
    from mquant_amd import ops
    from test_groupwise_cpu import DT, cases, load
    seen = 0
    for path in cases(golden_dir):
        g, c = load(path)
        if not c["had"] or c["dtc"] == 0:
            continue
        seen += 1
        dt = DT[c["dtc"]]
        wrap = _build_half_wrapper((c["K_in"], c["K_pad"], c["N"], c["seed"]), dt, True, False, c["bias"],
                                   lambda q: q.configure(bits=c["bits"], groupsize=c["g"], sym=True, clip_ratio=float(g["clip"])))
        x = torch.from_numpy(make_x(c["seed"] + 20, (1, c["M"], c["K_in"]))).to(dt).to(DEV)
        y_exact = wrap(x)
        real = wrap._real
        real.had.fast = True
        y = wrap(x)
        tol = {1: 2e-3, 2: 8e-3}[c["dtc"]] * float(np.abs(g["y"]).max())
        np.testing.assert_allclose(y.float().cpu().numpy().reshape(c["M"], c["N"]), g["y"], rtol=0, atol=tol, err_msg=path)
        rows = x.reshape(c["M"], c["K_in"])
        xr = ops.hadamard(rows, real.had.n, real.had.K, real.had.bits, fast=True)
        a, s = ops.quantize_act_group_i8(xr, c["g"], c["bits"], float(g["clip"]))
        lv = a[:, :c["K_pad"]].cpu().numpy()
        flips = int((lv != g["qx"]).sum())
        print(f"{path.split('/')[-1]}: fast stage, {flips} of {lv.size} int8 levels and {int((s.cpu().numpy() != g['s_groups']).sum())} of "
              f"{g['s_groups'].size} group scales differ from the reference's; max |y - y_ref| = "
              f"{float(np.abs(y.float().cpu().numpy().reshape(c['M'], c['N']) - g['y']).max()):.3e} (exact kernel: "
              f"{float(np.abs(y_exact.float().cpu().numpy().reshape(c['M'], c['N']) - g['y']).max()):.3e})")
        np.testing.assert_array_equal(s.cpu().numpy(), g["s_groups"], err_msg=path)
        np.testing.assert_array_equal(lv, g["qx"], err_msg=path)
    assert seen == 2


@pytest.mark.parametrize("case,dt", [("had_5120_split", torch.float16), ("had_5120", torch.bfloat16), ("down_19968", torch.float16),
                                     ("down_19968_split", torch.bfloat16), ("had_11008", torch.float16), ("had_14336", torch.float16)])
def test_fast_stage_static_quantizer_against_the_oracle(golden_dir, had_table, case, dt):
    """Static int8 mode (the benchmark's): the reference's static goldens are fp32 (exact kernel either way), so the wrapper is
    built on HALF tensors from the same fixtures' shapes and seeds and held to the oracle composition (pinned to the reference):
    calibrated scale identical, int8 levels and int32 accumulators bit for bit, outputs equal."""
    import os
    import oracle
    from fake_quant import quant_utils as qu
    from test_gpu_wrapper_golden import Args, build_wrapper
    g = np.load(os.path.join(golden_dir, f"wrapper_{case}.npz"))
    wrap, quantizers, (K_in, K_pad, N, M, seed, split) = build_wrapper(g, case, dt)
    batches = [torch.from_numpy(make_x(seed + 10 + i, (M, K_in))).to(device=DEV, dtype=dt) for i in range(3)]
    qu.calib_layer(wrap, batches, Args())
    x = torch.from_numpy(make_x(seed + 20, (M, K_in))).to(device=DEV, dtype=dt)
    y_exact = wrap(x)
    real = wrap._real
    a_exact, _ = real.quantize(x)
    q_exact = a_exact.to_rows().clone()
    real.had.fast = True
    y = wrap(x)
    a, _ = real.quantize(x)
    mode = 1 if dt == torch.float16 else 2
    Kh = had_table["n2k"][K_pad]
    s_x = np.float32(float(wrap.quantizer.quantizer.scale))
    rot = oracle.hadamard(x.float().cpu().numpy(), K_pad, Kh, had_table["mats"][Kh], mid_round=mode, out_round=mode)
    q_ref = oracle.quant_static(rot, s_x)
    if split:
        q_ref[:, 0] = 0
    lv = a.to_rows().cpu().numpy()[:, :K_pad]
    flips = int((lv != q_ref).sum())
    print(f"wrapper_{case} on {dt}: fast stage, {flips} of {lv.size} int8 levels differ from the oracle's; "
          f"max |y_fast - y_exact| = {float((y.float() - y_exact.float()).abs().max()):.3e}")
    np.testing.assert_array_equal(q_exact.cpu().numpy()[:, :K_pad], q_ref)
    np.testing.assert_array_equal(lv, q_ref)
    assert torch.equal(y, y_exact)
