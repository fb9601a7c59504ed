"""Pins the CPU oracle (oracle/mq_oracle.c) against outputs of the REFERENCE itself.

The fixtures in tests/golden/*.npz were produced by tools/gen_golden.py, which imports
/root/reference/fake_quant on CPU.  Integer quantities must match exactly; the Hadamard
transform matches bit for bit in all three evaluation modes; end-to-end fp outputs of
ActQuantWrapper.forward match within the 1e-3 bound of BASELINE.json's north_star (the oracle
multiplies integers then scales, the reference scales then multiplies floats).
"""
import os

import numpy as np
import pytest

import oracle
from golden_inputs import make_w, make_x


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"), allow_pickle=False)


# ------------------------------------------------------------------------------- rounding
def test_fp16_bf16_round_trips_match_numpy_and_torch():
    import torch
    rs = np.random.RandomState(0)
    x = np.concatenate([rs.standard_normal(4000).astype(np.float32) * 10 ** rs.uniform(-9, 5, 4000),
                        np.array([0.0, -0.0, 65504.0, 65519.9, 65520.0, 1e9, -1e9, 6e-8, 2.98e-8,
                                  2.99e-8, 5.96e-8, 6.1e-5, 6.097e-5], dtype=np.float32)]).astype(np.float32)
    with np.errstate(over="ignore"):
        ref16 = x.astype(np.float16).astype(np.float32)
    np.testing.assert_array_equal(oracle.round_to(x, 1), ref16)
    refbf = torch.from_numpy(x).to(torch.bfloat16).float().numpy()
    np.testing.assert_array_equal(oracle.round_to(x, 2), refbf)


# ------------------------------------------------------------------------------- Hadamard
@pytest.mark.parametrize("n", [64, 1280, 3584, 4096, 5120, 11008, 14336, 19968, 30720])
def test_hadamard_bit_exact_vs_reference(golden_dir, had_table, n):
    g = load(golden_dir, "hadamard_fwd")
    K = had_table["n2k"][n]
    hk = None if K == 1 else had_table["mats"][K]
    x = make_x(100 + n, (2 if n <= 5120 else 1, n))
    # fake_quant/hadamard_utils.py:79-100 (pure torch): butterflies, hadK @, / sqrt(n)
    np.testing.assert_array_equal(oracle.hadamard(x, n, K, hk, post_div=True), g[f"hadU_{n}"])
    # :115-128 (CUDA path): scale before hadK @, fp32 and fp16
    np.testing.assert_array_equal(oracle.hadamard(x, n, K, hk), g[f"cuda_{n}"])
    x16 = oracle.round_to(x, 1)
    np.testing.assert_array_equal(oracle.hadamard(x16, n, K, hk, mid_round=1, out_round=1),
                                  g[f"cuda16_{n}"].astype(np.float32))


def test_special_hadamard_matrices_are_hadamard(had_table):
    for K, H in had_table["mats"].items():
        H = H.astype(np.int64)
        assert np.array_equal(H @ H.T, K * np.eye(K, dtype=np.int64)), K


def test_hadk_dispatch_table(golden_dir):
    g = load(golden_dir, "hadk_table")
    order = (172, 156, 140, 108, 60, 52, 36, 28, 40, 20, 12)
    for n, K in zip(g["n"].tolist(), g["K"].tolist()):
        mine = next((k for k in order if n % k == 0), 1)
        assert mine == K, (n, K, mine)


# ------------------------------------------------------------------------------- quantizer
@pytest.mark.parametrize("tag", ["2d", "3d"])
def test_uniform_quantizer_levels_and_dequant(golden_dir, tag):
    g = load(golden_dir, "uniform_quantizer")
    x = g[f"x_{tag}"]
    rows = x.reshape(-1, x.shape[-1])
    s = np.float32(g["scale_lw"])
    q = oracle.quant_static(rows, s).reshape(x.shape)
    np.testing.assert_array_equal(q, g[f"q_{tag}_layer_wise"])
    np.testing.assert_array_equal(oracle.dequant_static(q.reshape(rows.shape), s).reshape(x.shape),
                                  g[f"dq_{tag}_layer_wise"])
    sv = g["scale_cw"]
    qc = oracle.quant_static(rows, sv).reshape(x.shape)
    np.testing.assert_array_equal(qc, g[f"q_{tag}_channel_wise"])
    np.testing.assert_array_equal(oracle.dequant_static(qc.reshape(rows.shape), sv).reshape(x.shape),
                                  g[f"dq_{tag}_channel_wise"])
    # fp16 input: x.float() first, result cast back to fp16
    x16 = oracle.round_to(rows, 1)
    q16 = oracle.quant_static(x16, s)
    np.testing.assert_array_equal(q16.reshape(x.shape), g[f"q16_{tag}_layer_wise"])
    np.testing.assert_array_equal(oracle.round_to(oracle.dequant_static(q16, s), 1).reshape(x.shape),
                                  g[f"dq16_{tag}_layer_wise"])


def test_minmax_scale_formula(golden_dir):
    g = load(golden_dir, "observers")
    mn, mx = g["minmax_layer_wise_min2"], g["minmax_layer_wise_max2"]
    np.testing.assert_array_equal(oracle.minmax_scale_sym(mn, mx), g["minmax_layer_wise_scale"])
    mn, mx = g["minmax_channel_wise_min2"], g["minmax_channel_wise_max2"]
    np.testing.assert_array_equal(oracle.minmax_scale_sym(mn, mx), g["minmax_channel_wise_scale"])
    # per-channel reduction of the first batch incl. the zero-inclusion rule
    b0 = g["batch0"].reshape(-1, g["batch0"].shape[-1])
    cmn, cmx = oracle.minmax_channels(b0)
    np.testing.assert_array_equal(np.minimum(cmn, 0), g["minmax_channel_wise_min0"])
    np.testing.assert_array_equal(np.maximum(cmx, 0), g["minmax_channel_wise_max0"])


def test_pack_i4_wire_format(golden_dir):
    g = load(golden_dir, "pack_i4")
    np.testing.assert_array_equal(oracle.pack_i4(g["q"]), g["packed"])
    np.testing.assert_array_equal(oracle.unpack_i4(g["packed"]), g["unpacked"].astype(np.int8))


@pytest.mark.parametrize("tag,bits,mse", [("w4_sym", 4, False), ("w8_sym", 8, False),
                                          ("w4_sym_mse", 4, True), ("w8_sym_mse", 8, True)])
def test_weight_quantizer_sym(golden_dir, tag, bits, mse):
    g = load(golden_dir, "weight_quantizer")
    W = g["W"]
    scale, levels = oracle.wquant_sym(W, bits=bits, mse=mse)
    ref_scale = g[f"scale_{tag}"].reshape(-1)
    if not mse:
        np.testing.assert_array_equal(scale, ref_scale)
    else:
        # the clip search compares sums of |err|^2.4 (libm powf vs torch pow): identical picks
        # except on exact near-ties; require agreement on >= 90 % of the channels and a bounded gap
        same = scale == ref_scale
        assert same.mean() >= 0.9, same.mean()
        assert np.max(np.abs(scale - ref_scale) / ref_scale) < 0.05
    ok = scale == ref_scale
    wq = levels.astype(np.float32) * scale[:, None]
    np.testing.assert_array_equal(wq[ok], g[f"wq_{tag}"][ok])


# ------------------------------------------------------------------------------- whole layer
WRAPPER_CASES = ["plain_3584", "plain_4096_mse", "plain_w8", "had_5120_split", "had_5120",
                 "had_5120_fp32had", "had_11008", "had_8192", "down_19968", "down_19968_split",
                 "had_14336"]


@pytest.mark.parametrize("case", WRAPPER_CASES)
def test_wrapper_forward_integer_restatement(golden_dir, had_table, case):
    """ActQuantWrapper.forward (quant_utils.py:330-391) restated on the integer grid."""
    g = load(golden_dir, "wrapper_" + case)
    K_in, K_pad, N, M, seed, had, split, w_bits, w_mse, bias = [int(v) for v in g["meta"]]
    fp32_had = case.endswith("fp32had")
    x = make_x(seed + 20, (M, K_in))
    rot = x
    if had:
        Kh = int(g["had_K"])
        hk = None if Kh == 1 else had_table["mats"][Kh]
        rot = oracle.hadamard(x, K_pad, Kh, hk)          # fp32 tensors: no intermediate casts
        np.testing.assert_array_equal(rot[:, :512], g["x_rot"])
    elif K_pad != K_in:
        rot = np.pad(x, ((0, 0), (0, K_pad - K_in)))
    s_x = np.float32(g["s_x"])
    q = oracle.quant_static(rot, s_x)
    if split:
        q[:, 0] = 0
    # weights: the reference's RTN on the same synthetic W
    W = make_w(seed, (N, K_pad))
    if had or split or True:
        Wq_src = W[:, 1:] if split else W
    s_w, levels = oracle.wquant_sym(np.ascontiguousarray(Wq_src), bits=w_bits, mse=bool(w_mse))
    if w_mse:
        s_w = g["s_w"]                                    # clip search: take the reference's pick
        levels = np.clip(np.rint(Wq_src / s_w[:, None]), -(1 << (w_bits - 1)), (1 << (w_bits - 1)) - 1).astype(np.int8)
    np.testing.assert_array_equal(s_w, g["s_w"])
    if split:
        levels = np.concatenate([np.zeros((N, 1), np.int8), levels], axis=1)
    acc = oracle.gemm_i32(q, levels)
    np.testing.assert_array_equal(acc, g["acc"])
    b = make_w(seed + 1, (N,), std=0.1) if bias else None
    y = oracle.epilogue(acc, s_x, s_w, bias=b,
                        x0=rot[:, 0] if split else None, w0=W[:, 0] if split else None)
    np.testing.assert_allclose(y, g["y"], rtol=0, atol=1e-3)
    assert np.abs(y - g["y"]).max() < 1e-4 * max(1.0, np.abs(g["y"]).max())


WQUANT_CASES = ["f32_4b_rtn", "f32_4b_mse", "f32_8b_rtn", "f32_8b_mse", "f16_4b_rtn", "f16_4b_mse",
                "bf16_4b_rtn", "bf16_4b_mse", "f16_8b_rtn"]


def wquant_case(golden_dir, case):
    """(golden, fp32 view of the weights in the case's dtype, rounding mode of that dtype)."""
    from golden_inputs import make_wquant_weights
    g = np.load(os.path.join(golden_dir, f"wquant_{case}.npz"))
    seed, bits, mse = [int(v) for v in g["meta"]]
    mode = {"f32": 0, "f16": 1, "bf16": 2}[case.split("_")[0]]
    w = oracle.round_to(make_wquant_weights(seed), mode)
    return g, w, mode, bits, bool(mse)


@pytest.mark.parametrize("case", WQUANT_CASES)
def test_weight_quantizer_on_every_weight_dtype(golden_dir, case):
    """quant_utils.py:446-518 on fp32 / fp16 / bf16 weights: half weights are promoted by the fp32
    ``tmp`` tensor (:458-460), so scales and errors are fp32 and only W~ is cast back."""
    g, w, mode, bits, mse = wquant_case(golden_dir, case)
    scale, levels = oracle.wquant_sym(w, bits=bits, mse=mse)
    np.testing.assert_array_equal(scale, g["scale"])
    wq = oracle.round_to(scale[:, None] * levels.astype(np.float32), mode)
    np.testing.assert_array_equal(wq, g["wq"])
    if mse:
        assert (scale < oracle.wquant_sym(w, bits=bits, mse=False)[0]).sum() >= 2      # the search does clip


def test_dynamic_per_token_quantizer_against_reference(golden_dir):
    """quant_utils.py:205-268 (default activation mode): levels and per-row scales."""
    g = np.load(os.path.join(golden_dir, "act_dynamic.npz"))
    x = g["x"].reshape(-1, g["x"].shape[-1])
    for tag, kw in (("tok_sym", dict(bits=8)), ("tok_sym4", dict(bits=4)), ("clip_sym", dict(bits=8, clip=0.9))):
        q, s = oracle.quant_dyn(x, **kw)
        np.testing.assert_array_equal((q.astype(np.float32) * s[:, None]).reshape(g["x"].shape), g[f"y_{tag}"], err_msg=tag)
        np.testing.assert_array_equal(np.broadcast_to(s.reshape(g["x"].shape[:-1] + (1,)), g["x"].shape), g[f"scale_{tag}"])


DYN_CASES = ["plain_3584", "clip_1280", "had_5120_split", "down_19968"]


@pytest.mark.parametrize("case", DYN_CASES)
def test_dynamic_wrapper_layer_against_reference(golden_dir, had_table, case):
    """Whole layer in the dynamic mode: [pad, Hadamard,] per-token quantize, int GEMM, dequant."""
    from golden_inputs import make_w, make_x
    g = np.load(os.path.join(golden_dir, f"wrapper_dyn_{case}.npz"))
    K_in, K_pad, N, M, seed, had, split, bias, a_bits = [int(v) for v in g["meta"]]
    x = make_x(seed + 20, (M, K_in))
    if had:
        K = had_table["n2k"][K_pad]
        x = oracle.hadamard(x, K_pad, K, had_table["mats"][K], mid_round=0, out_round=0)
    q, s_rows = oracle.quant_dyn(x, bits=a_bits, clip=float(g["clip"]), skip_col0=bool(split))
    np.testing.assert_array_equal(s_rows, g["s_rows"])
    np.testing.assert_array_equal(q[:, 1 if split else 0:65 if split else 64], g["qx_head"])
    W = make_w(seed, (N, K_pad))
    Wsrc = W[:, 1:] if split else W
    s_w, levels = oracle.wquant_sym(np.ascontiguousarray(Wsrc), bits=4)
    np.testing.assert_array_equal(s_w, g["s_w"])
    if split:
        levels = np.concatenate([np.zeros((N, 1), np.int8), levels], axis=1)
    acc = oracle.gemm_i32(q, levels)
    np.testing.assert_array_equal(acc, g["acc"])
    b = make_w(seed + 1, (N,), std=0.1) if bias else None
    y = oracle.epilogue(acc, s_rows, s_w, bias=b, x0=x[:, 0] if split else None, w0=W[:, 0] if split else None)
    np.testing.assert_allclose(y, g["y"], rtol=0, atol=1e-3)


ASYM_DYN_CASES = ["asym_3584", "asym_clip_1280", "asym_down_19968", "asym_a6_2048"]


@pytest.mark.parametrize("case", ASYM_DYN_CASES)
def test_asymmetric_dynamic_wrapper_layer_against_reference(golden_dir, had_table, case):
    """--a_asym: per-token scale and zero point (quant_utils.py:239-268 else-branch), levels stored minus
    2^(bits-1) for the int8 GEMM, zero point and offset restored by the rank-1 epilogue term."""
    from golden_inputs import make_w, make_x
    g = np.load(os.path.join(golden_dir, f"wrapper_dyn_{case}.npz"))
    K_in, K_pad, N, M, seed, had, split, bias, a_bits = [int(v) for v in g["meta"]]
    assert not split and int(g["sym"]) == 0
    x = make_x(seed + 20, (M, K_in))
    if had:
        K = had_table["n2k"][K_pad]
        x = oracle.hadamard(x, K_pad, K, had_table["mats"][K], mid_round=0, out_round=0)
    q, s_rows, zero, shift = oracle.quant_dyn_asym(x, bits=a_bits, clip=float(g["clip"]))
    np.testing.assert_array_equal(s_rows, g["s_rows"])
    np.testing.assert_array_equal(zero, g["zero"])
    np.testing.assert_array_equal(q[:, :64], g["qx_head"])
    half = 1 << (a_bits - 1)
    assert q.min() >= -half and q.max() <= half - 1
    np.testing.assert_array_equal(shift, s_rows * (np.float32(half) - zero))
    W = make_w(seed, (N, K_pad))
    s_w, levels = oracle.wquant_sym(np.ascontiguousarray(W), bits=4)
    np.testing.assert_array_equal(s_w, g["s_w"])
    acc = oracle.gemm_i32(q, levels)
    np.testing.assert_array_equal(acc, g["acc"])
    b = make_w(seed + 1, (N,), std=0.1) if bias else None
    w0 = levels.astype(np.int32).sum(axis=1).astype(np.float32) * s_w
    y = oracle.epilogue(acc, s_rows, s_w, bias=b, x0=shift, w0=w0)
    np.testing.assert_allclose(y, g["y"], rtol=0, atol=1e-3)


PT_DYN_CASES = ["pt_sym_3584", "pt_sym_had_5120_split", "pt_asym_1280", "pt_asym_down_19968"]


@pytest.mark.parametrize("case", PT_DYN_CASES)
def test_per_tensor_dynamic_wrapper_layer_against_reference(golden_dir, had_table, case):
    """act_per_tensor: one dynamic range for the whole activation tensor (quant_utils.py:214-237)."""
    from golden_inputs import make_w, make_x
    g = np.load(os.path.join(golden_dir, f"wrapper_dyn_{case}.npz"))
    K_in, K_pad, N, M, seed, had, split, bias, a_bits = [int(v) for v in g["meta"]]
    asym = "zero" in g.files
    x = make_x(seed + 20, (M, K_in))
    if had:
        K = had_table["n2k"][K_pad]
        x = oracle.hadamard(x, K_pad, K, had_table["mats"][K], mid_round=0, out_round=0)
    q, s, z, shift = oracle.quant_tensor(x, bits=a_bits, clip=float(g["clip"]), asym=asym, skip_col0=bool(split))
    np.testing.assert_array_equal(np.full(M, s, np.float32), g["s_rows"])
    if asym:
        np.testing.assert_array_equal(np.full(M, z, np.float32), g["zero"])
    np.testing.assert_array_equal(q[:, 1 if split else 0:65 if split else 64], g["qx_head"])
    W = make_w(seed, (N, K_pad))
    Wsrc = W[:, 1:] if split else W
    s_w, levels = oracle.wquant_sym(np.ascontiguousarray(Wsrc), bits=4)
    if split:
        levels = np.concatenate([np.zeros((N, 1), np.int8), levels], axis=1)
    acc = oracle.gemm_i32(q, levels)
    np.testing.assert_array_equal(acc, g["acc"])
    b = make_w(seed + 1, (N,), std=0.1) if bias else None
    if asym:
        x0, w0 = np.full(M, shift, np.float32), levels.astype(np.int32).sum(axis=1).astype(np.float32) * s_w
    else:
        x0, w0 = (x[:, 0], W[:, 0]) if split else (None, None)
    y = oracle.epilogue(acc, np.full(M, s, np.float32), s_w, bias=b, x0=x0, w0=w0)
    np.testing.assert_allclose(y, g["y"], rtol=0, atol=1e-3)


WASYM_CASES = ["plain_3584", "mse_1280", "w8_2048", "down_19968", "dyn_3584"]


@pytest.mark.parametrize("case", WASYM_CASES)
def test_asymmetric_weight_wrapper_layer_against_reference(golden_dir, had_table, case):
    """--w_asym: per-channel scale and zero point of the weights (quant_utils.py:446-509); the int GEMM
    stores q - 2^(bits-1), the zero points return through the rank-1 term (s_x rowsum_a) * s_w (2^(bits-1) - z_w)."""
    from golden_inputs import make_w, make_x
    g = np.load(os.path.join(golden_dir, f"wrapper_wasym_{case}.npz"))
    K_in, K_pad, N, M, seed, had, bias, w_bits, w_mse, dynamic = [int(v) for v in g["meta"]]
    W = make_w(seed, (N, K_pad))
    s_w, z_w, levels = oracle.wquant_asym(W, bits=w_bits, mse=bool(w_mse))
    np.testing.assert_array_equal(s_w, g["s_w"])
    np.testing.assert_array_equal(z_w, g["z_w"])
    np.testing.assert_array_equal(levels[:, :64], g["qw_head"])
    x = make_x(seed + 20, (M, K_in))
    if K_pad != K_in and not had:
        x = np.pad(x, ((0, 0), (0, K_pad - K_in)))
    if had:
        K = had_table["n2k"][K_pad]
        x = oracle.hadamard(x, K_pad, K, had_table["mats"][K], mid_round=0, out_round=0)
    if dynamic:
        q, s_rows = oracle.quant_dyn(x, bits=8, clip=1.0)
        np.testing.assert_array_equal(s_rows, g["s_rows"])
        s_x = s_rows
    else:
        s_x = np.float32(g["s_x"])
        q = oracle.quant_static(x, s_x)
        s_x = np.full(M, s_x, np.float32)
    acc = oracle.gemm_i32(q, levels)
    np.testing.assert_array_equal(acc, g["acc"])
    rowsum = q.astype(np.int64).sum(axis=1)
    np.testing.assert_array_equal(rowsum, g["qx_sum"])
    b = make_w(seed + 1, (N,), std=0.1) if bias else None
    x0 = s_x * rowsum.astype(np.float32)
    w0 = s_w * (np.float32(1 << (w_bits - 1)) - z_w)
    y = oracle.epilogue(acc, s_x, s_w, bias=b, x0=x0, w0=w0)
    np.testing.assert_allclose(y, g["y"], rtol=0, atol=1e-3)
