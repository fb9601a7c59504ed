"""GPU parity: every HIP entry point of include/mquant_hip.h against the CPU oracle.

Bar: bit-exact for int8 levels, int32 accumulators and packed bytes; fp outputs computed
with the same single-rounded fp32 operations as the oracle are also compared bit-exact
(the 1e-3 tolerance named by BASELINE.json's north_star is the bound against the
REFERENCE's fp evaluation order, exercised in test_gpu_wrapper_golden.py).
"""
import numpy as np
import pytest
import torch

import oracle
from golden_inputs import make_ties, make_w, make_x

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
DTYPES = [torch.float16, torch.bfloat16, torch.float32]
MODE = {torch.float16: 1, torch.bfloat16: 2, torch.float32: 0}


def ops():
    from mquant_amd import ops as o
    return o


def to_dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(DEV)


def as_f32(t):
    return t.float().cpu().numpy()


# ----------------------------------------------------------------------------- act quant
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(1, 16), (7, 100), (33, 1176), (64, 3584), (5, 4097)])
def test_quantize_act_per_tensor(dtype, shape):
    x = make_x(1, shape, outlier_gain=30.0)
    scale = np.float32(0.0371)
    flat = x.reshape(-1)
    ties = make_ties(scale, np.arange(-130, 130, 3))[: flat.size // 2]
    flat[: ties.size] = ties
    xt = to_dev(x, dtype)
    xr = as_f32(xt)  # what the kernel really sees after the dtype cast
    q, _ = ops().quantize_act_i8(xt, float(scale))
    K = shape[1]
    ref = oracle.quant_static(xr, scale)
    got = q.cpu().numpy()
    assert got.shape == (shape[0], (K + 127) // 128 * 128)
    np.testing.assert_array_equal(got[:, :K], ref)
    assert not got[:, K:].any()


@pytest.mark.parametrize("dtype", DTYPES)
def test_quantize_act_msq_rowsel_and_split(dtype):
    M, K = 48, 640
    x = make_x(2, (M, K))
    xt = to_dev(x, dtype)
    xr = as_f32(xt)
    sel = (np.arange(M) >= 16).astype(np.uint8)  # first 16 rows "vision", rest "text"
    s0, s1 = np.float32(0.05), np.float32(0.011)
    q, x0 = ops().quantize_act_i8(xt, float(s0), float(s1), row_sel=to_dev(sel), skip_col0=True)
    ref = oracle.quant_static(xr, s0, scale1=s1, row_sel=sel)
    ref[:, 0] = 0
    np.testing.assert_array_equal(q.cpu().numpy()[:, :K], ref)
    np.testing.assert_array_equal(x0.cpu().numpy(), xr[:, 0])


@pytest.mark.parametrize("dtype", DTYPES)
def test_quantize_act_per_channel(dtype):
    M, K = 20, 200
    x = make_x(3, (M, K))
    sv = (0.01 + 0.003 * np.arange(K)).astype(np.float32)
    xt = to_dev(x, dtype)
    q, _ = ops().quantize_act_i8(xt, scale_vec0=to_dev(sv))
    ref = oracle.quant_static(as_f32(xt), sv)
    np.testing.assert_array_equal(q.cpu().numpy()[:, :K], ref)


@pytest.mark.parametrize("dtype", DTYPES)
def test_fakequant_act(dtype):
    M, K = 9, 333
    x = make_x(4, (M, K))
    xt = to_dev(x, dtype)
    s = np.float32(0.043)
    y = ops().fakequant_act(xt, float(s))
    xr = as_f32(xt)
    ref = oracle.round_to(oracle.dequant_static(oracle.quant_static(xr, s), s), MODE[dtype])
    np.testing.assert_array_equal(as_f32(y), ref)


def test_quantize_noncontiguous_rows_and_empty():
    x = make_x(5, (10, 512))
    xt = to_dev(x, torch.float16)[:, :300]  # ldx = 512, K = 300
    q, _ = ops().quantize_act_i8(xt, 0.05)
    ref = oracle.quant_static(as_f32(xt), np.float32(0.05))
    np.testing.assert_array_equal(q.cpu().numpy()[:, :300], ref)
    e = torch.empty((0, 128), dtype=torch.float16, device=DEV)
    q, _ = ops().quantize_act_i8(e, 0.05)
    assert q.shape == (0, 128)


# ----------------------------------------------------------------------------- Hadamard
HAD_CASES = [
    # (n_in, n) ; K comes from the reference's get_hadK table
    (64, 64), (1280, 1280), (3584, 3584), (4096, 4096), (5120, 5120), (8192, 8192),
    (11008, 11008), (14336, 14336), (18944, 19968), (29568, 30720), (768, 768),
]


@pytest.mark.parametrize("n_in,n", HAD_CASES)
@pytest.mark.parametrize("dtype,fp32_had", [(torch.float16, False), (torch.float16, True),
                                            (torch.bfloat16, False), (torch.float32, False)])
def test_hadamard_matches_oracle(had_table, n_in, n, dtype, fp32_had):
    K = had_table["n2k"][n]
    M = 5
    x = make_x(100 + n, (M, n_in))
    xt = to_dev(x, dtype)
    bits = None if K == 1 else to_dev(had_table["words"][K])
    y = ops().hadamard(xt, n, K, bits, fp32_had=fp32_had)
    mode = MODE[dtype]
    mid = 0 if (fp32_had or dtype == torch.float32) else mode
    ref = oracle.hadamard(as_f32(xt), n, K, None if K == 1 else had_table["mats"][K],
                          mid_round=mid, out_round=mode)
    np.testing.assert_array_equal(as_f32(y), ref)


@pytest.mark.parametrize("n_in,n", [(5120, 5120), (18944, 19968), (8192, 8192), (11008, 11008)])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("split", [False, True])
def test_hadamard_quant_fused(had_table, n_in, n, dtype, split):
    K = had_table["n2k"][n]
    M = 6
    x = make_x(200 + n, (M, n_in))
    xt = to_dev(x, dtype)
    bits = None if K == 1 else to_dev(had_table["words"][K])
    sel = (np.arange(M) % 2).astype(np.uint8)
    mode = MODE[dtype]
    rot = oracle.hadamard(as_f32(xt), n, K, None if K == 1 else had_table["mats"][K],
                          mid_round=mode, out_round=mode)
    s0 = np.float32(np.abs(rot).max() / 127.0)
    s1 = np.float32(s0 * 0.6)
    q, x0 = ops().hadamard_quant_i8(xt, n, K, bits, float(s0), float(s1), row_sel=to_dev(sel),
                                    skip_col0=split)
    ref = oracle.quant_static(rot, s0, scale1=s1, row_sel=sel)
    if split:
        ref[:, 0] = 0
        np.testing.assert_array_equal(x0.cpu().numpy(), rot[:, 0])
    got = q.cpu().numpy()
    np.testing.assert_array_equal(got[:, :n], ref)
    assert not got[:, n:].any()


def test_hadamard_golden_reference(had_table, golden_dir):
    """HIP vs the reference's own outputs (tests/golden/hadamard_fwd.npz)."""
    g = np.load(f"{golden_dir}/hadamard_fwd.npz")
    for n in [64, 1280, 3584, 4096, 5120, 11008, 14336, 19968, 30720]:
        K = had_table["n2k"][n]
        x = make_x(100 + n, (2 if n <= 5120 else 1, n))
        bits = None if K == 1 else to_dev(had_table["words"][K])
        y32 = ops().hadamard(to_dev(x), n, K, bits)
        np.testing.assert_array_equal(as_f32(y32), g[f"cuda_{n}"])
        y16 = ops().hadamard(to_dev(x, torch.float16), n, K, bits)
        np.testing.assert_array_equal(as_f32(y16), g[f"cuda16_{n}"].astype(np.float32))


# ----------------------------------------------------------------------------- weights
def test_pack_unpack_wire_format(golden_dir):
    g = np.load(f"{golden_dir}/pack_i4.npz")
    q = to_dev(g["q"])
    packed = ops().pack_i4(q)
    np.testing.assert_array_equal(packed.cpu().numpy(), g["packed"])
    np.testing.assert_array_equal(packed.cpu().numpy(), oracle.pack_i4(g["q"]))
    back = ops().unpack_i4(packed)
    np.testing.assert_array_equal(back.cpu().numpy(), g["unpacked"].astype(np.int8))


@pytest.mark.parametrize("bits", [4, 8])
@pytest.mark.parametrize("dtype", DTYPES)
def test_weight_levels(bits, dtype):
    W = make_w(7, (40, 300))
    scale, levels = oracle.wquant_sym(W, bits=bits)
    Wq = (levels.astype(np.float32) * scale[:, None]).astype(np.float32)
    wt = to_dev(Wq, dtype)
    q = ops().weight_levels(wt, to_dev(scale), bits)
    np.testing.assert_array_equal(q.cpu().numpy(), levels)


# ----------------------------------------------------------------------------- GEMM
def _rand_levels(seed, shape, bits):
    rs = np.random.RandomState(seed)
    lo, hi = -(1 << (bits - 1)), (1 << (bits - 1)) - 1
    q = rs.randint(lo, hi + 1, size=shape).astype(np.int8)
    q.reshape(-1)[:4] = [lo, hi, lo, hi]
    return q


GEMM_SHAPES = [(16, 16, 128), (128, 128, 256), (37, 50, 384), (256, 384, 1280), (130, 132, 640),
               (768, 512, 3584), (1, 3584, 128), (300, 48, 19968), (64, 96, 30720)]


@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
@pytest.mark.parametrize("bits", [4, 8])
def test_gemm_int32_exact(M, N, K, bits):
    a = np.random.RandomState(M + N + K).randint(-128, 128, size=(M, K)).astype(np.int8)
    a[0, :8] = [-128, 127, -128, 127, -128, -128, 127, 127]
    w = _rand_levels(N * 7 + K, (N, K), bits)
    img = ops().prepack(to_dev(w), bits)
    acc = ops().gemm_w4a8_i32(to_dev(a), img, bits, N)
    ref = oracle.gemm_i32(a, w)
    np.testing.assert_array_equal(acc.cpu().numpy(), ref)


def test_gemm_transpose_detecting():
    """A = identity-like rows against an ASYMMETRIC W catches row/col swaps."""
    M = N = 64
    K = 128
    a = np.zeros((M, K), dtype=np.int8)
    a[np.arange(M), np.arange(M)] = 1
    w = np.zeros((N, K), dtype=np.int8)
    for n in range(N):
        for k in range(K):
            w[n, k] = ((3 * n + 5 * k) % 15) - 7
    acc = ops().gemm_w4a8_i32(to_dev(a), ops().prepack(to_dev(w), 4), 4, N)
    np.testing.assert_array_equal(acc.cpu().numpy(), w[:, :M].T.astype(np.int32))


def test_gemm_worst_case_accumulator_magnitude():
    """All -128 x all -8 over the largest supported K stays exact in int32."""
    M, N, K = 16, 16, 30720
    a = np.full((M, K), -128, dtype=np.int8)
    w = np.full((N, K), -8, dtype=np.int8)
    acc = ops().gemm_w4a8_i32(to_dev(a), ops().prepack(to_dev(w), 4), 4, N)
    assert int(acc.max()) == int(acc.min()) == 128 * 8 * K


@pytest.mark.parametrize("out_dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(48, 80, 256), (130, 36, 1280), (768, 256, 3584)])
def test_gemm_dequant_epilogue(out_dtype, M, N, K):
    rs = np.random.RandomState(M * N)
    a = rs.randint(-128, 128, size=(M, K)).astype(np.int8)
    w = _rand_levels(11, (N, K), 4)
    s_w = (0.001 + rs.rand(N) * 0.004).astype(np.float32)
    bias = (rs.randn(N) * 0.1).astype(np.float32)
    sel = (np.arange(M) >= M // 3).astype(np.uint8)
    x0 = rs.randn(M).astype(np.float32)
    w0 = (rs.randn(N) * 0.02).astype(np.float32)
    sx0, sx1 = 0.031, 0.0077
    img = ops().prepack(to_dev(w), 4, zero_col0=True)
    y = ops().gemm_w4a8(to_dev(a), img, 4, N, sx0, to_dev(s_w), s_x1=sx1, row_sel=to_dev(sel),
                        bias=to_dev(bias), x0=to_dev(x0), w0=to_dev(w0), out_dtype=out_dtype)
    w_eff = w.copy()
    w_eff[:, 0] = 0
    acc = oracle.gemm_i32(a, w_eff)
    ref = oracle.epilogue(acc, np.float32(sx0), s_w, bias=bias, sx1=np.float32(sx1), row_sel=sel,
                          x0=x0, w0=w0)
    ref = oracle.round_to(ref, MODE[out_dtype])
    np.testing.assert_array_equal(as_f32(y), ref)


def test_gemm_plain_epilogue_no_optional_terms():
    M, N, K = 64, 64, 512
    rs = np.random.RandomState(5)
    a = rs.randint(-128, 128, size=(M, K)).astype(np.int8)
    w = _rand_levels(12, (N, K), 4)
    s_w = (0.001 + rs.rand(N) * 0.004).astype(np.float32)
    y = ops().gemm_w4a8(to_dev(a), ops().prepack(to_dev(w), 4), 4, N, 0.02, to_dev(s_w))
    ref = oracle.round_to(oracle.epilogue(oracle.gemm_i32(a, w), np.float32(0.02), s_w), 1)
    np.testing.assert_array_equal(as_f32(y), ref)


# ----------------------------------------------------------------------------- observers
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape,col_begin", [((50, 24), 0), ((300, 3584), 0), ((64, 5120), 1),
                                             ((1, 9), 0)])
def test_minmax(dtype, shape, col_begin):
    x = make_x(9, shape)
    x[0, -1] = -0.0
    xt = to_dev(x, dtype)
    xr = as_f32(xt)[:, col_begin:]
    mn, mx = ops().minmax_channels(xt, col_begin)
    rmn, rmx = oracle.minmax_channels(xr)
    np.testing.assert_array_equal(mn.cpu().numpy(), rmn)
    np.testing.assert_array_equal(mx.cpu().numpy(), rmx)
    t = ops().minmax_tensor(xt, col_begin).cpu().numpy()
    assert t[0] == rmn.min() and t[1] == rmx.max()


# ----------------------------------------------------------------------------- errors
def test_errors_are_loud():
    from mquant_amd._lib import MQuantHipError
    with pytest.raises(MQuantHipError):
        ops().quantize_act_i8(torch.zeros(4, 16), 0.1)  # CPU tensor: no fallback
    a = torch.zeros((4, 100), dtype=torch.int8, device=DEV)  # K_pad not a multiple of 128
    with pytest.raises(MQuantHipError):
        ops().gemm_w4a8_i32(a, torch.zeros(4096, dtype=torch.uint8, device=DEV), 4, 16)
    with pytest.raises(MQuantHipError):
        ops().hadamard(torch.zeros((2, 24), dtype=torch.float16, device=DEV), 24, 5, None)


@pytest.mark.parametrize("w_bits", [4, 8])
def test_every_tile_shape_and_split_factor_is_exact(w_bits):
    """All template instantiations behind dispatch_tile (the plan picks among them by shape), with
    and without split-K, on a ragged problem: int32 accumulators and the fp16 epilogue."""
    from mquant_amd import ops
    M, N, K = 300, 520, 1408
    rng = np.random.default_rng(7)
    a = rng.integers(-128, 128, size=(M, K), dtype=np.int8)
    lim = 8 if w_bits == 4 else 128
    w = rng.integers(-lim, lim, size=(N, K), dtype=np.int8)
    s_w = rng.uniform(0.001, 0.01, size=N).astype(np.float32)
    bias = rng.normal(size=N).astype(np.float32)
    acc_ref = oracle.gemm_i32(a, w)
    y_ref = oracle.round_to(oracle.epilogue(acc_ref, np.float32(0.02), s_w, bias=bias), 1)
    at, img = to_dev(a), ops.prepack(to_dev(w), w_bits)
    swt, bt = to_dev(s_w), to_dev(bias)
    ops.splitk_workspace(torch.device(DEV), 64 << 20)
    try:
        for tile in (0, 1, 2, 3, 4, 5, 10, 11, 12, 13, 26, 31, 35):
            for splits in (1, 3):
                ops.gemm_debug_force(tile, splits)
                acc = ops.gemm_w4a8_i32(at, img, w_bits, N)
                np.testing.assert_array_equal(acc.cpu().numpy(), acc_ref, err_msg=f"tile {tile} splits {splits}")
                y = ops.gemm_w4a8(at, img, w_bits, N, 0.02, swt, bias=bt)
                np.testing.assert_array_equal(y.float().cpu().numpy(), y_ref, err_msg=f"tile {tile} splits {splits}")
    finally:
        ops.gemm_debug_force(-1, 0)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K,splits", [(130, 200, 640, 1), (768, 256, 19968, 0), (64, 96, 256, 3)])
def test_gemm_residual_epilogue_equals_torch_add(dtype, M, N, K, splits):
    """hidden + linear(x) in one launch == the GEMM followed by torch's add (same roundings), also
    in place and through the split-K reduce."""
    from mquant_amd import ops
    rng = np.random.default_rng(M + N)
    a = to_dev(rng.integers(-128, 128, size=(M, K), dtype=np.int8))
    w = to_dev(rng.integers(-8, 8, size=(N, K), dtype=np.int8))
    img = ops.prepack(w, 4)
    s_w = to_dev(rng.uniform(0.001, 0.01, size=N).astype(np.float32))
    bias = to_dev(rng.normal(size=N).astype(np.float32))
    res = to_dev(rng.normal(size=(M, N)).astype(np.float32) * 3).to(dtype)
    sel = to_dev((np.arange(M) % 2).astype(np.uint8))
    try:
        ops.gemm_debug_force(-1 if splits == 0 else 10, splits)
        y = ops.gemm_w4a8(a, img, 4, N, 0.02, s_w, s_x1=0.03, row_sel=sel, bias=bias, out_dtype=dtype)
        want = res + y
        got = ops.gemm_w4a8_residual(a, img, 4, N, 0.02, s_w, res, s_x1=0.03, row_sel=sel, bias=bias)
        assert got.dtype == dtype and torch.equal(got, want)
        inplace = res.clone()
        ops.gemm_w4a8_residual(a, img, 4, N, 0.02, s_w, inplace, s_x1=0.03, row_sel=sel, bias=bias, out=inplace)
        assert torch.equal(inplace, want)
    finally:
        ops.gemm_debug_force(-1, 0)
