"""GPU parity of the TILED activation layout (MQ_LD_TILED) and of the wave-specialised GEMM kernels
(csrc/gemm_ws.hip, V_MFMA_I32_32X32X32_I8) that consume it.

Bar as everywhere: int8 levels and int32 accumulators bit-exact against the CPU oracle, fp outputs
bit-exact against the oracle's single-rounded fp32 epilogue."""
import numpy as np
import pytest
import torch

import oracle
from golden_inputs import make_w, make_x

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
DTYPES = [torch.float16, torch.bfloat16, torch.float32]
MODE = {torch.float16: 1, torch.bfloat16: 2, torch.float32: 0}
WS_TILES = (40, 41, 42, 43, 44, 45, 46, 47, 48, 50, 51, 52, 53, 54)     # 44-48: V_MFMA_I32_16X16X64_I8 in the math waves (round 5); 50-54: those with the slab-free epilogue (round 6)
W4_ONLY = (42, 46, 52)


def ops():
    from mquant_amd import ops as o
    return o


def to_dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(DEV)


def test_layout_formula_of_the_header():
    """include/mquant_hip.h: byte offset of (m, k) = ((m/16)(K_pad/64) + k/64) 1024 + (((k/16)%4) 16 + m%16) 16 + k%16."""
    o = ops()
    M, K_pad = 37, 256
    a = torch.arange(M * K_pad, dtype=torch.int32).reshape(M, K_pad).to(torch.int8).to(DEV)
    t = o.TiledAct.from_rows(a)
    flat = t.data.reshape(-1).cpu().numpy()
    ar = a.cpu().numpy()
    for m, k in [(0, 0), (5, 17), (36, 255), (16, 64), (31, 130), (15, 63), (32, 192)]:
        off = ((m // 16) * (K_pad // 64) + k // 64) * 1024 + (((k // 16) % 4) * 16 + m % 16) * 16 + k % 16
        assert flat[off] == ar[m, k], (m, k)
    assert torch.equal(t.to_rows(), a)


# ------------------------------------------------------------------------------------- quantizers
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(1, 16), (7, 100), (33, 1176), (64, 3584), (130, 640)])
def test_static_quantizer_tiled_equals_oracle(dtype, shape):
    x = make_x(3, shape, outlier_gain=30.0)
    xt = to_dev(x, dtype)
    M, K = shape
    sel = to_dev((np.arange(M) % 3 == 0).astype(np.uint8))
    q, x0 = ops().quantize_act_i8(xt, 0.0371, 0.0113, row_sel=sel, skip_col0=True, tiled=True)
    assert isinstance(q, ops().TiledAct) and q.K_pad == (K + 127) // 128 * 128
    xr = xt.float().cpu().numpy()
    ref = np.where(sel.cpu().numpy()[:, None] != 0, oracle.quant_static(xr, np.float32(0.0113)),
                   oracle.quant_static(xr, np.float32(0.0371)))
    ref[:, 0] = 0
    got = q.to_rows().cpu().numpy()
    np.testing.assert_array_equal(got[:, :K], ref)
    assert not got[:, K:].any()
    np.testing.assert_array_equal(x0.cpu().numpy(), xr[:, 0])
    q2, _ = ops().quantize_act_i8(xt, 0.0371, 0.0113, row_sel=sel, skip_col0=True)
    assert torch.equal(q.to_rows(), q2)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(16, 64), (45, 1296), (768, 3584), (1030, 1280), (3, 3600)])
def test_static_quantizer_tiled_next_to_half_integers(dtype, shape):
    """The per-row-scale kernel of the tiled image (K % 16 == 0, aligned rows) multiplies by 1 / s and takes the IEEE quotient
    only where the two can round differently: quotients ON and one ulp either side of every half-integer, saturating values,
    zeros, pad columns (K < K_pad) and a ragged last row tile must give the oracle's levels (uniform.py:20-33)."""
    M, K = shape
    rng = np.random.default_rng(M * 7 + K)
    s0, s1 = np.float32(0.0371), np.float32(0.00931)
    sel = (np.arange(M) % 2 == 1).astype(np.uint8)
    s_row = np.where(sel != 0, s1, s0).astype(np.float32)[:, None]
    half = (rng.integers(-131, 131, size=(M, K)).astype(np.float32) + np.float32(0.5)) * s_row
    nudge = rng.integers(-2, 3, size=(M, K))
    x = half.copy()
    for d in (-2, -1, 1, 2):
        tgt = np.float32(np.inf) if d > 0 else np.float32(-np.inf)
        step = x.copy()
        for _ in range(abs(d)):
            step = np.nextafter(step, tgt)
        x = np.where(nudge == d, step, x)
    plain = make_x(11, shape, outlier_gain=30.0)
    x = np.where(rng.random((M, K)) < 0.5, x, plain).astype(np.float32)
    x[:, 1::97] = 0.0
    xt = to_dev(x, dtype)
    q, _ = ops().quantize_act_i8(xt, float(s0), float(s1), row_sel=to_dev(sel), tiled=True)
    xr = xt.float().cpu().numpy()
    ref = np.where(sel[:, None] != 0, oracle.quant_static(xr, s1), oracle.quant_static(xr, s0))
    got = q.to_rows().cpu().numpy()
    np.testing.assert_array_equal(got[:, :K], ref)
    assert not got[:, K:].any()


@pytest.mark.parametrize("n_in,n", [(5120, 5120), (18944, 19968), (1280, 1280), (700, 768)])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_hadamard_quant_tiled_equals_row_major(had_table, n_in, n, dtype):
    o = ops()
    K = {5120: 40, 19968: 156, 1280: 20, 768: 12}[n]
    bits = to_dev(had_table["words"][K])
    for M in (5, 48, 131):
        x = to_dev(make_x(M + n, (M, n_in)), dtype)
        sel = to_dev((np.arange(M) >= M // 2).astype(np.uint8))
        rows, x0r = o.hadamard_quant_i8(x, n, K, bits, 0.05, 0.02, row_sel=sel, skip_col0=True)
        tiled, x0t = o.hadamard_quant_i8(x, n, K, bits, 0.05, 0.02, row_sel=sel, skip_col0=True, tiled=True)
        assert torch.equal(tiled.to_rows(), rows) and torch.equal(x0r, x0t)
    # the row-major result itself is pinned to the oracle in test_gpu_kernels.py


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16, torch.float32])
def test_fused_activation_rmsn_and_dynamic_quantizers_tiled(had_table, dtype):
    o = ops()
    M, n_in, n, K = 70, 18944 // 4, 19968 // 4, 156       # 4736 -> 4992 = 156 x 32
    bits = to_dev(had_table["words"][K])
    g = to_dev(make_x(1, (M, n_in)), dtype)
    u = to_dev(make_x(2, (M, n_in)), dtype)
    rows, _ = o.act_hadamard_quant_i8(g, u, o.ACT_SILU_MUL, n, K, bits, 0.04)
    tiled, _ = o.act_hadamard_quant_i8(g, u, o.ACT_SILU_MUL, n, K, bits, 0.04, tiled=True)
    assert torch.equal(tiled.to_rows(), rows)
    x = to_dev(make_x(3, (M, 1280)), dtype)
    rows, _ = o.rmsn_quantize_i8(x, 1280.0, 1e-6, 0.03)
    tiled, _ = o.rmsn_quantize_i8(x, 1280.0, 1e-6, 0.03, tiled=True)
    assert torch.equal(tiled.to_rows(), rows)
    rows, s_r, x0_r = o.quantize_act_dyn_i8(x, 8, 0.9, skip_col0=True)
    tiled, s_t, x0_t = o.quantize_act_dyn_i8(x, 8, 0.9, skip_col0=True, tiled=True)
    assert torch.equal(tiled.to_rows(), rows) and torch.equal(s_r, s_t) and torch.equal(x0_r, x0_t)


# ------------------------------------------------------------------------------------------ GEMM
def _levels(seed, shape, bits):
    lim = 1 << (bits - 1)
    q = np.random.default_rng(seed).integers(-lim, lim, size=shape, dtype=np.int8)
    q.reshape(-1)[:4] = [-lim, lim - 1, -lim, lim - 1]
    return q


@pytest.mark.parametrize("w_bits", [4, 8])
@pytest.mark.parametrize("M,N,K", [(300, 520, 1408), (96, 128, 128), (17, 40, 256), (768, 1280, 640)])
def test_every_wave_specialised_tile_is_exact(w_bits, M, N, K):
    """All instantiations behind dispatch_ws, with and without split-K, ragged edges in M and N:
    int32 accumulators and the fp16 epilogue against the oracle."""
    o = ops()
    rng = np.random.default_rng(M + N + K)
    a = rng.integers(-128, 128, size=(M, K), dtype=np.int8)
    a[0, :8] = [-128, 127, -128, 127, -128, -128, 127, 127]
    w = _levels(7, (N, K), w_bits)
    s_w = rng.uniform(0.001, 0.01, size=N).astype(np.float32)
    bias = rng.normal(size=N).astype(np.float32)
    acc_ref = oracle.gemm_i32(a, w)
    y_ref = oracle.round_to(oracle.epilogue(acc_ref, np.float32(0.02), s_w, bias=bias), 1)
    at = o.TiledAct.from_rows(to_dev(a))
    img = o.prepack(to_dev(w), w_bits)
    swt, bt = to_dev(s_w), to_dev(bias)
    o.splitk_workspace(torch.device(DEV), 64 << 20)
    try:
        for tile in WS_TILES:
            if w_bits == 8 and tile in W4_ONLY:
                continue                      # these exist for int4 weights only
            for splits in (1, 3):
                o.gemm_debug_force(tile, splits)
                acc = o.gemm_w4a8_i32(at, img, w_bits, N)
                np.testing.assert_array_equal(acc.cpu().numpy(), acc_ref, err_msg=f"tile {tile} splits {splits}")
                y = o.gemm_w4a8(at, img, w_bits, N, 0.02, swt, bias=bt)
                np.testing.assert_array_equal(y.float().cpu().numpy(), y_ref, err_msg=f"tile {tile} splits {splits}")
    finally:
        o.gemm_debug_force(-1, 0)


def test_transpose_detecting_and_worst_case_magnitude_tiled():
    o = ops()
    M = N = 64
    K = 128
    a = np.zeros((M, K), dtype=np.int8)
    a[np.arange(M), np.arange(M)] = 1
    w = np.zeros((N, K), dtype=np.int8)
    for n in range(N):
        for k in range(K):
            w[n, k] = ((3 * n + 5 * k) % 15) - 7
    acc = o.gemm_w4a8_i32(o.TiledAct.from_rows(to_dev(a)), o.prepack(to_dev(w), 4), 4, N)
    np.testing.assert_array_equal(acc.cpu().numpy(), w[:, :M].T.astype(np.int32))
    K = 30720
    a = np.full((16, K), -128, dtype=np.int8)
    w = np.full((16, K), -8, dtype=np.int8)
    acc = o.gemm_w4a8_i32(o.TiledAct.from_rows(to_dev(a)), o.prepack(to_dev(w), 4), 4, 16)
    assert int(acc.max()) == int(acc.min()) == 128 * 8 * K


@pytest.mark.parametrize("out_dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(48, 80, 256), (130, 36, 1280), (768, 256, 3584)])
def test_dequant_epilogue_with_every_optional_term_tiled(out_dtype, M, N, K):
    o = ops()
    rs = np.random.RandomState(M * N)
    a = rs.randint(-128, 128, size=(M, K)).astype(np.int8)
    w = _levels(11, (N, K), 4)
    s_w = (0.001 + rs.rand(N) * 0.004).astype(np.float32)
    bias = (rs.randn(N) * 0.1).astype(np.float32)
    sel = (np.arange(M) >= M // 3).astype(np.uint8)
    x0 = rs.randn(M).astype(np.float32)
    w0 = (rs.randn(N) * 0.02).astype(np.float32)
    sx0, sx1 = 0.031, 0.0077
    img = o.prepack(to_dev(w), 4, zero_col0=True)
    at = o.TiledAct.from_rows(to_dev(a))
    w_eff = w.copy()
    w_eff[:, 0] = 0
    acc = oracle.gemm_i32(a, w_eff)
    ref = oracle.round_to(oracle.epilogue(acc, np.float32(sx0), s_w, bias=bias, sx1=np.float32(sx1), row_sel=sel,
                                          x0=x0, w0=w0), MODE[out_dtype])
    try:
        for tile in (-1, 40, 41, 43, 44, 45, 47):
            o.gemm_debug_force(tile, 0)
            y = o.gemm_w4a8(at, img, 4, N, sx0, to_dev(s_w), s_x1=sx1, row_sel=to_dev(sel), bias=to_dev(bias),
                            x0=to_dev(x0), w0=to_dev(w0), out_dtype=out_dtype)
            np.testing.assert_array_equal(y.float().cpu().numpy(), ref, err_msg=f"tile {tile}")
    finally:
        o.gemm_debug_force(-1, 0)


@pytest.mark.parametrize("dtype", DTYPES)
def test_residual_and_row_scale_entry_points_tiled(dtype):
    o = ops()
    M, N, K = 130, 200, 640
    rng = np.random.default_rng(3)
    a = to_dev(rng.integers(-128, 128, size=(M, K), dtype=np.int8))
    at = o.TiledAct.from_rows(a)
    img = o.prepack(to_dev(rng.integers(-8, 8, size=(N, K), dtype=np.int8)), 4)
    s_w = to_dev(rng.uniform(0.001, 0.01, size=N).astype(np.float32))
    bias = to_dev(rng.normal(size=N).astype(np.float32))
    res = to_dev(rng.normal(size=(M, N)).astype(np.float32) * 3).to(dtype)
    s_rows = to_dev(rng.uniform(0.01, 0.05, size=M).astype(np.float32))
    want = o.gemm_w4a8_residual(a, img, 4, N, 0.02, s_w, res, bias=bias)
    got = o.gemm_w4a8_residual(at, img, 4, N, 0.02, s_w, res, bias=bias)
    assert torch.equal(got, want)
    want = o.gemm_w4a8_rowscale(a, img, 4, N, s_rows, s_w, bias=bias, out_dtype=dtype)
    got = o.gemm_w4a8_rowscale(at, img, 4, N, s_rows, s_w, bias=bias, out_dtype=dtype)
    assert torch.equal(got, want)


def test_symmetric_kernels_accept_the_tiled_layout():
    """The 256 x 256 kernels (gate_up, split-K down_proj) read tiled activations too."""
    o = ops()
    M, N, K = 300, 520, 1408
    rng = np.random.default_rng(9)
    a = rng.integers(-128, 128, size=(M, K), dtype=np.int8)
    w = _levels(5, (N, K), 4)
    acc_ref = oracle.gemm_i32(a, w)
    at = o.TiledAct.from_rows(to_dev(a))
    img = o.prepack(to_dev(w), 4)
    o.splitk_workspace(torch.device(DEV), 64 << 20)
    try:
        for tile in (1, 3, 13, 14, 20, 15, 16, 17, 18, 19, 10, 26, 31, 35, 2):
            for splits in (1, 3):
                o.gemm_debug_force(tile, splits)
                np.testing.assert_array_equal(o.gemm_w4a8_i32(at, img, 4, N).cpu().numpy(), acc_ref,
                                              err_msg=f"tile {tile} splits {splits}")
    finally:
        o.gemm_debug_force(-1, 0)


def test_engine_uses_the_tiled_layout_end_to_end(had_table):
    """W4A8Linear.forward (quantize -> GEMM through the workspace) equals the oracle composition."""
    from mquant_amd import engine
    o = ops()
    assert engine.ACT_LAYOUT == "tiled"
    M, n_in, n, N, K = 100, 1216, 1280, 200, 20
    hk = had_table["mats"][K]
    x = to_dev(make_x(5, (M, n_in)), torch.float16)
    W = make_w(6, (N, n))
    s_w, levels = oracle.wquant_sym(W, bits=4)
    rot = oracle.hadamard(x.float().cpu().numpy(), n, K, hk, mid_round=1, out_round=1)
    s_x = float(np.float32(np.abs(rot).max() / 127.0))
    q_ref = oracle.quant_static(rot, np.float32(s_x))
    y_ref = oracle.round_to(oracle.epilogue(oracle.gemm_i32(q_ref, levels), np.float32(s_x), s_w), 1)
    lin = engine.W4A8Linear(to_dev(levels), to_dev(s_w), 4, None, s_x,
                            had=engine.HadamardSpec(n, K, to_dev(had_table["words"][K])), in_features=n_in)
    y = lin(x)
    np.testing.assert_array_equal(y.float().cpu().numpy(), y_ref)
    a, _ = lin.quantize(x)
    assert isinstance(a, o.TiledAct)
    np.testing.assert_array_equal(a.to_rows().cpu().numpy()[:, :n], q_ref)


# ------------------------------------------------------------------- prepared Hadamard descriptor
@pytest.mark.parametrize("n_in,n,K", [(18944, 19968, 156), (5120, 5120, 40), (11008, 11008, 172), (1280, 1280, 20),
                                      (8192, 8192, 1), (3456, 3456, 108), (4480, 4480, 140), (6656, 6656, 52)])
@pytest.mark.parametrize("dtype,fp32_had", [(torch.float16, False), (torch.bfloat16, False), (torch.float16, True),
                                            (torch.float32, False)])
def test_prepared_descriptor_gives_identical_results(had_table, n_in, n, K, dtype, fp32_had):
    """mq_hadamard_prepare + MQ_HAD_PREPARED (lane masks, 5 x 2 / 3 x 2 / 1 x 4 units) == plain sign words,
    bit for bit, for the rotated activations and for the fused quantizer in both layouts."""
    o = ops()
    words = to_dev(had_table["words"][K]) if K > 1 else None
    desc = o.hadamard_prepare(words, K) if K > 1 else None
    x = to_dev(make_x(n + K, (9, n_in)), dtype)
    y0 = o.hadamard(x, n, K, words, fp32_had)
    y1 = o.hadamard(x, n, K, desc, fp32_had)
    assert torch.equal(y0, y1)
    sel = to_dev((np.arange(9) % 2).astype(np.uint8))
    q0, z0 = o.hadamard_quant_i8(x, n, K, words, 0.05, 0.021, fp32_had=fp32_had, row_sel=sel, skip_col0=True)
    q1, z1 = o.hadamard_quant_i8(x, n, K, desc, 0.05, 0.021, fp32_had=fp32_had, row_sel=sel, skip_col0=True)
    q2, _ = o.hadamard_quant_i8(x, n, K, desc, 0.05, 0.021, fp32_had=fp32_had, row_sel=sel, skip_col0=True, tiled=True)
    assert torch.equal(q0, q1) and torch.equal(z0, z1) and torch.equal(q2.to_rows(), q0)


def test_prepared_descriptor_against_the_oracle(had_table):
    o = ops()
    n_in, n, K = 18944, 19968, 156
    x = to_dev(make_x(77, (6, n_in)), torch.float16)
    desc = o.hadamard_prepare(to_dev(had_table["words"][K]), K)
    rot = oracle.hadamard(x.float().cpu().numpy(), n, K, had_table["mats"][K], mid_round=1, out_round=1)
    y = o.hadamard(x, n, K, desc)
    np.testing.assert_array_equal(y.float().cpu().numpy(), rot)
    q, _ = o.hadamard_quant_i8(x, n, K, desc, 0.05)
    np.testing.assert_array_equal(q.cpu().numpy()[:, :n], oracle.quant_static(rot, np.float32(0.05)))


@pytest.mark.parametrize("n_in,n,K,M", [(18944, 19968, 156, 1300), (5120, 5120, 40, 2500), (11008, 11008, 172, 1100)])
def test_row_loop_of_the_resident_hadamard_workgroups(had_table, n_in, n, K, M):
    """The Hadamard launch starts only the resident workgroups (512 of eight waves, 1024 of four) and their row loop hands
    out the rest: with more rows than that, every row must equal the same row computed in a launch of its own block of 64
    rows (no loop there), in both output layouts."""
    o = ops()
    desc = o.hadamard_prepare(to_dev(had_table["words"][K]), K)
    x = to_dev(make_x(M + K, (M, n_in)), torch.float16)
    sel = to_dev((np.arange(M) % 3 == 1).astype(np.uint8))
    full, x0 = o.hadamard_quant_i8(x, n, K, desc, 0.05, 0.021, row_sel=sel, skip_col0=True)
    tiled, x0t = o.hadamard_quant_i8(x, n, K, desc, 0.05, 0.021, row_sel=sel, skip_col0=True, tiled=True)
    assert torch.equal(tiled.to_rows(), full) and torch.equal(x0, x0t)
    for r0 in (0, 448, 512, 960, M - 64):
        part, x0p = o.hadamard_quant_i8(x[r0:r0 + 64], n, K, desc, 0.05, 0.021, row_sel=sel[r0:r0 + 64], skip_col0=True)
        assert torch.equal(part, full[r0:r0 + 64]) and torch.equal(x0p, x0[r0:r0 + 64]), r0


@pytest.mark.parametrize("M,N,K,tile", [(768, 24576, 256, 14), (700, 33000, 384, 14), (700, 33000, 384, 20), (512, 40960, 128, 19)])
def test_persistent_ping_pong_launch_with_more_tiles_than_cus(M, N, K, tile):
    """More work ids than CUs: the ping-pong kernel starts 256 workgroups that walk ids b, b + 256, ... (gemm_pp.hip); every
    accumulator and the dequantised output must equal the wave-specialised kernel's (one workgroup per tile)."""
    o = ops()
    rng = np.random.default_rng(M + N + K)
    a = to_dev(rng.integers(-128, 128, size=(M, K), dtype=np.int8))
    w = to_dev(_levels(3, (N, K), 4))
    s_w = to_dev(rng.uniform(0.001, 0.01, size=N).astype(np.float32))
    bias = to_dev(rng.normal(size=N).astype(np.float32))
    at, img = o.TiledAct.from_rows(a), o.prepack(w, 4)
    try:
        o.gemm_debug_force(40, 1)
        acc_ref = o.gemm_w4a8_i32(at, img, 4, N)
        y_ref = o.gemm_w4a8(at, img, 4, N, 0.02, s_w, bias=bias)
        o.gemm_debug_force(tile, 1)
        assert torch.equal(o.gemm_w4a8_i32(at, img, 4, N), acc_ref)
        assert torch.equal(o.gemm_w4a8(at, img, 4, N, 0.02, s_w, bias=bias), y_ref)
    finally:
        o.gemm_debug_force(-1, 0)


@pytest.mark.parametrize("M,N,K", [(768, 1100, 512), (1000, 300, 384)])
def test_every_m_grouping_of_the_xcd_mapping_is_a_bijection(M, N, K):
    """tile_of_block with xm m-groups (gemm_common.h): forced through the debug hook (bits 8.. of
    `splits`) for every divisor of the m-block count; a tile computed twice or never shows up as a
    wrong accumulator."""
    o = ops()
    rng = np.random.default_rng(M + N)
    a = rng.integers(-128, 128, size=(M, K), dtype=np.int8)
    w = _levels(11, (N, K), 4)
    acc_ref = oracle.gemm_i32(a, w)
    at = o.TiledAct.from_rows(to_dev(a))
    img = o.prepack(to_dev(w), 4)
    o.splitk_workspace(torch.device(DEV), 64 << 20)
    try:
        for tile in (40, 41, 42, 43, 44, 45, 46, 47, 48, 50, 51, 52, 53, 54, 3, 1, 14, 15, 16, 17, 18, 19, 20):
            for xm in (1, 2, 3, 4, 6, 8):
                for splits in (1, 2):
                    o.gemm_debug_force(tile, splits | (xm << 8))
                    acc = o.gemm_w4a8_i32(at, img, 4, N)
                    np.testing.assert_array_equal(acc.cpu().numpy(), acc_ref, err_msg=f"tile {tile} xm {xm} splits {splits}")
    finally:
        o.gemm_debug_force(-1, 0)


def test_random_shapes_tiles_and_epilogues_against_the_oracle():
    """Seeded sweep over ragged shapes, every kernel family, split-K, XCD m-groups and the optional
    epilogue terms: int32 accumulators and the dequantised output bit for bit against the oracle."""
    o = ops()
    rng = np.random.default_rng(20261002)
    tiles_w4 = list(WS_TILES) + [1, 3, 13, 14, 15, 16, 17, 18, 19, 20, 2, 10, 26, 31, 35]
    tiles_w8 = [t for t in WS_TILES if t not in W4_ONLY] + [3, 2, 10, 26, 31]
    o.splitk_workspace(torch.device(DEV), 64 << 20)
    try:
        for case in range(96):
            w_bits = 4 if case % 3 else 8
            M = int(rng.integers(1, 700))
            N = int(rng.integers(1, 90)) * 8
            K = int(rng.integers(1, 12)) * 128
            tile = int(rng.choice(tiles_w4 if w_bits == 4 else tiles_w8))
            splits = int(rng.choice([1, 1, 2, 3]))
            xm = int(rng.choice([0, 1, 2, 4]))
            a = rng.integers(-128, 128, size=(M, K), dtype=np.int8)
            w = _levels(case, (N, K), w_bits)
            s_w = rng.uniform(0.001, 0.01, size=N).astype(np.float32)
            bias = rng.normal(size=N).astype(np.float32) if case % 2 else None
            sel = (rng.random(M) < 0.4).astype(np.uint8) if case % 4 == 1 else None
            use_x0 = case % 5 == 2
            x0 = rng.normal(size=M).astype(np.float32) if use_x0 else None
            w0 = rng.normal(size=N).astype(np.float32) if use_x0 else None
            out_dtype = (torch.float16, torch.bfloat16, torch.float32)[case % 3]
            w_eff = w.copy()
            acc_ref = oracle.gemm_i32(a, w_eff)
            y_ref = oracle.round_to(oracle.epilogue(acc_ref, np.float32(0.02), s_w, bias=bias, sx1=np.float32(0.05),
                                                    row_sel=sel, x0=x0, w0=w0), MODE[out_dtype])
            at = o.TiledAct.from_rows(to_dev(a))
            img = o.prepack(to_dev(w), w_bits)
            tag = f"case {case}: M={M} N={N} K={K} w{w_bits} tile {tile} splits {splits} xm {xm} {out_dtype}"
            o.gemm_debug_force(tile, splits | (xm << 8))
            np.testing.assert_array_equal(o.gemm_w4a8_i32(at, img, w_bits, N).cpu().numpy(), acc_ref, err_msg=tag)
            y = o.gemm_w4a8(at, img, w_bits, N, 0.02, to_dev(s_w), s_x1=0.05, row_sel=None if sel is None else to_dev(sel),
                            bias=None if bias is None else to_dev(bias), x0=None if x0 is None else to_dev(x0),
                            w0=None if w0 is None else to_dev(w0), out_dtype=out_dtype)
            np.testing.assert_array_equal(y.float().cpu().numpy(), y_ref, err_msg=tag)
    finally:
        o.gemm_debug_force(-1, 0)


# ---- advisor findings r4 (tests/test_gpu_tiled.py): persistent ping-pong tiles past the CU count, special values in the epilogue ----
@pytest.mark.parametrize("tile", [14, 15, 16, 17, 18, 19])
def test_persistent_ping_pong_tiles_walk_past_the_cu_count_into_ragged_tail_tiles(tile):
    """More work ids than CUs (the persistent loop hands a workgroup a second, third ... tile behind one barrier) with the LAST
    m-block and the LAST n-block ragged: every output against the oracle, so a second tile that is an M- or N-tail tile
    (different clamps, partial stores) is covered for every ping-pong instantiation."""
    o = ops()
    bm, bn = {14: (256, 256), 15: (128, 128), 16: (96, 128), 17: (192, 128), 18: (64, 128), 19: (128, 256)}[tile]
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    M = 2 * bm + bm // 3 + 5                                   # three m-blocks, the last one ragged
    n_blocks = cus // 3 + 3                                     # 3 * n_blocks > CUs
    N = (n_blocks - 1) * bn + 40                                # the last n-block holds 40 channels
    K = 256
    assert 3 * n_blocks > cus
    rng = np.random.default_rng(tile)
    a = rng.integers(-128, 128, size=(M, K), dtype=np.int8)
    w = _levels(tile, (N, K), 4)
    s_w = rng.uniform(0.001, 0.01, size=N).astype(np.float32)
    bias = rng.normal(size=N).astype(np.float32)
    acc_ref = oracle.gemm_i32(a, w)
    at = o.TiledAct.from_rows(to_dev(a))
    img = o.prepack(to_dev(w), 4)
    try:
        o.gemm_debug_force(tile, 1)
        np.testing.assert_array_equal(o.gemm_w4a8_i32(at, img, 4, N).cpu().numpy(), acc_ref)
        for dt in DTYPES:
            y = o.gemm_w4a8(at, img, 4, N, 0.02, to_dev(s_w), bias=to_dev(bias), out_dtype=dt)
            np.testing.assert_array_equal(y.float().cpu().numpy(), oracle.round_to(oracle.epilogue(acc_ref, np.float32(0.02), s_w, bias=bias), MODE[dt]))
    finally:
        o.gemm_debug_force(-1, 0)


@pytest.mark.parametrize("out_dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("tile", [14, 20, 13, 3, 26, 40, 44, 45, 46, 47, 48, 51, 52, 53, 54])
def test_epilogue_overflow_infinity_and_nan_equal_the_oracles_rounding(out_dtype, tile):
    """The straight-line epilogues convert with V_CVT_PK_F16_F32 / V_CVT_PK_BF16_F32 (pack2_f16 / pack2_bf16) where the
    general loop uses the bit-trick conversions: both must agree with the oracle's round-to-nearest-even on results beyond the
    largest finite value (-> inf), on infinities and on NaN (inf - inf through the bias, 0 x inf through the rank-1 term)."""
    o = ops()
    M, N, K = 70, 264, 256
    rng = np.random.default_rng(5)
    a = rng.integers(-128, 128, size=(M, K), dtype=np.int8)
    w = _levels(9, (N, K), 4)
    acc = oracle.gemm_i32(a, w)
    big = np.float32(3.0e38 if out_dtype == torch.bfloat16 else 6.0e4) / np.float32(np.abs(acc).max())
    s_w = np.full(N, 1.0, np.float32)
    s_w[::3] = np.float32(1.13)                                   # some channels end beyond the largest finite value: inf after rounding
    s_w[1::7] = np.float32(4.0)                                   # ... and some beyond fp32's for bf16 / far beyond fp16's
    bias = np.zeros(N, np.float32)
    bias[5] = -np.inf                                             # +inf + (-inf) -> NaN where the product overflowed, -inf elsewhere
    bias[6] = np.inf
    x0 = np.zeros(M, np.float32)
    w0 = np.zeros(N, np.float32)
    x0[3] = np.inf                                                # inf * 0 -> NaN across row 3
    w0[10] = 1.0
    at = o.TiledAct.from_rows(to_dev(a))
    img = o.prepack(to_dev(w), 4)
    want = oracle.round_to(oracle.epilogue(acc, big, s_w, bias=bias, x0=x0, w0=w0), MODE[out_dtype])
    assert np.isinf(want).any() and np.isnan(want).any() and np.isfinite(want).any()
    try:
        o.gemm_debug_force(tile, 1)
        aa = at if tile not in (3, 26) else to_dev(a)
        y = o.gemm_w4a8(aa, img, 4, N, float(big), to_dev(s_w), bias=to_dev(bias), x0=to_dev(x0), w0=to_dev(w0), out_dtype=out_dtype)
        np.testing.assert_array_equal(y.float().cpu().numpy(), want)
    finally:
        o.gemm_debug_force(-1, 0)
