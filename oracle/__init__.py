"""ctypes/numpy front-end of the CPU oracle (``oracle/mq_oracle.c``).

TEST INFRASTRUCTURE ONLY -- importable from ``tests/``, ``__graft_entry__.smoke``
and ``bench.py``'s ``cpu_baseline`` leg.  The product packages (``fake_quant``,
``mquant_amd``) never import this module.

Parity: pinned against outputs of the reference generated in the build container
(``tools/gen_golden.py`` -> ``tests/golden/*.npz``); see ``tests/test_oracle_golden.py``.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libmq_oracle.so")


def build(force: bool = False) -> str:
    """Compile the C restatement with gcc (idempotent)."""
    src = os.path.join(_HERE, "mq_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libmq_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.orc_round_f16.restype = C.c_float
        _lib.orc_round_f16.argtypes = [C.c_float]
        _lib.orc_round_bf16.restype = C.c_float
        _lib.orc_round_bf16.argtypes = [C.c_float]
    return _lib


def _p(a, ctype):
    if a is None:
        return None
    return a.ctypes.data_as(C.POINTER(ctype))


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def round_to(x: np.ndarray, mode: int) -> np.ndarray:
    """mode 0: identity, 1: fp16 round trip, 2: bf16 round trip (RNE)."""
    x = np.array(x, dtype=np.float32, order="C")
    lib().orc_round_array(_p(x, C.c_float), C.c_long(x.size), C.c_int(mode))
    return x


def quant_static(x, scale, zp=None, scale1=None, zp1=None, row_sel=None,
                 lo=-128, hi=127) -> np.ndarray:
    """uniform.py:20-33.  x: (rows, cols) fp32.  scale scalar or (cols,)."""
    x = _f32(x)
    rows, cols = x.shape
    scale = _f32(np.atleast_1d(scale))
    per_channel = int(scale.size > 1)
    zp = _f32(np.zeros_like(scale) if zp is None else np.atleast_1d(zp))
    if scale1 is not None:
        scale1 = _f32(np.atleast_1d(scale1))
        zp1 = _f32(np.zeros_like(scale1) if zp1 is None else np.atleast_1d(zp1))
        row_sel = np.ascontiguousarray(row_sel, dtype=np.uint8)
    else:
        scale1, zp1, row_sel = scale, zp, None
    q = np.empty((rows, cols), dtype=np.int8)
    lib().orc_quant_static(_p(x, C.c_float), C.c_long(rows), C.c_long(cols),
                           _p(scale, C.c_float), _p(zp, C.c_float),
                           _p(scale1, C.c_float), _p(zp1, C.c_float),
                           _p(row_sel, C.c_uint8), C.c_int(per_channel),
                           C.c_int(lo), C.c_int(hi), _p(q, C.c_int8))
    return q


def dequant_static(q, scale, zp=None, scale1=None, zp1=None, row_sel=None) -> np.ndarray:
    q = np.ascontiguousarray(q, dtype=np.int8)
    rows, cols = q.shape
    scale = _f32(np.atleast_1d(scale))
    per_channel = int(scale.size > 1)
    zp = _f32(np.zeros_like(scale) if zp is None else np.atleast_1d(zp))
    if scale1 is not None:
        scale1 = _f32(np.atleast_1d(scale1))
        zp1 = _f32(np.zeros_like(scale1) if zp1 is None else np.atleast_1d(zp1))
        row_sel = np.ascontiguousarray(row_sel, dtype=np.uint8)
    else:
        scale1, zp1, row_sel = scale, zp, None
    out = np.empty((rows, cols), dtype=np.float32)
    lib().orc_dequant_static(_p(q, C.c_int8), C.c_long(rows), C.c_long(cols),
                             _p(scale, C.c_float), _p(zp, C.c_float),
                             _p(scale1, C.c_float), _p(zp1, C.c_float),
                             _p(row_sel, C.c_uint8), C.c_int(per_channel),
                             _p(out, C.c_float))
    return out


def minmax_channels(x):
    x = _f32(x)
    rows, cols = x.shape
    mn = np.empty(cols, dtype=np.float32)
    mx = np.empty(cols, dtype=np.float32)
    lib().orc_minmax_channels(_p(x, C.c_float), C.c_long(rows), C.c_long(cols),
                              _p(mn, C.c_float), _p(mx, C.c_float))
    return mn, mx


def minmax_scale_sym(min_val, max_val, qmin=-128, qmax=127):
    """observer/minmax.py:30-46 (symmetric branch), fp32."""
    mn = np.float32(min_val)
    mx = np.float32(max_val)
    s = np.maximum(np.abs(mn / np.float32(qmin)), np.abs(mx / np.float32(qmax)))
    return np.maximum(s, np.float32(np.finfo(np.float32).eps)).astype(np.float32)


def hadamard(x, n, K, hadK=None, mid_round=0, out_round=0, post_div=False) -> np.ndarray:
    """hadamard_utils.py:115-128 incl. the zero pad of utils.py:465-471."""
    x = _f32(x)
    rows, n_in = x.shape
    out = np.empty((rows, n), dtype=np.float32)
    hk = None if hadK is None else np.ascontiguousarray(np.sign(hadK), dtype=np.int8)
    lib().orc_hadamard(_p(x, C.c_float), C.c_long(rows), C.c_long(n_in), C.c_long(n),
                       C.c_int(K), _p(hk, C.c_int8), C.c_int(mid_round),
                       C.c_int(out_round), C.c_int(int(post_div)), _p(out, C.c_float))
    return out


def pack_i4(q) -> np.ndarray:
    q = np.ascontiguousarray(q, dtype=np.int8)
    rows, cols = q.shape
    out = np.empty((rows, cols // 2), dtype=np.uint8)
    lib().orc_pack_i4(_p(q, C.c_int8), C.c_long(rows), C.c_long(cols), _p(out, C.c_uint8))
    return out


def unpack_i4(p) -> np.ndarray:
    p = np.ascontiguousarray(p, dtype=np.uint8)
    rows, half = p.shape
    out = np.empty((rows, half * 2), dtype=np.int8)
    lib().orc_unpack_i4(_p(p, C.c_uint8), C.c_long(rows), C.c_long(half * 2),
                        _p(out, C.c_int8))
    return out


def gemm_i32(a, w) -> np.ndarray:
    """acc[m][n] = sum_k a[m][k] * w[n][k]; a int8 (M,K), w int8-held int4 (N,K)."""
    a = np.ascontiguousarray(a, dtype=np.int8)
    w = np.ascontiguousarray(w, dtype=np.int8)
    M, K = a.shape
    N = w.shape[0]
    acc = np.empty((M, N), dtype=np.int32)
    lib().orc_gemm_i8i4_i32(_p(a, C.c_int8), _p(w, C.c_int8), C.c_long(M), C.c_long(N),
                            C.c_long(K), _p(acc, C.c_int32))
    return acc


def epilogue(acc, sx0, s_w, bias=None, sx1=None, row_sel=None, x0=None, w0=None, x1=None, w1=None):
    """y = ((float(acc) * s_x) * s_w[n]) + bias[n] + x0[m] * w0[n] [+ x1[m] * w1[n]], one fp32 rounding per operation, the terms
    added in that order (the second rank-1 term: flag combinations that need the epilogue slot twice, csrc/gemm_common.h)."""
    if x1 is not None:
        y = epilogue(acc, sx0, s_w, bias=bias, sx1=sx1, row_sel=row_sel, x0=x0, w0=w0)
        return (y + (_f32(x1).reshape(-1, 1) * _f32(w1)[None, :]).astype(np.float32)).astype(np.float32)
    acc = np.ascontiguousarray(acc, dtype=np.int32)
    M, N = acc.shape
    s_w = _f32(s_w).reshape(-1)
    if np.ndim(sx0) == 1:
        # one activation scale per row (dynamic per-token mode): same operation order, one fp32
        # rounding per numpy operation
        y = (acc.astype(np.float32) * _f32(sx0).reshape(-1, 1)) * s_w[None, :]
        if bias is not None:
            y = y + _f32(bias)[None, :]
        if x0 is not None:
            y = y + _f32(x0).reshape(-1, 1) * _f32(w0)[None, :]
        return y.astype(np.float32)
    bias = _f32(bias)
    x0 = _f32(x0)
    w0 = _f32(w0)
    rs = None if row_sel is None else np.ascontiguousarray(row_sel, dtype=np.uint8)
    out = np.empty((M, N), dtype=np.float32)
    lib().orc_epilogue(_p(acc, C.c_int32), C.c_long(M), C.c_long(N),
                       C.c_float(sx0), C.c_float(sx0 if sx1 is None else sx1),
                       _p(rs, C.c_uint8), _p(s_w, C.c_float), _p(bias, C.c_float),
                       _p(x0, C.c_float), _p(w0, C.c_float), _p(out, C.c_float))
    return out


def gemm_wgroup(a, w, s_wg, g, sx0=1.0, sx1=None, row_sel=None, sx_rows=None, s_xg=None, bias=None, want_acc=False):
    """Group-wise weight scales (``--w_groupsize``, reference gptq/gptq_utils.py:263-273 + quant_utils.py:384):
    y = (sum_g (float(acc_g) [* s_xg[m][g]]) * s_wg[g][n]) * s_x(m) + bias[n]; a int8 (M, K), w levels (N, K), s_wg fp32 (K / g, N).
    Returns y (and the exact per-group accumulators [M, G, N] when ``want_acc``)."""
    a = np.ascontiguousarray(a, dtype=np.int8)
    w = np.ascontiguousarray(w, dtype=np.int8)
    M, K = a.shape
    N = w.shape[0]
    assert w.shape[1] == K and K % g == 0
    s_wg = _f32(s_wg)
    assert s_wg.shape == (K // g, N)
    s_xg = _f32(s_xg)
    sx_rows = None if sx_rows is None else _f32(sx_rows).reshape(-1)
    rs = None if row_sel is None else np.ascontiguousarray(row_sel, dtype=np.uint8)
    bias = _f32(bias)
    acc = np.empty((M, K // g, N), dtype=np.int32) if want_acc else None
    out = np.empty((M, N), dtype=np.float32)
    lib().orc_gemm_wgroup(_p(a, C.c_int8), _p(w, C.c_int8), C.c_long(M), C.c_long(N), C.c_long(K), C.c_long(g),
                          _p(s_wg, C.c_float), _p(s_xg, C.c_float), C.c_float(sx0), C.c_float(sx0 if sx1 is None else sx1),
                          _p(rs, C.c_uint8), _p(sx_rows, C.c_float), _p(bias, C.c_float), _p(acc, C.c_int32), _p(out, C.c_float))
    return (out, acc) if want_acc else out


def linear_fakequant_f32(x, s_x, w_dq, bias=None) -> np.ndarray:
    x = _f32(x)
    w_dq = _f32(w_dq)
    bias = _f32(bias)
    M, K = x.shape
    N = w_dq.shape[0]
    out = np.empty((M, N), dtype=np.float32)
    lib().orc_linear_fakequant_f32(_p(x, C.c_float), C.c_long(M), C.c_long(K),
                                   C.c_float(s_x), _p(w_dq, C.c_float), C.c_long(N),
                                   _p(bias, C.c_float), _p(out, C.c_float))
    return out


def rmsn(x, mean_dim, eps=1e-5, mode=0):
    """module_util.py:55-61 (weight-less RMS norm) with the device kernel's summation order."""
    x = _f32(x)
    rows, cols = x.shape
    y = np.empty_like(x)
    lib().orc_rmsn(_p(x, C.c_float), C.c_long(rows), C.c_long(cols), C.c_float(mean_dim), C.c_float(eps),
                   C.c_int(mode), _p(y, C.c_float))
    return y


def quant_dyn(x, bits=8, clip=1.0, skip_col0=False):
    """quant_utils.py:205-268: dynamic symmetric per-token -> (int8 levels, fp32 scale per row)."""
    x = _f32(x)
    rows, cols = x.shape
    scale = np.empty(rows, dtype=np.float32)
    q = np.empty((rows, cols), dtype=np.int8)
    lib().orc_quant_dyn(_p(x, C.c_float), C.c_long(rows), C.c_long(cols), C.c_int(bits), C.c_float(clip),
                        C.c_int(int(skip_col0)), _p(scale, C.c_float), _p(q, C.c_int8))
    return q, scale


def quant_group(x, groupsize, bits=8, clip=1.0, mode=0):
    """quant_utils.py:181-203: (levels int8 [rows, cols], scales fp32 [rows, cols / groupsize]); mode = dtype of x."""
    x = _f32(x)
    rows, cols = x.shape
    assert cols % groupsize == 0
    q = np.empty((rows, cols), dtype=np.int8)
    scale = np.empty((rows, cols // groupsize), dtype=np.float32)
    lib().orc_quant_group(_p(x, C.c_float), C.c_long(rows), C.c_long(cols), C.c_long(groupsize), C.c_int(bits),
                          C.c_float(clip), C.c_int(mode), _p(scale, C.c_float), _p(q, C.c_int8))
    return q, scale


def quant_group_asym(x, g, bits=8, clip=1.0, mode=0):
    """quant_utils.py:181-203, sym = False: (stored int8 levels q - 2^(bits-1), scale [rows, cols/g], zero, shift)."""
    x = _f32(x)
    rows, cols = x.shape
    G = cols // g
    scale, zero, shift = (np.empty((rows, G), dtype=np.float32) for _ in range(3))
    q = np.empty((rows, cols), dtype=np.int8)
    lib().orc_quant_group_asym(_p(x, C.c_float), C.c_long(rows), C.c_long(cols), C.c_long(g), C.c_int(bits), C.c_float(clip),
                               C.c_int(mode), _p(scale, C.c_float), _p(zero, C.c_float), _p(shift, C.c_float), _p(q, C.c_int8))
    return q, scale, zero, shift


def quant_dyn_asym(x, bits=8, clip=1.0):
    """quant_utils.py:239-268 (asymmetric branch): dynamic per-token -> (stored int8 levels q - 2^(bits-1),
    scale, zero, shift = scale * (2^(bits-1) - zero)) per row."""
    x = _f32(x)
    rows, cols = x.shape
    scale, zero, shift = (np.empty(rows, dtype=np.float32) for _ in range(3))
    q = np.empty((rows, cols), dtype=np.int8)
    lib().orc_quant_dyn_asym(_p(x, C.c_float), C.c_long(rows), C.c_long(cols), C.c_int(bits), C.c_float(clip),
                             _p(scale, C.c_float), _p(zero, C.c_float), _p(shift, C.c_float), _p(q, C.c_int8))
    return q, scale, zero, shift


def quant_tensor(x, bits=8, clip=1.0, asym=False, skip_col0=False, mode=0):
    """quant_utils.py:214-237: dynamic per-tensor -> (stored int8 levels, scale, zero, shift); mode = dtype of x
    (0 fp32, 1 fp16, 2 bf16: the reference evaluates this mode in x's dtype)."""
    x = _f32(x)
    rows, cols = x.shape
    params = np.empty(3, dtype=np.float32)
    q = np.empty((rows, cols), dtype=np.int8)
    lib().orc_quant_tensor(_p(x, C.c_float), C.c_long(rows), C.c_long(cols), C.c_int(bits), C.c_float(clip),
                           C.c_int(int(asym)), C.c_int(int(skip_col0)), C.c_int(mode), _p(params, C.c_float), _p(q, C.c_int8))
    return q, params[0], params[1], params[2]


def silu_mul(g, u, mode=0):
    g, u = _f32(g), _f32(u)
    out = np.empty_like(g)
    lib().orc_silu_mul(_p(g, C.c_float), _p(u, C.c_float), C.c_long(g.size), C.c_int(mode), _p(out, C.c_float))
    return out


def quick_gelu(x, mode=0):
    x = _f32(x)
    out = np.empty_like(x)
    lib().orc_quick_gelu(_p(x, C.c_float), C.c_long(x.size), C.c_int(mode), _p(out, C.c_float))
    return out


def pow_pos(d, norm):
    lib().orc_pow_pos.restype = C.c_float
    return np.float32(lib().orc_pow_pos(C.c_float(d), C.c_float(norm)))


def wquant_sym(w, bits=4, mse=False, norm=2.4, grid=100, maxshrink=0.8, want_levels=True):
    w = _f32(w)
    N, K = w.shape
    scale = np.empty(N, dtype=np.float32)
    levels = np.empty((N, K), dtype=np.int8) if want_levels else None
    lib().orc_wquant_sym(_p(w, C.c_float), C.c_long(N), C.c_long(K), C.c_int(bits),
                         C.c_int(int(mse)), C.c_float(norm), C.c_int(grid),
                         C.c_float(maxshrink), _p(scale, C.c_float), _p(levels, C.c_int8))
    return scale, levels


def wquant_asym(w, bits=4, mse=False, norm=2.4, grid=100, maxshrink=0.8, want_levels=True):
    """quant_utils.py:446-509 (sym = False): per-channel (scale, zero, stored levels q - 2^(bits-1))."""
    w = _f32(w)
    N, K = w.shape
    scale, zero = np.empty(N, dtype=np.float32), np.empty(N, dtype=np.float32)
    levels = np.empty((N, K), dtype=np.int8) if want_levels else None
    lib().orc_wquant_asym(_p(w, C.c_float), C.c_long(N), C.c_long(K), C.c_int(bits), C.c_int(int(mse)),
                          C.c_float(norm), C.c_int(grid), C.c_float(maxshrink), _p(scale, C.c_float),
                          _p(zero, C.c_float), _p(levels, C.c_int8))
    return scale, zero, levels


def gptq_block(W1, Hb, scale, bits):
    """Column loop of one GPTQ block (reference gptq_utils.py:249-286, symmetric per-channel
    quantizer without groups).  W1 [N, cols], Hb = Hinv[i1:i2, i1:i2], scale [N] -> (Q1, Err1)."""
    W1, Hb, scale = _f32(W1), _f32(Hb), _f32(scale).reshape(-1)
    N, cols = W1.shape
    Q1 = np.zeros((N, cols), dtype=np.float32)
    E1 = np.zeros((N, cols), dtype=np.float32)
    lib().orc_gptq_block(_p(W1, C.c_float), C.c_long(N), C.c_long(cols), C.c_long(cols),
                         _p(Hb, C.c_float), C.c_long(Hb.shape[1]), _p(scale, C.c_float), C.c_int(bits),
                         _p(Q1, C.c_float), C.c_long(cols), _p(E1, C.c_float), C.c_long(cols))
    return Q1, E1


def fp8_e4m3fn_encode(x):
    """float32 array -> OCP e4m3fn bytes (round to nearest even, saturating)."""
    x = _f32(x)
    out = np.empty(x.shape, dtype=np.uint8)
    f = lib().orc_fp8_e4m3fn_encode
    f.restype, f.argtypes = C.c_uint8, [C.c_float]
    flat_in, flat_out = x.reshape(-1), out.reshape(-1)
    for i in range(flat_in.size):
        flat_out[i] = f(float(flat_in[i]))
    return out


def fp8_e4m3fn_decode(b):
    b = np.ascontiguousarray(b, dtype=np.uint8)
    out = np.empty(b.shape, dtype=np.float32)
    f = lib().orc_fp8_e4m3fn_decode
    f.restype, f.argtypes = C.c_float, [C.c_uint8]
    flat_in, flat_out = b.reshape(-1), out.reshape(-1)
    for i in range(flat_in.size):
        flat_out[i] = f(int(flat_in[i]))
    return out


def kv_quant_fp8(x, scale):
    """x [T, H, D] float32 (values of the source dtype), scale [H] -> uint8 e4m3fn [T, H, D]."""
    x, scale = _f32(x), _f32(scale).reshape(-1)
    T, H, D = x.shape
    out = np.empty((T, H, D), dtype=np.uint8)
    lib().orc_kv_quant_fp8(_p(x, C.c_float), C.c_long(T), C.c_long(H), C.c_long(D), _p(scale, C.c_float),
                           _p(out, C.c_uint8))
    return out


def kv_dequant_fp8(q, scale, mode=0):
    """uint8 e4m3fn [T, H, D] -> float32 of the fp32 product rounded to the output dtype (``round_to`` mode)."""
    q, scale = np.ascontiguousarray(q, dtype=np.uint8), _f32(scale).reshape(-1)
    T, H, D = q.shape
    out = np.empty((T, H, D), dtype=np.float32)
    lib().orc_kv_dequant_fp8(_p(q, C.c_uint8), C.c_long(T), C.c_long(H), C.c_long(D), _p(scale, C.c_float),
                             _p(out, C.c_float))
    return round_to(out, mode)
