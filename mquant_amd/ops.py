"""Torch-tensor front-end of the C ABI.  PyTorch is plumbing here (device memory and
streams); every function below launches hand-written gfx950 kernels through
``include/mquant_hip.h`` and has no eager/CPU fallback.
"""
from __future__ import annotations

import math
import weakref
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import MQ_BF16, MQ_F16, MQ_F32, call

_DT = {torch.float16: MQ_F16, torch.bfloat16: MQ_BF16, torch.float32: MQ_F32}


def dtype_code(dt: torch.dtype) -> int:
    try:
        return _DT[dt]
    except KeyError:
        raise TypeError(f"unsupported activation dtype {dt}") from None


LD_TILED = 0   # include/mquant_hip.h MQ_LD_TILED


class TiledAct:
    """int8 activations in the TILED layout (``MQ_LD_TILED``): ``data`` is [ceil(M/16), K_pad/64, 64, 16],
    one 1 KiB piece = the MFMA operand fragment of a 16-row x 64-k tile in lane order.  The GEMM
    fetches a piece with one contiguous LDS-DMA (3x the L2 rate of gathering rows)."""
    __slots__ = ("data", "M", "K_pad", "generation")

    def __init__(self, data: torch.Tensor, M: int, K_pad: int):
        assert data.dtype == torch.int8 and data.is_contiguous() and K_pad % 64 == 0
        assert data.numel() >= ceil_to(M, 16) * K_pad
        self.data, self.M, self.K_pad = data, M, K_pad
        #: (workspace key, hand-out counter) when the image lives in engine.Workspace: a handle whose counter is no longer the
        #: key's latest has been overwritten by a later quantizer of the same width (checked under MQ_DEBUG_WORKSPACE=1)
        self.generation = None

    @staticmethod
    def empty(M: int, K_pad: int, device) -> "TiledAct":
        return TiledAct(torch.empty((ceil_to(max(M, 1), 16) // 16, K_pad // 64, 64, 16), dtype=torch.int8,
                                    device=device), M, K_pad)

    @property
    def device(self):
        return self.data.device

    @property
    def is_cuda(self):
        return self.data.is_cuda

    def data_ptr(self) -> int:
        return self.data.data_ptr()

    def to_rows(self) -> torch.Tensor:
        """Row-major [M, K_pad] copy (tests, debugging)."""
        mt, kt = ceil_to(max(self.M, 1), 16) // 16, self.K_pad // 64
        t = self.data.reshape(-1)[: mt * kt * 1024].reshape(mt, kt, 4, 16, 16)   # [mt][kt][c][r][16]
        return t.permute(0, 3, 1, 2, 4).reshape(mt * 16, self.K_pad)[: self.M].contiguous()

    @staticmethod
    def from_rows(a: torch.Tensor) -> "TiledAct":
        """Tile a row-major int8 [M, K_pad] matrix with torch ops (tests; the quantizer kernels write
        the layout directly)."""
        M, K_pad = a.shape
        mt, kt = ceil_to(max(M, 1), 16) // 16, K_pad // 64
        full = torch.zeros((mt * 16, K_pad), dtype=torch.int8, device=a.device)
        full[:M] = a
        t = full.reshape(mt, 16, kt, 4, 16).permute(0, 2, 3, 1, 4).contiguous()    # [mt][kt][c][r][16]
        return TiledAct(t.reshape(mt, kt, 64, 16), M, K_pad)


def _a_args(a):
    """(pointer, leading dimension, M, K_pad) of an int8 activation operand in either layout."""
    if isinstance(a, TiledAct):
        return a.data_ptr(), LD_TILED, a.M, a.K_pad
    assert a.dtype == torch.int8 and a.dim() == 2 and a.stride(1) == 1
    # a row stride of 0 (an expanded row) would alias the MQ_LD_TILED sentinel of the C ABI
    assert a.stride(0) >= a.shape[1], "row-major int8 activations need a row stride >= K_pad (no expanded operands)"
    return a.data_ptr(), a.stride(0), a.shape[0], a.shape[1]


def _out_act(out, tiled: bool, M: int, K_pad_default: int, device):
    """Destination of a quantizer: (object to return, pointer, K_pad, leading dimension)."""
    if out is None:
        out = TiledAct.empty(M, K_pad_default, device) if tiled else \
            torch.empty((M, K_pad_default), dtype=torch.int8, device=device)
    if isinstance(out, TiledAct):
        assert out.M == M
        return out, out.data_ptr(), out.K_pad, LD_TILED
    return out, out.data_ptr(), out.shape[1], out.stride(0)


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _need_cuda(*ts):
    dev = None
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise _lib.MQuantHipError(
                "the W4A8 path runs on the GPU only (got a CPU tensor); there is no CPU fallback")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise _lib.MQuantHipError(f"operands live on different devices ({dev} and {t.device})")


def _on_device(fn):
    """Run the wrapped op with the FIRST tensor argument's device current: the library launches on
    the current device and its current stream, and with the reference's device_map="auto" placement
    a wrapper's tensors may live on a GPU that is not the current one."""
    import functools

    @functools.wraps(fn)
    def wrapped(*args, **kwargs):
        for a in args:
            if isinstance(a, (torch.Tensor, TiledAct)) and a.is_cuda:
                if a.device.index != torch.cuda.current_device():
                    with torch.cuda.device(a.device):
                        return fn(*args, **kwargs)
                break
        return fn(*args, **kwargs)
    return wrapped


def ceil_to(v: int, m: int) -> int:
    return (v + m - 1) // m * m


def _rows(x: torch.Tensor) -> torch.Tensor:
    """View as [M, C] with a unit inner stride (no copy when already so)."""
    x2 = x.reshape(-1, x.shape[-1])
    if x2.stride(-1) != 1:
        x2 = x2.contiguous()
    return x2


# --------------------------------------------------------------------------- quantizer
@_on_device
def quantize_act_i8(x: torch.Tensor, scale0: float = 1.0, scale1: Optional[float] = None, *,
                    scale_vec0: Optional[torch.Tensor] = None,
                    scale_vec1: Optional[torch.Tensor] = None,
                    row_sel: Optional[torch.Tensor] = None, skip_col0: bool = False,
                    out=None, x0_out: Optional[torch.Tensor] = None, tiled: bool = False):
    """fp -> int8 levels, [M, K] -> [M, ceil128(K)] (pad columns are zero); ``tiled`` (or a
    ``TiledAct`` destination) selects the layout the GEMM streams fastest."""
    x2 = _rows(x)
    _need_cuda(x2, scale_vec0, scale_vec1, row_sel, out)
    M, K = x2.shape
    out, optr, K_pad, ldo = _out_act(out, tiled, M, ceil_to(K, 128), x.device)
    if skip_col0 and x0_out is None:
        x0_out = torch.empty((M,), dtype=torch.float32, device=x.device)
    call("mq_quantize_act_i8", x2.data_ptr(), dtype_code(x2.dtype), M, K, x2.stride(0),
         float(scale0), float(scale0 if scale1 is None else scale1),
         _ptr(scale_vec0), _ptr(scale_vec1), _ptr(row_sel), int(skip_col0), _ptr(x0_out),
         optr, K_pad, ldo, _stream())
    return out, x0_out


@_on_device
def rmsn_quantize_i8(x: torch.Tensor, mean_dim: float, eps: float, scale0: float,
                     scale1: Optional[float] = None, *, row_sel: Optional[torch.Tensor] = None,
                     out=None, want_y: bool = False, tiled: bool = False):
    """Weight-less RMS norm (module_util.RMSN) + static int8 quantizer in one pass.
    Returns (int8 [M, ceil128(K)] or TiledAct, normalised activations in x's dtype | None)."""
    x2 = _rows(x)
    _need_cuda(x2, row_sel, out)
    M, K = x2.shape
    out, optr, K_pad, ldo = _out_act(out, tiled, M, ceil_to(K, 128), x.device)
    y = torch.empty((M, K), dtype=x.dtype, device=x.device) if want_y else None
    call("mq_rmsn_quantize_i8", x2.data_ptr(), dtype_code(x2.dtype), M, K, x2.stride(0), float(mean_dim),
         float(eps), float(scale0), float(scale0 if scale1 is None else scale1), _ptr(row_sel),
         _ptr(y), K, optr, K_pad, ldo, _stream())
    return out, (y.reshape(x.shape) if want_y else None)


@_on_device
def fakequant_act(x: torch.Tensor, scale0: float = 1.0, scale1: Optional[float] = None, *,
                  scale_vec0: Optional[torch.Tensor] = None,
                  scale_vec1: Optional[torch.Tensor] = None,
                  row_sel: Optional[torch.Tensor] = None, skip_col0: bool = False) -> torch.Tensor:
    """Fused quantize->dequantize in x's dtype (same shape as x)."""
    x2 = _rows(x)
    _need_cuda(x2, scale_vec0, scale_vec1, row_sel)
    M, K = x2.shape
    out = torch.empty((M, K), dtype=x.dtype, device=x.device)
    call("mq_fakequant_act", x2.data_ptr(), dtype_code(x2.dtype), M, K, x2.stride(0),
         float(scale0), float(scale0 if scale1 is None else scale1),
         _ptr(scale_vec0), _ptr(scale_vec1), _ptr(row_sel), int(skip_col0),
         out.data_ptr(), out.stride(0), _stream())
    return out.reshape(x.shape)


# --------------------------------------------------------------------------- Hadamard
HAD_FP32, HAD_PREPARED, HAD_FAST = 1, 2, 4      # include/mquant_hip.h MQ_HAD_*


@_on_device
def hadamard_prepare(words: torch.Tensor, K: int) -> torch.Tensor:
    """Sign words [K, ceil(K/32)] (int32) -> prepared descriptor (``mq_hadamard_prepare``): the words plus
    the 64-lane masks of the MFMA sign operand.  Returned as an int64 tensor; every Hadamard op below
    recognises it by that dtype and sets MQ_HAD_PREPARED.  Results are identical, the K x K stage is faster."""
    _need_cuda(words)
    assert words.dtype == torch.int32 and words.is_contiguous()
    nbytes = _lib.load().mq_hadamard_prepared_bytes(K)
    out = torch.zeros((nbytes // 8,), dtype=torch.int64, device=words.device)
    call("mq_hadamard_prepare", words.data_ptr(), K, out.data_ptr(), _stream())
    return out


def _had_flags(had_bits: Optional[torch.Tensor], fp32_had: bool, fast: bool = False) -> int:
    """Flag word of the Hadamard entry points.  ``fast`` = MQ_HAD_FAST: THIS call may run the K x K stage on the
    half-precision matrix core (same exact +-1 products, another fp32 accumulation order: NOT bit-identical to the
    reference's CPU run).  Never set by default; there is no process-wide switch."""
    return ((HAD_FP32 if fp32_had else 0) | (HAD_PREPARED if had_bits is not None and had_bits.dtype == torch.int64 else 0)
            | (HAD_FAST if fast else 0))


@_on_device
def hadamard(x: torch.Tensor, n: int, K: int, had_bits: Optional[torch.Tensor],
             fp32_had: bool = False, fast: bool = False) -> torch.Tensor:
    """Rotated activations in x's dtype, last dim zero-padded from x.shape[-1] to n."""
    x2 = _rows(x)
    _need_cuda(x2, had_bits)
    M, n_in = x2.shape
    out = torch.empty((M, n), dtype=x.dtype, device=x.device)
    call("mq_hadamard", x2.data_ptr(), dtype_code(x2.dtype), M, n_in, x2.stride(0), n, K,
         _ptr(had_bits), _had_flags(had_bits, fp32_had, fast), out.data_ptr(), out.stride(0), _stream())
    return out.reshape(*x.shape[:-1], n)


@_on_device
def hadamard_quant_i8(x: torch.Tensor, n: int, K: int, had_bits: Optional[torch.Tensor],
                      scale0: float, scale1: Optional[float] = None, *, fp32_had: bool = False,
                      row_sel: Optional[torch.Tensor] = None, skip_col0: bool = False,
                      out=None, x0_out: Optional[torch.Tensor] = None, tiled: bool = False, fast: bool = False):
    x2 = _rows(x)
    _need_cuda(x2, had_bits, row_sel, out)
    M, n_in = x2.shape
    out, optr, K_pad, ldo = _out_act(out, tiled, M, ceil_to(n, 128), x.device)
    if skip_col0 and x0_out is None:
        x0_out = torch.empty((M,), dtype=torch.float32, device=x.device)
    call("mq_hadamard_quant_i8", x2.data_ptr(), dtype_code(x2.dtype), M, n_in, x2.stride(0), n, K,
         _ptr(had_bits), _had_flags(had_bits, fp32_had, fast), float(scale0), float(scale0 if scale1 is None else scale1),
         _ptr(row_sel), int(skip_col0), _ptr(x0_out), optr, K_pad, ldo, _stream())
    return out, x0_out


@_on_device
def rope_inplace(x: torch.Tensor, heads: int, head_dim: int, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    """Rotate-half RoPE in place on the first heads*head_dim columns of x [T, >= heads*head_dim]
    (x may be a column slice of a wider tensor).  cos / sin: [T, head_dim], x's dtype."""
    _need_cuda(x, cos, sin)
    assert x.dim() == 2 and x.stride(1) == 1 and cos.dtype == x.dtype == sin.dtype
    assert cos.is_contiguous() and sin.is_contiguous() and cos.shape == (x.shape[0], head_dim) == sin.shape
    call("mq_rope_inplace", x.data_ptr(), dtype_code(x.dtype), x.shape[0], heads, head_dim, x.stride(0),
         cos.data_ptr(), sin.data_ptr(), _stream())
    return x


ACT_SILU_MUL, ACT_QUICK_GELU = 1, 2


@_on_device
def act_hadamard_quant_i8(x: torch.Tensor, x2: Optional[torch.Tensor], act: int, n: int, K: int,
                          had_bits: Optional[torch.Tensor], scale0: float, scale1: Optional[float] = None, *,
                          fp32_had: bool = False, row_sel: Optional[torch.Tensor] = None,
                          skip_col0: bool = False, out=None,
                          x0_out: Optional[torch.Tensor] = None, tiled: bool = False, fast: bool = False):
    """silu(x) * x2 (ACT_SILU_MUL) or quick_gelu(x) (ACT_QUICK_GELU) -> [pad] -> Hadamard -> int8,
    one launch.  x and x2 may be column slices of one tensor (same row stride)."""
    a = _rows(x)
    b = _rows(x2) if x2 is not None else None
    _need_cuda(a, b, had_bits, row_sel, out)
    if b is not None:
        assert b.shape == a.shape and b.stride(0) == a.stride(0) and b.dtype == a.dtype
    M, n_in = a.shape
    out, optr, K_pad, ldo = _out_act(out, tiled, M, ceil_to(n, 128), x.device)
    if skip_col0 and x0_out is None:
        x0_out = torch.empty((M,), dtype=torch.float32, device=x.device)
    call("mq_act_hadamard_quant_i8", a.data_ptr(), _ptr(b), int(act), dtype_code(a.dtype), M, n_in, a.stride(0),
         n, K, _ptr(had_bits), _had_flags(had_bits, fp32_had, fast), float(scale0), float(scale0 if scale1 is None else scale1),
         _ptr(row_sel), int(skip_col0), _ptr(x0_out), optr, K_pad, ldo, _stream())
    return out, x0_out


# --------------------------------------------------------------------------- weights
@_on_device
def pack_i4(q: torch.Tensor) -> torch.Tensor:
    _need_cuda(q)
    q = q.to(torch.int8).contiguous()
    rows, cols = q.reshape(-1, q.shape[-1]).shape
    out = torch.empty((rows, cols // 2), dtype=torch.uint8, device=q.device)
    call("mq_pack_i4", q.data_ptr(), rows, cols, out.data_ptr(), _stream())
    return out.reshape(*q.shape[:-1], cols // 2)


@_on_device
def unpack_i4(p: torch.Tensor) -> torch.Tensor:
    _need_cuda(p)
    p = p.contiguous()
    rows, half = p.reshape(-1, p.shape[-1]).shape
    out = torch.empty((rows, half * 2), dtype=torch.int8, device=p.device)
    call("mq_unpack_i4", p.data_ptr(), rows, half * 2, out.data_ptr(), _stream())
    return out.reshape(*p.shape[:-1], half * 2)


@_on_device
def weight_levels(w: torch.Tensor, scale: torch.Tensor, bits: int) -> torch.Tensor:
    """Integer levels of a fake-quantized weight: clamp(rint(w / scale[n]), -2^(b-1), 2^(b-1)-1)."""
    _need_cuda(w, scale)
    w2 = w.reshape(w.shape[0], -1)
    if w2.stride(-1) != 1:
        w2 = w2.contiguous()
    N, K = w2.shape
    scale = scale.reshape(-1).to(torch.float32).contiguous()
    assert scale.numel() == N
    q = torch.empty((N, K), dtype=torch.int8, device=w.device)
    call("mq_weight_levels", w2.data_ptr(), dtype_code(w2.dtype), N, K, w2.stride(0),
         scale.data_ptr(), -(1 << (bits - 1)), (1 << (bits - 1)) - 1, q.data_ptr(), _stream())
    return q


@_on_device
def wquant_sym(w: torch.Tensor, bits: int = 4, mse: bool = False, norm: float = 2.4, grid: int = 100,
               maxshrink: float = 0.8, want_levels: bool = True, want_packed: bool = False,
               want_wq: bool = False):
    """Symmetric per-output-channel weight quantizer in one launch (``mq_wquant_sym``).
    Returns (scale fp32 [N], levels int8 [N, K] | None, packed uint8 [N, K/2] | None, W~ | None)."""
    _need_cuda(w)
    w2 = w.flatten(1) if w.dim() > 1 else w.reshape(1, -1)
    if w2.stride(-1) != 1:
        w2 = w2.contiguous()
    N, K = w2.shape
    dev = w.device
    scale = torch.empty((N,), dtype=torch.float32, device=dev)
    levels = torch.empty((N, K), dtype=torch.int8, device=dev) if want_levels else None
    packed = torch.empty((N, K // 2), dtype=torch.uint8, device=dev) if want_packed else None
    wq = torch.empty((N, K), dtype=w.dtype, device=dev) if want_wq else None
    call("mq_wquant_sym", w2.data_ptr(), dtype_code(w2.dtype), N, K, w2.stride(0), bits, int(mse), float(norm),
         int(grid), float(maxshrink), scale.data_ptr(), levels.data_ptr() if want_levels else None,
         packed.data_ptr() if want_packed else None, wq.data_ptr() if want_wq else None, K, _stream())
    return scale, levels, packed, (wq.reshape(w.shape) if want_wq else None)


@_on_device
def gptq_block(W: torch.Tensor, i1: int, i2: int, Hinv: torch.Tensor, scale: torch.Tensor, bits: int,
               Q: torch.Tensor, Err: torch.Tensor) -> None:
    """Column loop of one GPTQ block (``mq_gptq_block``): reads W[:, i1:i2], writes Q[:, i1:i2] and
    Err[:, :i2-i1].  W, Q: fp32 [N, columns] row-major; Hinv: fp32 [columns, columns] upper factor."""
    _need_cuda(W, Hinv, scale, Q, Err)
    assert W.dtype == Hinv.dtype == Q.dtype == Err.dtype == scale.dtype == torch.float32
    assert W.stride(1) == 1 and Q.stride(1) == 1 and Err.stride(1) == 1 and Hinv.stride(1) == 1 and scale.is_contiguous()
    N = W.shape[0]
    fsz = 4
    call("mq_gptq_block", W.data_ptr() + i1 * fsz, N, i2 - i1, W.stride(0),
         Hinv.data_ptr() + (i1 * Hinv.stride(0) + i1) * fsz, Hinv.stride(0), scale.data_ptr(), bits,
         Q.data_ptr() + i1 * fsz, Q.stride(0), Err.data_ptr(), Err.stride(0), _stream())


@_on_device
def rotate_f64_(x: torch.Tensor, signs, K: int, had_words) -> torch.Tensor:
    """In place ``x <- cast((H_K (x) H_{n/K}) (signs . x) / sqrt(n))`` over the last dim, evaluated in
    fp64 (``mq_rotate_f64``).  x: [..., n] with contiguous rows; signs: float64 [n] or None;
    had_words: the plain int32 sign words of hadK (None when K == 1)."""
    _need_cuda(x)
    n = x.shape[-1]
    x2 = x.view(-1, n)
    assert x2.stride(1) == 1
    code = 3 if x.dtype == torch.float64 else dtype_code(x.dtype)
    if signs is not None:
        _need_cuda(x, signs)
        assert signs.dtype == torch.float64 and signs.numel() == n and signs.is_contiguous()
    call("mq_rotate_f64", x2.data_ptr(), code, x2.shape[0], n, x2.stride(0) if x2.shape[0] > 1 else n,
         0 if signs is None else signs.data_ptr(), K, 0 if had_words is None else had_words.data_ptr(), _stream())
    return x


FP8_E4M3_MAX = 448.0


@_on_device
def kv_quant_fp8(kv: torch.Tensor, scale: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
    """K or V [T, kv_heads, head_dim] (token stride free: a slice of the fused qkv output works in
    place) -> e4m3fn bytes [T, kv_heads, head_dim] (torch.float8_e4m3fn) with one static scale per head
    (``mq_kv_quant_fp8``).  scale: float32 [kv_heads] = calibrated absmax / 448."""
    _need_cuda(kv, scale, out)
    T, H, D = kv.shape
    assert kv.stride(2) == 1 and kv.stride(1) == D and scale.dtype == torch.float32 and scale.numel() == H
    if out is None:
        out = torch.empty((T, H, D), dtype=torch.float8_e4m3fn, device=kv.device)
    assert out.dtype == torch.float8_e4m3fn and out.stride(2) == 1 and out.stride(1) == D
    call("mq_kv_quant_fp8", kv.data_ptr(), dtype_code(kv.dtype), T, H, D, kv.stride(0) if T > 1 else H * D,
         scale.data_ptr(), out.data_ptr(), out.stride(0) if T > 1 else H * D, _stream())
    return out


@_on_device
def kv_quant_fp8_readback(kv: torch.Tensor, scale: torch.Tensor, out: torch.Tensor = None,
                          readback: torch.Tensor = None):
    """Quantize-on-write with the read-back fused in (``mq_kv_quant_fp8_readback``): returns (cache bytes, the
    values a later ``kv_dequant_fp8`` of those bytes gives, in kv's dtype).  kv may hold K and V side by side
    ([T, 2 * kv_heads, head_dim], the K|V columns of the fused q|k|v output) with scale [2 * kv_heads]."""
    _need_cuda(kv, scale, out, readback)
    T, H, D = kv.shape
    assert kv.stride(2) == 1 and kv.stride(1) == D and scale.dtype == torch.float32 and scale.numel() == H
    if out is None:
        out = torch.empty((T, H, D), dtype=torch.float8_e4m3fn, device=kv.device)
    if readback is None:
        readback = torch.empty((T, H, D), dtype=kv.dtype, device=kv.device)
    assert out.dtype == torch.float8_e4m3fn and out.stride(2) == 1 and out.stride(1) == D
    assert readback.dtype == kv.dtype and readback.stride(2) == 1 and readback.stride(1) == D and readback.shape == kv.shape
    row = H * D
    call("mq_kv_quant_fp8_readback", kv.data_ptr(), dtype_code(kv.dtype), T, H, D, kv.stride(0) if T > 1 else row,
         scale.data_ptr(), out.data_ptr(), out.stride(0) if T > 1 else row, readback.data_ptr(),
         readback.stride(0) if T > 1 else row, _stream())
    return out, readback


@_on_device
def kv_dequant_fp8(q: torch.Tensor, scale: torch.Tensor, dtype: torch.dtype = torch.float16,
                   out: torch.Tensor = None) -> torch.Tensor:
    """e4m3fn cache [T, kv_heads, head_dim] -> ``dtype`` in front of SDPA (``mq_kv_dequant_fp8``)."""
    _need_cuda(q, scale, out)
    T, H, D = q.shape
    assert q.dtype == torch.float8_e4m3fn and q.stride(2) == 1 and q.stride(1) == D and scale.numel() == H
    if out is None:
        out = torch.empty((T, H, D), dtype=dtype, device=q.device)
    assert out.stride(2) == 1 and out.stride(1) == D
    call("mq_kv_dequant_fp8", q.data_ptr(), T, H, D, q.stride(0) if T > 1 else H * D, scale.data_ptr(),
         out.data_ptr(), dtype_code(out.dtype), out.stride(0) if T > 1 else H * D, _stream())
    return out


_KV_SCALE_OK = {}      # id(tensor) -> (weak reference to it, version counter, address) the check was made at


def _check_kv_scale(kv_scale: torch.Tensor, kv_heads: int) -> None:
    """kv_heads K scales then kv_heads V scales, fp32, positive and finite (the attention kernels apply them AFTER the running
    maximum over raw scores: a zero / negative scale silently gives wrong probabilities).  The value check reads the tensor back
    once per tensor OBJECT and version counter (the entry holds a weak reference: an address -- or an id -- recycled after the
    tensor died belongs to a new object and is checked again; advisor finding r4) and never during stream capture."""
    assert kv_scale.dtype == torch.float32 and kv_scale.is_contiguous() and kv_scale.numel() == 2 * kv_heads, \
        f"kv_scale must hold 2 * kv_heads = {2 * kv_heads} fp32 values (got {tuple(kv_scale.shape)})"
    version = kv_scale._version if not kv_scale.is_inference() else -1
    seen = _KV_SCALE_OK.get(id(kv_scale))
    if (seen is not None and seen[0]() is kv_scale and seen[1:] == (version, kv_scale.data_ptr())) or torch.cuda.is_current_stream_capturing():
        return
    ok = bool((torch.isfinite(kv_scale) & (kv_scale > 0)).all().item())
    assert ok, "kv_scale entries must be positive and finite"
    if len(_KV_SCALE_OK) > 256:
        for k in [k for k, v in _KV_SCALE_OK.items() if v[0]() is None]:
            del _KV_SCALE_OK[k]
        if len(_KV_SCALE_OK) > 256:
            _KV_SCALE_OK.clear()
    _KV_SCALE_OK[id(kv_scale)] = (weakref.ref(kv_scale), version, kv_scale.data_ptr())


@_on_device
def attn_prefill_fp8kv(q: torch.Tensor, kv_cache: torch.Tensor, kv_scale: torch.Tensor, causal: bool = True,
                       softmax_scale: float = None, out: torch.Tensor = None) -> torch.Tensor:
    """Prefill attention that reads the e4m3 cache directly (``mq_attn_prefill_fp8kv``): q [T, heads, 128] fp16 /
    bf16 (may be a column slice of the fused q|k|v output), kv_cache [T, 2 * kv_heads, 128] float8_e4m3fn (K heads
    then V heads, what ``kv_quant_fp8`` writes for the K|V columns), kv_scale [2 * kv_heads] -> [T, heads * 128]."""
    _need_cuda(q, kv_cache, kv_scale, out)
    T, H, D = q.shape
    T2, H2, D2 = kv_cache.shape
    assert T2 == T and D2 == D and H2 % 2 == 0 and kv_cache.dtype == torch.float8_e4m3fn
    assert q.stride(2) == 1 and q.stride(1) == D and kv_cache.stride(2) == 1 and kv_cache.stride(1) == D
    _check_kv_scale(kv_scale, H2 // 2)
    if out is None:
        out = torch.empty((T, H * D), dtype=q.dtype, device=q.device)
    assert out.dtype == q.dtype and out.shape == (T, H * D) and out.stride(1) == 1
    if softmax_scale is None:
        softmax_scale = D ** -0.5
    call("mq_attn_prefill_fp8kv", q.data_ptr(), dtype_code(q.dtype), T, H, H2 // 2, D, q.stride(0) if T > 1 else H * D,
         kv_cache.data_ptr(), kv_cache.stride(0) if T > 1 else H2 * D, kv_scale.data_ptr(), float(softmax_scale),
         1 if causal else 0, out.data_ptr(), out.stride(0) if T > 1 else H * D, _stream())
    return out


@_on_device
def attn_prefill(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, causal: bool = True, softmax_scale: float = None,
                 out: torch.Tensor = None) -> torch.Tensor:
    """Prefill attention over unquantised K / V (``mq_attn_prefill``): q [T, heads, D], k and v [T, kv_heads, D] of the
    same fp16 / bf16 dtype and the same token stride (column slices of the fused q|k|v output work in place), D = 128
    (Qwen2-VL decoder) or 80 (its vision tower) -> [T, heads * D], the layout o_proj / proj consumes."""
    _need_cuda(q, k, v, out)
    T, H, D = q.shape
    assert k.shape == v.shape and k.shape[0] == T and k.shape[2] == D and k.dtype == q.dtype and v.dtype == q.dtype
    HKV = k.shape[1]
    assert q.stride(2) == 1 and q.stride(1) == D
    for t in (k, v):
        assert t.stride(2) == 1 and t.stride(1) == D
    assert T <= 1 or k.stride(0) == v.stride(0), "k and v must share their token stride"
    if out is None:
        out = torch.empty((T, H * D), dtype=q.dtype, device=q.device)
    assert out.dtype == q.dtype and out.shape == (T, H * D) and out.stride(1) == 1
    if softmax_scale is None:
        softmax_scale = D ** -0.5
    call("mq_attn_prefill", q.data_ptr(), dtype_code(q.dtype), T, H, HKV, D, q.stride(0) if T > 1 else H * D,
         k.data_ptr(), v.data_ptr(), k.stride(0) if T > 1 else HKV * D, float(softmax_scale), 1 if causal else 0,
         out.data_ptr(), out.stride(0) if T > 1 else H * D, _stream())
    return out


@_on_device
def gemv_f16(x: torch.Tensor, w: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
    """``x @ w.t()`` for a few rows x ([M <= 8, K]) against a large 16-bit matrix w ([N, K]) of the same fp16 / bf16 dtype
    (``mq_gemv_f16``): the unquantized lm_head on the last position(s) -- w is streamed once at the chip's HBM rate, fp32
    products and sums, one rounding.  Returns [M, N]."""
    _need_cuda(x, w, out)
    M, K = x.shape
    N = w.shape[0]
    assert w.shape[1] == K and w.dtype == x.dtype and x.stride(1) == 1 and w.stride(1) == 1
    if out is None:
        out = torch.empty((M, N), dtype=x.dtype, device=x.device)
    assert out.dtype == x.dtype and out.shape == (M, N) and out.stride(1) == 1
    call("mq_gemv_f16", x.data_ptr(), dtype_code(x.dtype), M, K, x.stride(0) if M > 1 else K, w.data_ptr(), N,
         w.stride(0) if N > 1 else K, out.data_ptr(), out.stride(0) if M > 1 else N, _stream())
    return out


@_on_device
def attn_prefill_quant_i8(q: torch.Tensor, scale0: float, scale1: Optional[float] = None, *, k: torch.Tensor = None,
                          v: torch.Tensor = None, kv_cache: torch.Tensor = None, kv_scale: torch.Tensor = None,
                          causal: bool = True, softmax_scale: float = None, row_sel: Optional[torch.Tensor] = None,
                          out=None, tiled: bool = False):
    """``attn_prefill`` (k, v) or ``attn_prefill_fp8kv`` (kv_cache, kv_scale) with the next Linear's static int8 quantizer
    fused into the store (``mq_attn_prefill_quant_i8``): returns the int8 activations ``quantize_act_i8`` would produce
    from the 16-bit attention output ([T, heads * D] row-major, or a ``TiledAct``)."""
    _need_cuda(q, k, v, kv_cache, kv_scale, row_sel, out)
    T, H, D = q.shape
    assert q.stride(2) == 1 and q.stride(1) == D
    out, optr, K_pad, ldo = _out_act(out, tiled, T, H * D, q.device)
    if softmax_scale is None:
        softmax_scale = D ** -0.5
    if kv_cache is not None:
        assert kv_cache.dtype == torch.float8_e4m3fn and kv_cache.shape[0] == T and kv_cache.shape[2] == D
        assert kv_cache.stride(2) == 1 and kv_cache.stride(1) == D
        _check_kv_scale(kv_scale, kv_cache.shape[1] // 2)
        hkv, kp, vp, ldkv = kv_cache.shape[1] // 2, None, None, 0
        cp, ldc, sp = kv_cache.data_ptr(), kv_cache.stride(0) if T > 1 else kv_cache.shape[1] * D, kv_scale.data_ptr()
    else:
        assert k.shape == v.shape and k.shape[0] == T and k.shape[2] == D and k.dtype == q.dtype and v.dtype == q.dtype
        for t in (k, v):
            assert t.stride(2) == 1 and t.stride(1) == D
        assert T <= 1 or k.stride(0) == v.stride(0)
        hkv, kp, vp, ldkv = k.shape[1], k.data_ptr(), v.data_ptr(), k.stride(0) if T > 1 else k.shape[1] * D
        cp, ldc, sp = None, 0, None
    call("mq_attn_prefill_quant_i8", q.data_ptr(), dtype_code(q.dtype), T, H, hkv, D, q.stride(0) if T > 1 else H * D,
         kp, vp, ldkv, cp, ldc, sp, float(softmax_scale), 1 if causal else 0, float(scale0),
         float(scale0 if scale1 is None else scale1), _ptr(row_sel), optr, K_pad, ldo, _stream())
    return out


def kv_scale_from_absmax(kv: torch.Tensor) -> torch.Tensor:
    """Static per-head scale from calibration activations [T, kv_heads, head_dim]: absmax / 448."""
    return (kv.float().abs().amax(dim=(0, 2)).clamp_min(1e-8) / FP8_E4M3_MAX).contiguous()


@_on_device
def prepack(q: torch.Tensor, bits: int, zero_col0: bool = False) -> torch.Tensor:
    """int levels [N, K] -> the pre-tiled image streamed by gemm_w4a8."""
    _need_cuda(q)
    q = q.to(torch.int8).contiguous()
    N, K = q.shape
    nbytes = _lib.load().mq_prepacked_bytes(N, K, bits)
    out = torch.empty((nbytes,), dtype=torch.uint8, device=q.device)
    call("mq_prepack_w4" if bits == 4 else "mq_prepack_w8", q.data_ptr(), N, K, int(zero_col0),
         out.data_ptr(), _stream())
    return out


# --------------------------------------------------------------------------- GEMM
_SPLITK_WS = {}
_SPLITK_PINNED = []      # outgrown workspaces that a captured hipGraph may still replay into


def splitk_workspace(device, nbytes: int = 64 << 20) -> torch.Tensor:
    """Per-device scratch for split-K partial sums: ONE grow-only buffer per device, reused by every call in the
    device's current stream order.  A buffer handed out during stream capture is kept alive when a larger one replaces it."""
    key = (device.index or 0)
    ent = _SPLITK_WS.get(key)
    capturing = torch.cuda.is_current_stream_capturing()
    if ent is None or ent[0].numel() < nbytes:
        if ent is not None and ent[1]:
            _SPLITK_PINNED.append(ent[0])
        ent = [torch.empty((nbytes,), dtype=torch.uint8, device=device), False]
        _SPLITK_WS[key] = ent
    ent[1] = ent[1] or capturing
    return ent[0]


def gemm_debug_force(tile: int = -1, splits: int = 0) -> None:
    """Tuning hook: force the tile shape (ids: csrc/gemm_w4a8.hip dispatch_tile) / split-K."""
    call("mq_gemm_debug_force", tile, splits)


@_on_device
def gemm_w4a8(a: torch.Tensor, w_img: torch.Tensor, w_bits: int, N: int, s_x0: float,
              s_w: torch.Tensor, *, s_x1: Optional[float] = None,
              row_sel: Optional[torch.Tensor] = None, bias: Optional[torch.Tensor] = None,
              x0: Optional[torch.Tensor] = None, w0: Optional[torch.Tensor] = None,
              out_dtype: torch.dtype = torch.float16, M: Optional[int] = None,
              out: Optional[torch.Tensor] = None, use_workspace: bool = True) -> torch.Tensor:
    _need_cuda(a, w_img, s_w, row_sel, bias, x0, w0, out)
    aptr, lda, M_a, K_pad = _a_args(a)
    M = M_a if M is None else M
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype, device=a.device)
    ws = splitk_workspace(a.device) if use_workspace else None
    call("mq_gemm_w4a8_ws", aptr, lda, w_img.data_ptr(), w_bits, M, N, K_pad,
         float(s_x0), float(s_x0 if s_x1 is None else s_x1), _ptr(row_sel), s_w.data_ptr(),
         _ptr(bias), _ptr(x0), _ptr(w0), out.data_ptr(), dtype_code(out.dtype), out.stride(0),
         _ptr(ws), 0 if ws is None else ws.numel(), _stream())
    return out


@_on_device
def gemm_w4a8_act(a, w_img: torch.Tensor, w_bits: int, N: int, s_x0: float, s_w: torch.Tensor, act: int, *,
                  s_x1: Optional[float] = None, row_sel: Optional[torch.Tensor] = None,
                  s_x_rows: Optional[torch.Tensor] = None, bias: Optional[torch.Tensor] = None,
                  out_dtype: torch.dtype = torch.float16, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The Linear with its CONSUMER's activation folded into the store (``mq_gemm_w4a8_act_ws``): ``ACT_SILU_MUL`` -- the image
    holds gate (channels 0 .. N/2-1) then up, the result is silu(gate) * up, [M, N/2]; ``ACT_QUICK_GELU`` -- [M, N].  Bit-identical
    to ``gemm_w4a8`` followed by the torch ops on the rounded output.  Tiled activations only."""
    _need_cuda(a, w_img, s_w, row_sel, s_x_rows, bias, out)
    aptr, lda, M, K_pad = _a_args(a)
    n_out = N // 2 if act == ACT_SILU_MUL else N
    if out is None:
        out = torch.empty((M, n_out), dtype=out_dtype, device=w_img.device)
    assert out.shape == (M, n_out)
    call("mq_gemm_w4a8_act_ws", aptr, lda, w_img.data_ptr(), w_bits, M, N, K_pad, float(s_x0),
         float(s_x0 if s_x1 is None else s_x1), _ptr(row_sel), _ptr(s_x_rows), s_w.data_ptr(), _ptr(bias), int(act),
         out.data_ptr(), dtype_code(out.dtype), out.stride(0), _stream())
    return out


@_on_device
def quantize_act_dyn_i8(x: torch.Tensor, bits: int = 8, clip_ratio: float = 1.0, *, skip_col0: bool = False,
                        out=None, tiled: bool = False):
    """Dynamic symmetric per-token quantizer (the reference's default activation mode).
    Returns (int8 [M, ceil128(K)], per-row scales fp32 [M], column 0 as fp32 [M] | None)."""
    x2 = _rows(x)
    _need_cuda(x2, out)
    M, K = x2.shape
    out, optr, K_pad, ldo = _out_act(out, tiled, M, ceil_to(K, 128), x.device)
    scale = torch.empty((M,), dtype=torch.float32, device=x.device)
    x0 = torch.empty((M,), dtype=torch.float32, device=x.device) if skip_col0 else None
    call("mq_quantize_act_dyn_i8", x2.data_ptr(), dtype_code(x2.dtype), M, K, x2.stride(0), int(bits),
         float(clip_ratio), int(skip_col0), _ptr(x0), scale.data_ptr(), optr, K_pad, ldo,
         _stream())
    return out, scale, x0


@_on_device
def quantize_act_group_i8(x: torch.Tensor, groupsize: int, bits: int = 8, clip_ratio: float = 1.0, *, out=None,
                          tiled: bool = False):
    """Dynamic symmetric GROUP-WISE quantizer (``--a_groupsize``; reference quant_utils.py:181-203), every
    intermediate in x's dtype like the reference.  Returns (int8 [M, ceil128(K)], scales fp32 [M, K / groupsize])."""
    x2 = _rows(x)
    _need_cuda(x2, out)
    M, K = x2.shape
    out, optr, K_pad, ldo = _out_act(out, tiled, M, ceil_to(K, 128), x.device)
    scales = torch.empty((M, K // groupsize), dtype=torch.float32, device=x.device)
    call("mq_quantize_act_group_i8", x2.data_ptr(), dtype_code(x2.dtype), M, K, x2.stride(0), int(groupsize), int(bits),
         float(clip_ratio), scales.data_ptr(), optr, K_pad, ldo, _stream())
    return out, scales


@_on_device
def gemm_w4a8_groupscale(a, w_img: torch.Tensor, w_bits: int, N: int, s_groups: torch.Tensor, group_k: int,
                         s_w: torch.Tensor, *, bias: Optional[torch.Tensor] = None,
                         out_dtype: torch.dtype = torch.float16, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y = (sum_g float(acc_g) * s_groups[m][g]) * s_w[n] + bias[n] (``mq_gemm_w4a8_groupscale``): exact int32 sums
    inside a group of ``group_k`` consecutive k, fp32 across groups in ascending order."""
    _need_cuda(a, w_img, s_groups, s_w, bias, out)
    aptr, lda, M, K_pad = _a_args(a)
    assert s_groups.dtype == torch.float32 and s_groups.is_contiguous() and s_groups.shape[0] == M
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype, device=w_img.device)
    call("mq_gemm_w4a8_groupscale", aptr, lda, w_img.data_ptr(), w_bits, M, N, K_pad, s_groups.data_ptr(),
         s_groups.shape[1], int(group_k), s_w.data_ptr(), _ptr(bias), out.data_ptr(), dtype_code(out.dtype),
         out.stride(0), _stream())
    return out


@_on_device
def quantize_act_group_asym_i8(x: torch.Tensor, groupsize: int, bits: int = 8, clip_ratio: float = 1.0, *, out=None,
                               tiled: bool = False):
    """Dynamic ASYMMETRIC group-wise quantizer (``--a_groupsize`` + ``--a_asym``; reference quant_utils.py:181-203, sym = False),
    every intermediate in x's dtype like the reference.  Returns (stored int8 levels q - 2^(bits-1) [M, ceil128(K)], scales,
    zero points, shift = scale * (2^(bits-1) - zero): fp32 [M, K / groupsize] each)."""
    x2 = _rows(x)
    _need_cuda(x2, out)
    M, K = x2.shape
    out, optr, K_pad, ldo = _out_act(out, tiled, M, ceil_to(K, 128), x.device)
    scales, zero, shift = (torch.empty((M, K // groupsize), dtype=torch.float32, device=x.device) for _ in range(3))
    call("mq_quantize_act_group_asym_i8", x2.data_ptr(), dtype_code(x2.dtype), M, K, x2.stride(0), int(groupsize), int(bits),
         float(clip_ratio), scales.data_ptr(), zero.data_ptr(), shift.data_ptr(), optr, K_pad, ldo, _stream())
    return out, scales, zero, shift


@_on_device
def gemm_w4a8_rope(a, w_img: torch.Tensor, w_bits: int, N: int, s_x0: float, s_w: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor,
                   rope_cols: int, *, s_x1: Optional[float] = None, row_sel: Optional[torch.Tensor] = None,
                   bias: Optional[torch.Tensor] = None, out_dtype: torch.dtype = torch.float16,
                   out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """W4A8 Linear whose first ``rope_cols`` output columns (heads of 128) leave the GEMM already rotated
    (``mq_gemm_w4a8_rope_ws``): bit-identical to ``gemm_w4a8`` + ``rope_inplace``.  cos / sin: [M, 128] in the output dtype."""
    _need_cuda(a, w_img, s_w, row_sel, bias, cos, sin, out)
    aptr, lda, M, K_pad = _a_args(a)
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype, device=s_w.device)
    assert cos.dtype == out.dtype and sin.dtype == out.dtype and cos.is_contiguous() and sin.is_contiguous()
    assert tuple(cos.shape) == tuple(sin.shape) and cos.shape[-1] == 128 and cos.numel() >= M * 128
    call("mq_gemm_w4a8_rope_ws", aptr, lda, w_img.data_ptr(), w_bits, M, N, K_pad, float(s_x0), float(s_x0 if s_x1 is None else s_x1),
         _ptr(row_sel), s_w.data_ptr(), _ptr(bias), cos.data_ptr(), sin.data_ptr(), int(rope_cols), 128, out.data_ptr(),
         dtype_code(out.dtype), out.stride(0), _stream())
    return out


@_on_device
def gemm_w4a8_wgroupscale(a, w_img: torch.Tensor, w_bits: int, N: int, s_w_groups: torch.Tensor, group_k: int, *,
                          s_x0: float = 1.0, s_x1: Optional[float] = None, row_sel: Optional[torch.Tensor] = None,
                          s_x_rows: Optional[torch.Tensor] = None, s_x_groups: Optional[torch.Tensor] = None,
                          bias: Optional[torch.Tensor] = None, out_dtype: torch.dtype = torch.float16,
                          out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Group-wise WEIGHT scales (``--w_groupsize``, ``mq_gemm_w4a8_wgroupscale``): s_w_groups fp32 [K / group_k, N];
    y = (sum_g ((float(acc_g) * s_x_groups[m][g]) * s_w_groups[g][n])) * s_x(m) + bias[n] with s_x(m) the per-tensor /
    token-type / per-token scale (1 with group-wise activation scales of the same group size)."""
    _need_cuda(a, w_img, s_w_groups, row_sel, s_x_rows, s_x_groups, bias, out)
    aptr, lda, M, K_pad = _a_args(a)
    assert s_w_groups.dtype == torch.float32 and s_w_groups.dim() == 2 and s_w_groups.is_contiguous() and s_w_groups.shape[1] == N
    G = s_w_groups.shape[0]
    if s_x_groups is not None:
        assert s_x_groups.dtype == torch.float32 and s_x_groups.is_contiguous() and tuple(s_x_groups.shape) == (M, G)
    if s_x_rows is not None:
        assert s_x_rows.dtype == torch.float32 and s_x_rows.is_contiguous() and s_x_rows.numel() == M
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype, device=s_w_groups.device)
    call("mq_gemm_w4a8_wgroupscale", aptr, lda, w_img.data_ptr(), w_bits, M, N, K_pad, s_w_groups.data_ptr(), G, int(group_k),
         float(s_x0), float(s_x0 if s_x1 is None else s_x1), _ptr(row_sel), _ptr(s_x_rows), _ptr(s_x_groups), _ptr(bias),
         out.data_ptr(), dtype_code(out.dtype), out.stride(0), _stream())
    return out


@_on_device
def gemm_w4a8_groupscale_asym(a, w_img: torch.Tensor, w_bits: int, N: int, s_groups: torch.Tensor, shift_groups: torch.Tensor,
                              wsum_groups: torch.Tensor, group_k: int, s_w: torch.Tensor, *, bias: Optional[torch.Tensor] = None,
                              out_dtype: torch.dtype = torch.float16, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y = (sum_g (float(acc_g) * s_groups[m][g] + shift_groups[m][g] * wsum_groups[g][n])) * s_w[n] + bias[n]
    (``mq_gemm_w4a8_groupscale_asym``); wsum_groups: fp32 [K / group_k, N], the per-group sums of the weight levels."""
    _need_cuda(a, w_img, s_groups, shift_groups, wsum_groups, s_w, bias, out)
    aptr, lda, M, K_pad = _a_args(a)
    G = s_groups.shape[1]
    for t in (s_groups, shift_groups):
        assert t.dtype == torch.float32 and t.is_contiguous() and tuple(t.shape) == (M, G)
    assert wsum_groups.dtype == torch.float32 and wsum_groups.is_contiguous() and tuple(wsum_groups.shape) == (G, N)
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype, device=w_img.device)
    call("mq_gemm_w4a8_groupscale_asym", aptr, lda, w_img.data_ptr(), w_bits, M, N, K_pad, s_groups.data_ptr(),
         shift_groups.data_ptr(), wsum_groups.data_ptr(), G, int(group_k), s_w.data_ptr(), _ptr(bias), out.data_ptr(),
         dtype_code(out.dtype), out.stride(0), _stream())
    return out


@_on_device
def quantize_act_dyn_asym_i8(x: torch.Tensor, bits: int = 8, clip_ratio: float = 1.0, *, out=None, tiled: bool = False):
    """Dynamic ASYMMETRIC per-token quantizer (``--a_asym``).  Returns (stored int8 levels q - 2^(bits-1)
    [M, ceil128(K)], scale [M], zero [M], shift [M] = scale * (2^(bits-1) - zero)); dequantised value =
    scale * stored + shift."""
    x2 = _rows(x)
    _need_cuda(x2, out)
    M, K = x2.shape
    out, optr, K_pad, ldo = _out_act(out, tiled, M, ceil_to(K, 128), x.device)
    scale, zero, shift = (torch.empty((M,), dtype=torch.float32, device=x.device) for _ in range(3))
    call("mq_quantize_act_dyn_asym_i8", x2.data_ptr(), dtype_code(x2.dtype), M, K, x2.stride(0), int(bits),
         float(clip_ratio), scale.data_ptr(), zero.data_ptr(), shift.data_ptr(), optr, K_pad, ldo, _stream())
    return out, scale, zero, shift


@_on_device
def quantize_act_tensor_i8(x: torch.Tensor, bits: int = 8, clip_ratio: float = 1.0, *, asym: bool = False,
                           skip_col0: bool = False, out=None, tiled: bool = False):
    """Dynamic PER-TENSOR quantizer (``act_per_tensor``): min/max of the whole tensor on the device
    (``mq_minmax_tensor``), then ``mq_quantize_act_range_i8``.  Returns (int8 levels, scale [M], zero [M] | None,
    shift [M] | None, column 0 as fp32 [M] | None)."""
    x2 = _rows(x)
    _need_cuda(x2, out)
    assert not (asym and skip_col0)
    M, K = x2.shape
    rng = minmax_tensor(x2, 1 if skip_col0 else 0)
    out, optr, K_pad, ldo = _out_act(out, tiled, M, ceil_to(K, 128), x.device)
    scale = torch.empty((M,), dtype=torch.float32, device=x.device)
    zero = torch.empty((M,), dtype=torch.float32, device=x.device) if asym else None
    shift = torch.empty((M,), dtype=torch.float32, device=x.device) if asym else None
    x0 = torch.empty((M,), dtype=torch.float32, device=x.device) if skip_col0 else None
    call("mq_quantize_act_range_i8", x2.data_ptr(), dtype_code(x2.dtype), M, K, x2.stride(0), int(bits),
         float(clip_ratio), int(asym), int(skip_col0), rng.data_ptr(), _ptr(x0), scale.data_ptr(), _ptr(zero),
         _ptr(shift), optr, K_pad, ldo, _stream())
    return out, scale, zero, shift, x0


@_on_device
def act_rowsum_scaled(a, s_x0: float = 1.0, s_x1: Optional[float] = None, row_sel: Optional[torch.Tensor] = None,
                      s_x_rows: Optional[torch.Tensor] = None) -> torch.Tensor:
    """s_x(m) * sum_k a[m][k] per row of the int8 activations (``mq_act_rowsum_scaled``): the row factor of
    the rank-1 epilogue term that carries the zero points of asymmetric weights."""
    _need_cuda(a if not isinstance(a, TiledAct) else a.data, row_sel, s_x_rows)
    aptr, lda, M, K_pad = _a_args(a)
    dev = a.data.device if isinstance(a, TiledAct) else a.device
    out = torch.empty((M,), dtype=torch.float32, device=dev)
    call("mq_act_rowsum_scaled", aptr, lda, M, K_pad, float(s_x0), float(s_x0 if s_x1 is None else s_x1),
         _ptr(row_sel), _ptr(s_x_rows), out.data_ptr(), _stream())
    return out


@_on_device
def gemm_w4a8_rowscale(a: torch.Tensor, w_img: torch.Tensor, w_bits: int, N: int, s_x_rows: torch.Tensor,
                       s_w: torch.Tensor, *, bias: Optional[torch.Tensor] = None, x0: Optional[torch.Tensor] = None,
                       w0: Optional[torch.Tensor] = None, out_dtype: torch.dtype = torch.float16,
                       out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _need_cuda(a, w_img, s_x_rows, s_w, bias, x0, w0, out)
    assert s_x_rows.dtype == torch.float32
    aptr, lda, M, K_pad = _a_args(a)
    assert s_x_rows.numel() == M and s_x_rows.is_contiguous()
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype, device=a.device)
    ws = splitk_workspace(a.device)
    call("mq_gemm_w4a8_rowscale_ws", aptr, lda, w_img.data_ptr(), w_bits, M, N, K_pad,
         s_x_rows.data_ptr(), s_w.data_ptr(), _ptr(bias), _ptr(x0), _ptr(w0), out.data_ptr(),
         dtype_code(out.dtype), out.stride(0), _ptr(ws), 0 if ws is None else ws.numel(), _stream())
    return out


@_on_device
def gemm_w4a8_rank2(a: torch.Tensor, w_img: torch.Tensor, w_bits: int, N: int, s_w: torch.Tensor, x0: torch.Tensor, w0: torch.Tensor,
                    x1: torch.Tensor, w1: torch.Tensor, *, s_x0: float = 1.0, s_x1: Optional[float] = None,
                    row_sel: Optional[torch.Tensor] = None, s_x_rows: Optional[torch.Tensor] = None,
                    bias: Optional[torch.Tensor] = None, out_dtype: torch.dtype = torch.float16,
                    out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The GEMM with TWO rank-1 epilogue terms, y += x0[m] * w0[n] + x1[m] * w1[n] (``mq_gemm_w4a8_rank2_ws``): the flag
    combinations that need the slot twice (asymmetric weights / asymmetric activations / the split column, two at a time).
    ``s_x_rows`` (one scale per row, dynamic quantizers) overrides the static scale set."""
    _need_cuda(a, w_img, s_w, x0, w0, x1, w1, row_sel, s_x_rows, bias, out)
    aptr, lda, M, K_pad = _a_args(a)
    for t in (x0, x1):
        assert t.dtype == torch.float32 and t.numel() == M and t.is_contiguous()
    for t in (w0, w1):
        assert t.dtype == torch.float32 and t.numel() == N and t.is_contiguous()
    if s_x_rows is not None:
        assert s_x_rows.dtype == torch.float32 and s_x_rows.numel() == M and s_x_rows.is_contiguous()
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype, device=a.device)
    ws = splitk_workspace(a.device)
    call("mq_gemm_w4a8_rank2_ws", aptr, lda, w_img.data_ptr(), w_bits, M, N, K_pad, float(s_x0),
         float(s_x0 if s_x1 is None else s_x1), _ptr(row_sel), _ptr(s_x_rows), s_w.data_ptr(), _ptr(bias),
         x0.data_ptr(), w0.data_ptr(), x1.data_ptr(), w1.data_ptr(), out.data_ptr(), dtype_code(out.dtype), out.stride(0),
         _ptr(ws), 0 if ws is None else ws.numel(), _stream())
    return out


@_on_device
def rank1_add_cast(y32: torch.Tensor, x: torch.Tensor, w: torch.Tensor, out_dtype: torch.dtype, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """cast(y32 + x[m] * w[n]) (``mq_rank1_add_cast``): the third rank-1 term behind ``gemm_w4a8_rank2(..., out_dtype=float32)``."""
    _need_cuda(y32, x, w, out)
    M, N = y32.shape
    assert y32.dtype == torch.float32 and y32.stride(1) == 1 and x.dtype == torch.float32 and w.dtype == torch.float32
    assert x.numel() == M and w.numel() == N and x.is_contiguous() and w.is_contiguous()
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype, device=y32.device)
    call("mq_rank1_add_cast", y32.data_ptr(), M, N, y32.stride(0), x.data_ptr(), w.data_ptr(), out.data_ptr(), dtype_code(out.dtype),
         out.stride(0), _stream())
    return out


@_on_device
def gemm_w4a8_residual(a: torch.Tensor, w_img: torch.Tensor, w_bits: int, N: int, s_x0: float, s_w: torch.Tensor,
                       residual: torch.Tensor, *, s_x1: Optional[float] = None, row_sel: Optional[torch.Tensor] = None,
                       bias: Optional[torch.Tensor] = None, x0: Optional[torch.Tensor] = None,
                       w0: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """residual + Linear(x) in one launch; the output has the residual's dtype (may alias it)."""
    _need_cuda(a, w_img, s_w, residual, row_sel, bias, x0, w0, out)
    aptr, lda, M, K_pad = _a_args(a)
    assert residual.shape == (M, N) and residual.stride(1) == 1
    if out is None:
        out = torch.empty((M, N), dtype=residual.dtype, device=a.device)
    assert out.dtype == residual.dtype
    ws = splitk_workspace(a.device)
    call("mq_gemm_w4a8_residual_ws", aptr, lda, w_img.data_ptr(), w_bits, M, N, K_pad,
         float(s_x0), float(s_x0 if s_x1 is None else s_x1), _ptr(row_sel), s_w.data_ptr(), _ptr(bias), _ptr(x0),
         _ptr(w0), residual.data_ptr(), residual.stride(0), out.data_ptr(), dtype_code(out.dtype), out.stride(0),
         _ptr(ws), 0 if ws is None else ws.numel(), _stream())
    return out


@_on_device
def gemm_w4a8_i32(a: torch.Tensor, w_img: torch.Tensor, w_bits: int, N: int,
                  use_workspace: bool = True) -> torch.Tensor:
    _need_cuda(a, w_img)
    aptr, lda, M, K_pad = _a_args(a)
    acc = torch.empty((M, N), dtype=torch.int32, device=a.device)
    ws = splitk_workspace(a.device) if use_workspace else None
    call("mq_gemm_w4a8_i32_ws", aptr, lda, w_img.data_ptr(), w_bits, M, N, K_pad,
         acc.data_ptr(), acc.stride(0), _ptr(ws), 0 if ws is None else ws.numel(), _stream())
    return acc


# --------------------------------------------------------------------------- observers
@_on_device
def minmax_channels(x: torch.Tensor, col_begin: int = 0) -> Tuple[torch.Tensor, torch.Tensor]:
    x2 = _rows(x)
    _need_cuda(x2)
    M, Cn = x2.shape
    mn = torch.empty((Cn - col_begin,), dtype=torch.float32, device=x.device)
    mx = torch.empty_like(mn)
    call("mq_minmax_channels", x2.data_ptr(), dtype_code(x2.dtype), M, Cn, x2.stride(0), col_begin,
         mn.data_ptr(), mx.data_ptr(), _stream())
    return mn, mx


@_on_device
def minmax_tensor(x: torch.Tensor, col_begin: int = 0) -> torch.Tensor:
    """Returns a device tensor [min, max] (fp32)."""
    x2 = _rows(x)
    _need_cuda(x2)
    M, Cn = x2.shape
    out = torch.empty((2,), dtype=torch.float32, device=x.device)
    call("mq_minmax_tensor", x2.data_ptr(), dtype_code(x2.dtype), M, Cn, x2.stride(0), col_begin,
         out.data_ptr(), _stream())
    return out


def had_scale(n: int) -> float:
    return 1.0 / math.sqrt(n)
