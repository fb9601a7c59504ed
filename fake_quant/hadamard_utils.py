"""Hadamard helpers with the reference's public surface (``fake_quant/hadamard_utils.py``).

    y = (H_K (x) H_m) x / sqrt(n),   n = K * m,  m a power of two,  flat index k*m + j

* The 11 non-power-of-two factors (K = 12 ... 172) are stored as packed sign bits in
  ``had_signs.npz`` next to this file (12 KB instead of 97 k lines of literals); they were
  captured from the reference by ``tools/gen_golden.py`` and are validated in the tests both
  against that capture and mathematically (H H^T = K I).
* ``matmul_hadU_cuda`` -- the online transform of the hot path -- runs the hand-written
  gfx950 kernel (``mquant_amd/csrc/hadamard.hip``) for fp16/bf16/fp32 CUDA tensors.  There is
  no third-party ``fast_hadamard_transform`` dependency.
* ``matmul_hadU`` is the device-agnostic torch definition (butterflies, then hadK @, then
  / sqrt(n)) used by the offline fp64 weight surgery, like upstream (:79-100).
"""
from __future__ import annotations

import math
import os
from functools import lru_cache

import numpy as np
import torch

from fake_quant import utils

_HERE = os.path.dirname(os.path.abspath(__file__))

# order in which special factors are tried (reference :28-71; note 40 sits between 28 and 20)
_FACTORS = (172, 156, 140, 108, 60, 52, 36, 28, 40, 20, 12)
# sizes considered by auto_pad_size (reference :6; 40 is absent there)
default_size = [172, 156, 140, 108, 60, 52, 36, 28, 20, 12, 1]


def is_pow2(n) -> bool:
    n = int(n)
    return n > 0 and (n & (n - 1)) == 0


@lru_cache(maxsize=None)
def _signs() -> dict:
    with np.load(os.path.join(_HERE, "had_signs.npz")) as z:
        return {int(k[3:]): z[k].copy() for k in z.files}


@lru_cache(maxsize=None)
def _had_np(K: int) -> np.ndarray:
    bits = np.unpackbits(_signs()[K])[: K * K].reshape(K, K)
    return bits.astype(np.float32) * 2.0 - 1.0


def _make_getter(K: int):
    def getter() -> torch.Tensor:
        return torch.from_numpy(_had_np(K).copy())
    getter.__name__ = f"get_had{K}"
    getter.__doc__ = f"The {K}x{K} +-1 Hadamard factor (fp32 tensor, fresh copy)."
    return getter


for _k in _FACTORS:
    globals()[f"get_had{_k}"] = _make_getter(_k)
del _k

_BITS_CACHE: dict = {}


def sign_words(had: np.ndarray) -> np.ndarray:
    """+-1 matrix (K, K) -> uint32 [K, ceil(K/32)] word-aligned sign rows: bit b of word w of
    row j is 1 when had[j, 32*w + b] > 0.  This is the operand layout of the gfx950 kernel."""
    K = had.shape[0]
    wpr = (K + 31) // 32
    bits = np.zeros((K, wpr * 32), dtype=np.uint8)
    bits[:, :K] = (np.asarray(had) > 0)
    packed = np.packbits(bits.reshape(K, wpr, 32), axis=-1, bitorder="little")
    return np.ascontiguousarray(packed).view("<u4").reshape(K, wpr)


def had_sign_bits(K: int, device, prepared: bool = True) -> torch.Tensor:
    """Sign rows of hadK on ``device``, cached: by default the PREPARED descriptor (int64 tensor: the
    word-aligned sign rows of ``sign_words`` followed by the MFMA lane masks, built once by
    ``mq_hadamard_prepare``); ``prepared=False`` gives the plain int32 words of the C ABI."""
    device = torch.device(device)
    key = (K, device.type, device.index, bool(prepared and device.type == "cuda"))
    t = _BITS_CACHE.get(key)
    if t is None:
        t = torch.from_numpy(sign_words(_had_np(K)).view(np.int32).copy()).to(device)
        if key[3]:
            from mquant_amd import ops
            t = ops.hadamard_prepare(t, K)
        _BITS_CACHE[key] = t
    return t


def auto_pad_size(n: int) -> int:
    """Smallest size >= n of the form size * 2^p (size in ``default_size``); n itself if it
    already factors that way."""
    for size in default_size:
        if n % size == 0 and is_pow2(n // size):
            return n
    best = math.inf
    for size in default_size:
        blocks = 2 ** math.ceil(math.log2(math.ceil(n / size)))
        best = min(best, blocks * size)
    return best


def get_hadK(n: int, transpose: bool = False):
    """(hadK, K): the largest special factor K dividing n (tried in the reference's order)
    whose co-factor is a power of two; (None, 1) for a pure power of two."""
    for K in _FACTORS:
        if n % K == 0:
            assert is_pow2(n // K)
            h = torch.from_numpy(_had_np(K).copy())
            return (h.T if transpose else h), K
    assert is_pow2(n)
    return None, 1


def _butterflies(x: torch.Tensor, m: int) -> torch.Tensor:
    """Unnormalised Walsh-Hadamard over contiguous blocks of m along the last dim,
    strides ascending (a+b, a-b).  Pure torch; any device / float dtype."""
    shape = x.shape
    y = x.reshape(-1, shape[-1])
    h = 1
    while h < m:
        v = y.reshape(y.shape[0], -1, 2, h)
        y = torch.stack((v[:, :, 0, :] + v[:, :, 1, :], v[:, :, 0, :] - v[:, :, 1, :]), dim=2)
        y = y.reshape(-1, shape[-1])
        h *= 2
    return y.reshape(shape)


def matmul_hadU(X: torch.Tensor, transpose: bool = False) -> torch.Tensor:
    n = X.shape[-1]
    hadK, K = get_hadK(n, transpose)
    m = n // K
    y = _butterflies(X.clone(), m)
    if K > 1:
        y = (hadK.to(device=y.device, dtype=y.dtype) @ y.reshape(-1, K, m)).reshape(X.shape)
    return y / torch.tensor(n).sqrt()


def matmul_hadUt(X: torch.Tensor) -> torch.Tensor:
    return matmul_hadU(X, transpose=True)


def random_hadamard_matrix(size: int, device):
    """Q = H diag(+-1): signs from the global CPU RNG (torch.randint), fp64."""
    signs = torch.randint(low=0, high=2, size=(size,)).to(torch.float64) * 2 - 1
    Q = matmul_hadU(torch.diag(signs)).to(device)
    Q._mq_signs = signs            # structure tag: rotation_utils.mul_q runs sign flip + fast Hadamard on the GPU
    Q._mq_tag_version = utils.tensor_version(Q)  # any later in-place edit of Q moves the counter and voids the tag
    return Q


def _bits_for(hadK, K: int, device):
    """Sign bits of the given hadK tensor (honours a caller-supplied / transposed matrix)."""
    if K == 1:
        return None
    ref = _had_np(K)
    if hadK is None or (tuple(hadK.shape) == ref.shape and
                        np.array_equal(np.sign(hadK.detach().cpu().float().numpy()), ref)):
        return had_sign_bits(K, device)
    words = sign_words(hadK.detach().cpu().float().numpy())
    t = torch.from_numpy(words.view(np.int32).copy()).to(device)
    if t.is_cuda:
        from mquant_amd import ops
        t = ops.hadamard_prepare(t, K)
    return t


def _hadamard_torch(X: torch.Tensor, hadK, K: int) -> torch.Tensor:
    """Torch evaluation with the CUDA path's operation order (scale before hadK @).
    Offline weight surgery on CPU / fp64 tensors only -- never the online path."""
    n = X.shape[-1]
    m = n // K
    scale = 1.0 / torch.tensor(n).sqrt()
    y = _butterflies(X.float() if X.dtype in (torch.float16, torch.bfloat16) else X.clone(), m)
    y = (y * scale.to(y.dtype)).to(X.dtype)
    if K > 1:
        y = (hadK.to(device=y.device, dtype=y.dtype) @ y.reshape(-1, K, m)).reshape(X.shape)
    return y


def matmul_hadU_cuda(X: torch.Tensor, hadK, K: int, fp32_had: bool = False) -> torch.Tensor:
    """Online Hadamard over the last dim (reference :115-128).  CUDA tensors in
    fp16/bf16/fp32 run the gfx950 kernel; the result has X's dtype and shape."""
    if not X.is_cuda:
        from mquant_amd._lib import MQuantHipError
        raise MQuantHipError("matmul_hadU_cuda needs a CUDA tensor: the online Hadamard has no "
                             "CPU fallback (use matmul_hadU for offline torch math)")
    if X.dtype == torch.float64:
        return _hadamard_torch(X, hadK, K)
    from mquant_amd import ops
    n = X.shape[-1]
    return ops.hadamard(X, n, K, _bits_for(hadK, K, X.device), fp32_had=fp32_had)


def matmul_hadUt_cuda(X, hadK, K):
    """Transposed variant (upstream's version is broken and has no caller, :131-132)."""
    return matmul_hadU_cuda(X, None if hadK is None else hadK.T.contiguous(), K)


def _rotate_rows(W: torch.Tensor, hadK, K: int) -> torch.Tensor:
    if W.is_cuda and W.dtype != torch.float64:
        return matmul_hadU_cuda(W.contiguous(), hadK, K)
    return _hadamard_torch(W.contiguous(), hadK, K)


def _fht_blocks(W: torch.Tensor, had_dim: int) -> torch.Tensor:
    """Walsh-Hadamard over contiguous chunks of had_dim of the last dim, / sqrt(had_dim)."""
    if W.is_cuda and W.dtype != torch.float64:
        shp = W.shape
        return matmul_hadU_cuda(W.reshape(-1, had_dim).contiguous(), None, 1).reshape(shp)
    y = _butterflies(W.clone(), had_dim)
    return y * (1.0 / math.sqrt(had_dim))


def apply_exact_had_to_linear(module, had_dim: int = -1, output: bool = False) -> None:
    """Offline: rotate a Linear's weight in fp32 (reference :135-191).

    had_dim == -1: full (H_K (x) H_m) over the input features (``output=False``) or over the
    output features incl. the bias (``output=True``).  had_dim > 0: block-diagonal Walsh-
    Hadamard of that size (per attention head).  Runs on the GPU when one is present (as
    upstream, which hard-codes ``.cuda()``), otherwise with torch ops on the CPU.
    """
    assert isinstance(module, torch.nn.Linear)
    in_f, out_f = module.in_features, module.out_features
    if had_dim != -1:
        assert is_pow2(had_dim), "Hadamard dimension must be a power of 2!"
    W = module.weight.data
    dtype, dev = W.dtype, W.device
    work = torch.device("cuda") if torch.cuda.is_available() else W.device
    Wf = W.float().to(work)
    Bf = module.bias.data.float().to(work) if module.bias is not None else None

    if had_dim == -1:
        if output:
            hadK, K = get_hadK(out_f)
            Wf = _rotate_rows(Wf.t(), hadK, K).t()
            if Bf is not None:
                Bf = _rotate_rows(Bf.view(1, -1), hadK, K).view(-1)
        else:
            hadK, K = get_hadK(in_f)
            Wf = _rotate_rows(Wf, hadK, K)
    else:
        if output:
            Wt = Wf.t().contiguous()
            Wf = _fht_blocks(Wt, had_dim).t()
            if Bf is not None:
                Bf = _fht_blocks(Bf.contiguous(), had_dim)
        else:
            Wf = _fht_blocks(Wf.contiguous(), had_dim)
    module.weight.data = Wf.to(device=dev, dtype=dtype)
    if Bf is not None and output:
        module.bias.data = Bf.to(device=dev, dtype=dtype)
