"""NON-DEFAULT fast mode of the online Hadamard kernels (``ops.hadamard_fast_mode`` / ``mq_hadamard_set_mode``):
the K x K stage on the half-precision matrix core.  Same exact +-1 products as the exact mode, another fp32
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


@pytest.fixture()
def fast_mode():
    from mquant_amd import ops
    prev = ops.hadamard_fast_mode(True)
    try:
        yield
    finally:
        ops.hadamard_fast_mode(prev)


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
    prev = ops.hadamard_fast_mode(True)
    try:
        fast = ops.hadamard(x, n, K, bits).float()
    finally:
        ops.hadamard_fast_mode(prev)
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
        prev = ops.hadamard_fast_mode(fast)
        try:
            q, x0 = ops.hadamard_quant_i8(x, n, K, bits, s0, s1, row_sel=sel, skip_col0=split, tiled=tiled)
        finally:
            ops.hadamard_fast_mode(prev)
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


def test_fast_mode_with_the_fused_activation_prologue(fast_mode):
    from fake_quant import hadamard_utils as hu
    from mquant_amd import ops
    K, n, n_in, M = 156, 19968, 18944, 9
    bits = hu.had_sign_bits(K, DEV)
    g = torch.from_numpy(make_x(5, (M, n_in))).to(DEV).half()
    u = torch.from_numpy(make_x(6, (M, n_in))).to(DEV).half()
    h = torch.nn.functional.silu(g) * u
    ref, _ = ops.hadamard_quant_i8(h, n, K, bits, 0.02, tiled=True)
    got, _ = ops.act_hadamard_quant_i8(g, u, ops.ACT_SILU_MUL, n, K, bits, 0.02, tiled=True)
    assert torch.equal(ref.to_rows(), got.to_rows())                   # same mode on both sides: identical bits


def test_what_the_fast_mode_does_not_cover_runs_the_exact_kernel_bit_for_bit(fast_mode):
    from fake_quant import hadamard_utils as hu
    from mquant_amd import ops
    cases = [(156, 19968, torch.float32, False),      # fp32 activations are not half numbers
             (40, 5120, torch.float16, True),         # --fp32_had keeps fp32 between the stages
             (1, 8192, torch.float16, False),         # pure power of two: no K x K stage at all
             (12, 384, torch.float16, False)]         # co-factor 32 < 64
    for K, n, dt, fp32_had in cases:
        bits = hu.had_sign_bits(K, DEV) if K > 1 else None
        x = torch.from_numpy(make_x(K + 7, (6, n))).to(DEV).to(dt)
        got = ops.hadamard(x, n, K, bits, fp32_had)
        ops.hadamard_fast_mode(False)
        want = ops.hadamard(x, n, K, bits, fp32_had)
        ops.hadamard_fast_mode(True)
        assert torch.equal(got, want), (K, n, dt)


def test_the_default_is_the_exact_mode():
    from mquant_amd import _lib
    assert _lib.load().mq_hadamard_get_mode() == 0
