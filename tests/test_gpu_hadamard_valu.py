"""The vector-ALU exact Hadamard kernel (csrc/hadamard_valu.hip: half-precision activations, co-factor n / K = 128, K x K stage
as sequential v_pk_fma_f32 chains with scalar sign operands) against the matrix-core exact kernel (csrc/hadamard.hip), which
the goldens pin to the reference: every output bit for bit -- rotated values, int8 levels in both layouts, the split column,
both scale sets, the fused activation prologue, ragged input widths and row counts.  (test_gpu_kernels.py / test_gpu_tiled.py /
the wrapper goldens run the default path, i.e. this kernel wherever it applies, against the oracle and the reference.)"""
import numpy as np
import pytest
import torch

from golden_inputs import make_x

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
SHAPES = [(40, 5120), (52, 6656), (20, 2560), (60, 7680)]       # (K, n) with n / K = 128 that the kernel serves


def _rows(q):
    """row-major view of a quantizer result in either layout"""
    return q.to_rows() if hasattr(q, "to_rows") else q


def _classic(fn):
    from mquant_amd import ops
    ops.hadamard_debug_impl(1)
    try:
        return fn()
    finally:
        ops.hadamard_debug_impl(0)


@pytest.mark.parametrize("K,n", SHAPES)
@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_rotated_values_bit_for_bit(K, n, dt):
    from fake_quant import hadamard_utils as hu
    from mquant_amd import ops
    bits = hu.had_sign_bits(K, DEV)
    for M, n_in in ((1, n), (7, n - n // 16), (67, n - 8), (33, 8)):
        x = torch.from_numpy(make_x(K + M, (M, n_in))).to(DEV).to(dt)
        got = ops.hadamard(x, n, K, bits)
        want = _classic(lambda: ops.hadamard(x, n, K, bits))
        assert torch.equal(got.view(torch.int16), want.view(torch.int16)), (M, n_in)


@pytest.mark.parametrize("K,n", SHAPES)
@pytest.mark.parametrize("tiled", [True, False])
@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_levels_bit_for_bit(K, n, tiled, dt):
    from fake_quant import hadamard_utils as hu
    from mquant_amd import ops
    bits = hu.had_sign_bits(K, DEV)
    rng = np.random.default_rng(K)
    for M, n_in, split, msq in ((130, n - n // 16, False, True), (19, n, True, False), (1024 if K == 40 else 257, n - 1024, True, True)):
        x = torch.from_numpy(make_x(3 * K + M, (M, n_in))).to(DEV).to(dt)
        sel = torch.from_numpy((rng.random(M) < 0.4).astype(np.uint8)).to(DEV) if msq else None
        s0 = float(x.float().abs().max()) / 40.0
        args = dict(row_sel=sel, skip_col0=split, tiled=tiled)

        def run():
            return ops.hadamard_quant_i8(x, n, K, bits, s0, 0.37 * s0, **args)
        q, x0 = run()
        qc, x0c = _classic(run)
        assert torch.equal(_rows(q), _rows(qc)), (M, n_in, split, msq)
        if split:
            assert torch.equal(x0, x0c)


@pytest.mark.parametrize("act", [1, 2])
@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_fused_activation_prologue_bit_for_bit(act, dt):
    from fake_quant import hadamard_utils as hu
    from mquant_amd import ops
    K, n, n_in, M = (52, 6656, 6400, 96) if act == 1 else (40, 5120, 5120, 200)
    bits = hu.had_sign_bits(K, DEV)
    g = torch.from_numpy(make_x(11, (M, n_in))).to(DEV).to(dt)
    u = torch.from_numpy(make_x(12, (M, n_in))).to(DEV).to(dt) if act == 1 else None

    def run():
        return ops.act_hadamard_quant_i8(g, u, act, n, K, bits, 0.05, 0.02, tiled=True)[0]
    q = run()
    qc = _classic(run)
    assert torch.equal(_rows(q), _rows(qc))
