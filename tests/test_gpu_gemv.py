"""``mq_gemv_f16``: a few rows against a large 16-bit matrix (the unquantized lm_head on the last position of a prefill; the
reference keeps lm_head in 16 bits, exam/quant_qwen2vl.py:130-143).  Glue of the whole-prefill report, not the W4A8 path: the
checker is the float64 product, the tolerance what one rounding to the output dtype plus fp32 accumulation over K allows."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
torch.set_grad_enabled(False)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", [(1, 152064, 3584), (4, 92553, 4096), (1, 7, 8), (3, 1001, 1288), (8, 4099, 4096), (2, 300, 8192), (2, 5, 3584)])
def test_gemv_equals_the_float64_product(dtype, M, N, K):
    from mquant_amd import ops
    g = torch.Generator(device=DEV).manual_seed(M + N + K)
    x = torch.randn((M, K), generator=g, device=DEV).to(dtype)
    w = (torch.randn((N, K), generator=g, device=DEV) * 0.02).to(dtype)
    got = ops.gemv_f16(x, w)
    assert got.shape == (M, N) and got.dtype == dtype
    want = x.double() @ w.double().t()
    eps = 2.0 ** -11 if dtype == torch.float16 else 2.0 ** -8
    # one rounding of the result (eps * |y|) + fp32 accumulation error (K * 2^-24 * sum |x w|, a loose bound)
    bound = eps * want.abs() + K * 2.0 ** -24 * (x.double().abs() @ w.double().abs().t()) + 1e-7
    assert bool(((got.double() - want).abs() <= bound).all()), float(((got.double() - want).abs() / bound).max())
    # and next to the op it replaces
    ref = x @ w.t()
    assert float((got.float() - ref.float()).abs().max()) <= 4 * eps * float(want.abs().max()) + 1e-6


def test_strided_operands_and_out_buffer():
    from mquant_amd import ops
    g = torch.Generator(device=DEV).manual_seed(1)
    xw = torch.randn((2, 2 * 512), generator=g, device=DEV).half()
    ww = (torch.randn((300, 3 * 512), generator=g, device=DEV) * 0.05).half()
    x, w = xw[:, :512], ww[:, 512:1024]                         # row strides larger than K
    out = torch.full((2, 1024), 7.0, dtype=torch.float16, device=DEV)
    ops.gemv_f16(x, w, out=out[:, 100:400])
    want = (x.double() @ w.double().t())
    assert float((out[:, 100:400].double() - want).abs().max()) < 2e-3 * float(want.abs().max())
    assert bool((out[:, :100] == 7.0).all()) and bool((out[:, 400:] == 7.0).all())


def test_bad_arguments_are_refused():
    from mquant_amd import ops
    from mquant_amd._lib import MQuantHipError
    x = torch.zeros((9, 64), dtype=torch.float16, device=DEV)
    w = torch.zeros((16, 64), dtype=torch.float16, device=DEV)
    with pytest.raises(MQuantHipError):
        ops.gemv_f16(x, w)                                      # more than 8 rows
    with pytest.raises(MQuantHipError):
        ops.gemv_f16(x[:1, :60], w[:, :60])                     # K not a multiple of 8
    assert ops.gemv_f16(x[:0], w).shape == (0, 16)
