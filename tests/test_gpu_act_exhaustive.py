"""The fused activations of the 16-bit dtypes are EXACTLY the reference functions: csrc/mq_common.h evaluates silu and sigmoid of a
half-precision value with ~14 instead of ~25 instructions (no range selects in exp, V_RCP_F32 + one Newton step instead of the IEEE
division sequence) and claims the same rounded result on every input.  A function of a half-precision argument has 65 536 inputs:
this test runs ALL of them, fp16 and bf16, fast form against reference form (the device library's expf + IEEE division, i.e. what
torch's kernels execute for the HF modules around the reference's wrapped Linears, fake_quant/quant_utils.py:330-391)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _table(dtype, which, x_bits, u_bits=None):
    from mquant_amd import ops
    code = ops.dtype_code(dtype)
    n = x_bits.numel()
    fast = torch.empty(n, dtype=torch.int16, device=DEV)
    ref = torch.empty(n, dtype=torch.int16, device=DEV)
    ops.call("mq_debug_act_table", code, which, x_bits.data_ptr(), None if u_bits is None else u_bits.data_ptr(), n,
             fast.data_ptr(), ref.data_ptr(), ops._stream())
    torch.cuda.synchronize()
    return fast.view(dtype), ref.view(dtype)


def _same(a, b):
    """identical bits, NaN for NaN"""
    nan = torch.isnan(a) & torch.isnan(b)
    return bool(((a.view(torch.int16) == b.view(torch.int16)) | nan).all())


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("which,name", [(0, "silu"), (1, "sigmoid"), (3, "quick_gelu")])
def test_every_input_of_the_dtype(dtype, which, name):
    bits = torch.arange(-32768, 32768, dtype=torch.int32, device=DEV).to(torch.int16)          # all 65 536 patterns
    fast, ref = _table(dtype, which, bits)
    bad = ~((fast.view(torch.int16) == ref.view(torch.int16)) | (torch.isnan(fast) & torch.isnan(ref)))
    assert not bool(bad.any()), (name, int(bad.sum()), bits[bad][:8].tolist(), fast[bad][:8].tolist(), ref[bad][:8].tolist())
    # and the reference form is torch's own function on that tensor (finite inputs; torch rounds once per op, as the form does)
    x = bits.view(dtype)
    fin = torch.isfinite(x)
    xf = x.float()
    if which == 0:
        want = torch.nn.functional.silu(xf).to(dtype)
    elif which == 1:
        want = torch.sigmoid(xf).to(dtype)
    else:
        want = (xf * torch.sigmoid((1.702 * xf).to(dtype).float()).to(dtype).float()).to(dtype)
    assert _same(ref[fin], want[fin])


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("which", [2, 4, 5])
def test_products_and_packed_forms(dtype, which):
    """silu(x) * u and the two-at-a-time forms of the GEMM act epilogues (V_CVT_PK_*, V_PK_MUL_F16): every x against 64 different u."""
    g = torch.Generator(device=DEV).manual_seed(which)
    bits = torch.arange(-32768, 32768, dtype=torch.int32, device=DEV).to(torch.int16).repeat(64)
    u = (torch.randn(bits.numel(), generator=g, device=DEV) * 3).to(dtype)
    u[::97] = 0
    u[5::1013] = float("inf")
    fast, ref = _table(dtype, which, bits, u.view(torch.int16))
    assert _same(fast, ref)
