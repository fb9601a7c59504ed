"""mq_rmsn_quantize_i8 (RMSN + static quantizer) against the oracle (bit-exact) and
against the reference composition module_util.RMSN -> UniformQuantizer (within the 1-ulp
freedom of the summation order)."""
import numpy as np
import pytest
import torch

import oracle
from golden_inputs import make_x

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
torch.set_grad_enabled(False)


@pytest.mark.parametrize("M,K,dtype", [(768, 3584, torch.float16), (1024, 1280, torch.float16), (5, 16, torch.float16),
                                       (33, 5120, torch.float32), (3, 16384, torch.float16), (64, 4112, torch.float16),
                                       (768, 3584, torch.bfloat16), (9, 1280, torch.bfloat16)])
def test_matches_oracle_bit_for_bit(M, K, dtype):
    from mquant_amd import ops
    x = torch.from_numpy(make_x(M + K, (M, K))).to(device=DEV, dtype=dtype)
    mode = {torch.float16: 1, torch.float32: 0, torch.bfloat16: 2}[dtype]
    sel = (torch.arange(M, device=DEV) % 3 == 0).to(torch.uint8)
    q, y = ops.rmsn_quantize_i8(x, K, 1e-6, 0.031, 0.017, row_sel=sel, want_y=True)
    y_ref = oracle.rmsn(x.float().cpu().numpy(), K, 1e-6, mode)
    np.testing.assert_array_equal(y.float().cpu().numpy(), y_ref)
    q_ref = oracle.quant_static(y_ref, np.float32(0.031), scale1=np.float32(0.017), row_sel=sel.cpu().numpy())
    K_pad = (K + 127) // 128 * 128
    assert q.shape == (M, K_pad)
    np.testing.assert_array_equal(q.cpu().numpy()[:, :K], q_ref)
    assert not q[:, K:].any()


def test_within_one_ulp_of_the_reference_composition():
    from fake_quant.module_util import RMSN
    from mquant_amd import ops
    x = torch.from_numpy(make_x(11, (768, 3584))).to(DEV).half()
    want = RMSN(3584, eps=1e-6)(x)                          # torch reduction order
    q, y = ops.rmsn_quantize_i8(x, 3584, 1e-6, 0.02, want_y=True)
    diff = (y.float() - want.float()).abs()
    ulp = torch.finfo(torch.float16).eps * want.float().abs().clamp(min=2.0 ** -14)
    assert bool((diff <= ulp).all())
    assert float((y != want).float().mean()) < 1e-3
    q_ref, _ = ops.quantize_act_i8(want, 0.02)
    assert float((q != q_ref).float().mean()) < 1e-3 and int((q.int() - q_ref.int()).abs().max()) <= 1


def test_feeds_the_gemm_like_the_unfused_pair():
    from mquant_amd import ops
    from mquant_amd.engine import W4A8Linear
    x = torch.from_numpy(make_x(5, (96, 1280))).to(DEV).half()
    w = torch.randn((256, 1280), device=DEV).half() * 0.02
    lin = W4A8Linear.from_float(w, 4, s_x0=0.03)
    a, y = ops.rmsn_quantize_i8(x, 1280, 1e-6, 0.03, want_y=True)
    fused = lin.gemm(a, None, torch.float16)
    torch.testing.assert_close(fused, lin(y), rtol=0, atol=0)


def test_bf16_rows_follow_the_unpromoted_torch_ops():
    """Qwen2-VL's native dtype: RMSN leaves bf16 tensors in bf16 (module_util.py:56-57 promotes fp16 only)."""
    from fake_quant.module_util import RMSN
    from mquant_amd import ops
    x = torch.from_numpy(make_x(21, (768, 3584))).to(DEV).bfloat16()
    want = RMSN(3584, eps=1e-6)(x)
    q, y = ops.rmsn_quantize_i8(x, 3584, 1e-6, 0.02, want_y=True)
    assert float((y != want).float().mean()) < 1e-3                 # the row sum's order is torch's own
    q_ref, _ = ops.quantize_act_i8(want, 0.02)
    assert float((q != q_ref).float().mean()) < 1e-3


def test_rejects_unaligned_rows():
    from mquant_amd import ops
    from mquant_amd._lib import MQuantHipError
    with pytest.raises(MQuantHipError):
        ops.rmsn_quantize_i8(torch.zeros((4, 40), device=DEV, dtype=torch.float16), 40, 1e-6, 0.1)
