"""mq_act_hadamard_quant_i8: silu(gate)*up / quick_gelu fused in front of the online Hadamard +
quantizer.  These activations are torch / HF code, not MQuant's: the bar is the torch
composition on the same GPU followed by the unfused kernel, bit for bit (the fused kernel calls the
same device expf).  The C oracle (library-free exp) agrees up to its last-bit freedom."""
import numpy as np
import pytest
import torch

import oracle
from golden_inputs import make_x

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
torch.set_grad_enabled(False)


def words(had_table, K):
    return torch.from_numpy(np.ascontiguousarray(had_table["words"][K])).to(DEV)


@pytest.mark.parametrize("M,n_in,n,K,dtype", [(48, 18944, 19968, 156, torch.float16), (33, 5120, 5120, 40, torch.float16),
                                              (7, 700, 768, 12, torch.bfloat16), (5, 512, 512, 1, torch.float32)])
def test_silu_mul_prologue(had_table, M, n_in, n, K, dtype):
    from mquant_amd import ops
    mode = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}[dtype]
    gu = torch.from_numpy(make_x(M + n, (M, 2 * n_in))).to(device=DEV, dtype=dtype) * 2
    g, u = gu[:, :n_in], gu[:, n_in:]
    bits = words(had_table, K) if K > 1 else None
    s0, s1 = 0.05, 0.021
    sel = (torch.arange(M, device=DEV) % 2).to(torch.uint8)
    q, _ = ops.act_hadamard_quant_i8(g, u, ops.ACT_SILU_MUL, n, K, bits, s0, s1, row_sel=sel)
    h = oracle.silu_mul(g.float().cpu().numpy(), u.float().cpu().numpy(), mode).reshape(M, n_in)
    rot = oracle.hadamard(h, n, K, had_table["mats"].get(K), mid_round=mode, out_round=mode)
    want = oracle.quant_static(rot, np.float32(s0), scale1=np.float32(s1), row_sel=sel.cpu().numpy())
    # the torch ops this launch replaces: identical
    ref, _ = ops.hadamard_quant_i8((torch.nn.functional.silu(g) * u).contiguous(), n, K, bits, s0, s1, row_sel=sel)
    assert torch.equal(q, ref)
    # the CPU oracle (its exp is correctly rounded, the device library's is not always)
    got = q.cpu().numpy()[:, :n]
    assert (got != want).mean() < 1e-3 and np.abs(got.astype(np.int32) - want).max() <= 1


def test_quick_gelu_prologue_with_split(had_table):
    from mquant_amd import ops
    M, n, K = 40, 5120, 40
    x = torch.from_numpy(make_x(9, (M, n))).to(DEV).half() * 2
    q, x0 = ops.act_hadamard_quant_i8(x, None, ops.ACT_QUICK_GELU, n, K, words(had_table, K), 0.04,
                                      skip_col0=True)
    h = oracle.quick_gelu(x.float().cpu().numpy(), 1).reshape(M, n)
    rot = oracle.hadamard(h, n, K, had_table["mats"][K], mid_round=1, out_round=1)
    want = oracle.quant_static(rot, np.float32(0.04))
    want[:, 0] = 0
    ref, x0_ref = ops.hadamard_quant_i8((x * torch.sigmoid(1.702 * x)).contiguous(), n, K, words(had_table, K), 0.04,
                                        skip_col0=True)
    assert torch.equal(q, ref) and torch.equal(x0, x0_ref)
    got = q.cpu().numpy()[:, :n]
    assert (got != want).mean() < 1e-3 and np.abs(got.astype(np.int32) - want).max() <= 1
    np.testing.assert_allclose(x0.cpu().numpy(), rot[:, 0], rtol=0, atol=2e-3)


def test_engine_forward_from_the_fused_gate_up_output(had_table):
    from mquant_amd import ops
    from mquant_amd.engine import HadamardSpec, W4A8Linear
    M, n_in, n, K = 64, 700, 768, 12
    gu = torch.from_numpy(make_x(3, (M, 2 * n_in))).to(DEV).half()
    w = torch.randn((96, n), device=DEV).half() * 0.02
    lin = W4A8Linear.from_float(w, 4, s_x0=0.03, had=HadamardSpec(n, K, words(had_table, K)), in_features=n_in)
    a, x0 = lin.quantize_act(gu[:, :n_in], gu[:, n_in:], ops.ACT_SILU_MUL)
    fused = lin.gemm(a, x0, torch.float16).clone()
    unfused = lin((torch.nn.functional.silu(gu[:, :n_in]) * gu[:, n_in:]).contiguous())
    torch.testing.assert_close(fused, unfused, rtol=0, atol=0)


def test_argument_checks():
    from mquant_amd import ops
    from mquant_amd._lib import MQuantHipError
    x = torch.zeros((4, 64), device=DEV, dtype=torch.float16)
    with pytest.raises(MQuantHipError):
        ops.act_hadamard_quant_i8(x, None, ops.ACT_SILU_MUL, 64, 1, None, 0.1)       # no second operand
    with pytest.raises(MQuantHipError):
        ops.act_hadamard_quant_i8(x, x, 7, 64, 1, None, 0.1)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16, torch.float32])
@pytest.mark.parametrize("T,heads,kvh,d", [(77, 6, 2, 64), (768, 28, 4, 128), (1024, 16, 16, 80), (5, 3, 1, 24)])
def test_rope_inplace_equals_the_torch_formula(dtype, T, heads, kvh, d):
    """Harness glue (not MQuant): rotate-half RoPE on the q|k part of a fused q|k|v output (16-byte vector kernel for
    half dtypes and head_dim % 16 == 0, scalar kernel otherwise -- same arithmetic)."""
    from mquant_amd import ops
    from mquant_amd.full_prefill import _rope, _rope_tables
    qkv = torch.from_numpy(make_x(3, (T, (heads + 2 * kvh) * d))).to(device=DEV, dtype=dtype)
    cos, sin = _rope_tables(T, d, torch.device(DEV), dtype)
    want_q = _rope(qkv[:, :heads * d].view(T, heads, d), cos, sin)
    want_k = _rope(qkv[:, heads * d:(heads + kvh) * d].view(T, kvh, d), cos, sin)
    v_before = qkv[:, (heads + kvh) * d:].clone()
    ops.rope_inplace(qkv[:, :(heads + kvh) * d], heads + kvh, d, cos[:, 0].contiguous(), sin[:, 0].contiguous())
    assert torch.equal(qkv[:, :heads * d].view(T, heads, d), want_q)
    assert torch.equal(qkv[:, heads * d:(heads + kvh) * d].view(T, kvh, d), want_k)
    assert torch.equal(qkv[:, (heads + kvh) * d:], v_before)
