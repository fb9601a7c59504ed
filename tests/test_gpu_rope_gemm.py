"""RoPE folded into the store of the fused q|k|v GEMM (mq_gemm_w4a8_rope_ws, whole-prefill glue): bit-identical to the
GEMM followed by the one-launch rotation it replaces (mq_rope_inplace), for every wave-specialised tile."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
torch.set_grad_enabled(False)


def tables(T, dtype, base=1e6):
    inv = 1.0 / (base ** (torch.arange(0, 128, 2, device=DEV, dtype=torch.float32) / 128))
    ang = torch.arange(T, device=DEV, dtype=torch.float32)[:, None] * inv[None, :]
    ang = torch.cat([ang, ang], dim=-1)
    return ang.cos().to(dtype).contiguous(), ang.sin().to(dtype).contiguous()


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,N,K,cols", [(768, 4608, 3584, 4096), (100, 384, 256, 256), (37, 1280, 640, 1280), (300, 256, 1408, 128)])
def test_rotation_in_the_store_equals_gemm_then_rope(dtype, M, N, K, cols):
    from mquant_amd import ops
    rng = np.random.default_rng(M + N + K)
    a = ops.TiledAct.from_rows(torch.from_numpy(rng.integers(-128, 128, size=(M, K), dtype=np.int8)).to(DEV))
    img = ops.prepack(torch.from_numpy(rng.integers(-8, 8, size=(N, K), dtype=np.int8)).to(DEV), 4)
    s_w = torch.from_numpy(rng.uniform(0.001, 0.01, size=N).astype(np.float32)).to(DEV)
    bias = torch.from_numpy(rng.normal(size=N).astype(np.float32)).to(DEV)
    sel = torch.from_numpy((rng.random(M) < 0.4).astype(np.uint8)).to(DEV)
    cos, sin = tables(M, dtype)
    try:
        for tile in (-1, 40, 41, 42, 43, 44, 45, 46, 47, 48):
            ops.gemm_debug_force(tile, 0)
            want = ops.gemm_w4a8(a, img, 4, N, 0.02, s_w, s_x1=0.007, row_sel=sel, bias=bias, out_dtype=dtype)
            ops.rope_inplace(want[:, :cols], cols // 128, 128, cos, sin)
            got = ops.gemm_w4a8_rope(a, img, 4, N, 0.02, s_w, cos, sin, cols, s_x1=0.007, row_sel=sel, bias=bias, out_dtype=dtype)
            assert torch.equal(got, want), f"tile {tile}"
            assert got[:, :cols].float().abs().sum() > 0
        # a forced tile outside the wave-specialised family is ignored (the rotation lives in their epilogue), not mis-run
        ops.gemm_debug_force(14, 0)
        assert torch.equal(ops.gemm_w4a8_rope(a, img, 4, N, 0.02, s_w, cos, sin, cols, s_x1=0.007, row_sel=sel, bias=bias, out_dtype=dtype), want)
    finally:
        ops.gemm_debug_force(-1, 0)


def test_what_the_fused_form_cannot_do_is_refused():
    from mquant_amd import ops
    from mquant_amd._lib import MQuantHipError
    a = torch.zeros((16, 256), dtype=torch.int8, device=DEV)
    at = ops.TiledAct.from_rows(a)
    img = ops.prepack(torch.zeros((256, 256), dtype=torch.int8, device=DEV), 4)
    s_w = torch.ones(256, device=DEV)
    cos, sin = tables(16, torch.float16)
    with pytest.raises(MQuantHipError):                      # row-major activations
        ops.gemm_w4a8_rope(a, img, 4, 256, 1.0, s_w, cos, sin, 128)
    with pytest.raises(MQuantHipError):                      # heads must be whole tiles
        ops.gemm_w4a8_rope(at, img, 4, 256, 1.0, s_w, cos, sin, 192)
    with pytest.raises(MQuantHipError):                      # more rotated columns than outputs
        ops.gemm_w4a8_rope(at, img, 4, 256, 1.0, s_w, cos, sin, 384)
    assert ops.gemm_w4a8_rope(at, img, 4, 256, 1.0, s_w, cos, sin, 256).shape == (16, 256)
