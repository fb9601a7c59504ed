"""The weight-streaming GEMM for a few activation rows (csrc/gemm_skinny.hip, plan id 60: generation steps, M <= 64) against the
oracle and against the tiled kernel it replaces there: int32 accumulators and every epilogue term bit for bit, with one to many K
slices, ragged N, 1 / 2 / 4 row tiles, back-to-back launches on the same counters."""
import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
torch.set_grad_enabled(False)
MODE = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}


def to_dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def _case(M, N, K, seed):
    rng = np.random.default_rng(seed)
    a = rng.integers(-128, 128, size=(M, K), dtype=np.int8)
    a[0, :8] = [-128, 127, -128, 127, -128, -128, 127, 127]
    w = rng.integers(-8, 8, size=(N, K), dtype=np.int8)
    w.reshape(-1)[:4] = [-8, 7, -8, 7]
    s_w = rng.uniform(0.001, 0.01, size=N).astype(np.float32)
    bias = rng.normal(size=N).astype(np.float32)
    return rng, a, w, s_w, bias


@pytest.mark.parametrize("M", [1, 3, 16, 17, 33, 64])
@pytest.mark.parametrize("N,K", [(200, 1280), (4608, 3584), (520, 19968)])
def test_plan_and_every_slice_count_are_exact(M, N, K):
    from mquant_amd import ops as o
    rng, a, w, s_w, bias = _case(M, N, K, M + N + K)
    acc_ref = oracle.gemm_i32(a, w)
    sel = (rng.random(M) < 0.5).astype(np.uint8)
    y_ref = oracle.round_to(oracle.epilogue(acc_ref, np.float32(0.02), s_w, bias=bias, sx1=np.float32(0.007), row_sel=sel), 1)
    at = o.TiledAct.from_rows(to_dev(a))
    img = o.prepack(to_dev(w), 4)
    swt, bt, selt = to_dev(s_w), to_dev(bias), to_dev(sel)
    o.splitk_workspace(torch.device(DEV), 64 << 20)
    tile = torch.zeros(2, dtype=torch.int32)
    o.call("mq_gemm_debug_plan", M, N, at.K_pad, 4, 1, 1, tile[0:].data_ptr(), tile[1:].data_ptr())
    assert (tile[0].item() in (60, 61)) == (M <= 16 or K >= 8192 or (M <= 32 and K <= 4096 and N >= 2048)), (M, N, K, tile)
    try:
        for force in ((-1, 0), (60, 1), (60, 2), (60, 5), (47, 1)) + (((61, 1), (61, 3)) if M <= 32 else ()):   # the plan, forced slice counts, the tiled kernel, the in-workgroup split
            o.gemm_debug_force(*force)
            for _ in range(2):                                                 # twice: back-to-back launches over the same workspace
                acc = o.gemm_w4a8_i32(at, img, 4, N)
                np.testing.assert_array_equal(acc.cpu().numpy(), acc_ref, err_msg=f"force {force}")
                y = o.gemm_w4a8(at, img, 4, N, 0.02, swt, s_x1=0.007, row_sel=selt, bias=bt)
                np.testing.assert_array_equal(y.float().cpu().numpy(), y_ref, err_msg=f"force {force}")
    finally:
        o.gemm_debug_force(-1, 0)


@pytest.mark.parametrize("out_dtype", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", [(1, 3584, 3584), (5, 1000, 2048), (40, 264, 704)])
def test_epilogue_terms(out_dtype, M, N, K):
    """Per-token scales, the split column's rank-1 term, ragged N (no 16-byte stores), K_pad > K, three output types."""
    from mquant_amd import ops as o
    rng, a, w, s_w, bias = _case(M, N, K, 3 * M + N + K)
    K_pad = (K + 127) // 128 * 128
    ap = np.zeros((M, K_pad), np.int8)
    ap[:, :K] = a
    wp = np.zeros((N, K_pad), np.int8)
    wp[:, :K] = w
    rows = rng.uniform(0.01, 0.05, size=M).astype(np.float32)
    x0 = rng.normal(size=M).astype(np.float32)
    w0 = rng.normal(size=N).astype(np.float32)
    acc_ref = oracle.gemm_i32(a, w)
    y_ref = oracle.round_to(oracle.epilogue(acc_ref, rows, s_w, bias=bias, x0=x0, w0=w0), MODE[out_dtype])
    at = o.TiledAct.from_rows(to_dev(ap))
    img = o.prepack(to_dev(wp), 4)
    o.splitk_workspace(torch.device(DEV), 64 << 20)
    y = o.gemm_w4a8_rowscale(at, img, 4, N, to_dev(rows), to_dev(s_w), bias=to_dev(bias), x0=to_dev(x0), w0=to_dev(w0), out_dtype=out_dtype)
    np.testing.assert_array_equal(y.float().cpu().numpy(), y_ref)


@pytest.mark.parametrize("shape", ["qkv", "down"])
def test_a_rows_result_does_not_depend_on_the_batch_it_came_in(shape):
    """Static quantization is row-local, so a row's output must be the same bits whether it arrives alone (generation: the weight-streaming
    GEMM, a Hadamard row shared by eight workgroups), in a short batch, or inside a 300-row prefill (tiled GEMMs, one workgroup per
    Hadamard row, the XCD row map) -- the Qwen2-VL-7B q|k|v shape and down_proj with its padded online Hadamard (MSQ scales)."""
    from fake_quant import hadamard_utils as hu
    from mquant_amd import ops as o
    from mquant_amd.engine import HadamardSpec, W4A8Linear
    dev = torch.device(DEV)
    o.splitk_workspace(dev, 64 << 20)
    gen = torch.Generator(device=DEV).manual_seed(11)
    if shape == "qkv":
        N, K, k_in, had = 4608, 3584, 3584, None
    else:
        N, K, k_in = 3584, 19968, 18944
        _, Kh = hu.get_hadK(K)
        had = HadamardSpec(K, Kh, hu.had_sign_bits(Kh, dev), False)
    w = torch.randn((N, K), generator=gen, device=DEV) * 0.02
    lin = W4A8Linear.from_float(w, 4, 0.05, 0.02, bias=torch.randn((N,), generator=gen, device=DEV), had=had, in_features=k_in)
    x = torch.randn((300, k_in), generator=gen, device=DEV, dtype=torch.float16)
    sel = (torch.arange(300, device=DEV) % 3 == 0).to(torch.uint8)
    full = lin.forward(x, sel)
    for rows in ([7], [0, 1, 2, 3, 4], list(range(16, 40)), list(range(100, 164))):
        idx = torch.tensor(rows, device=DEV)
        part = lin.forward(x.index_select(0, idx).contiguous(), sel.index_select(0, idx).contiguous())
        assert torch.equal(part, full.index_select(0, idx)), (shape, rows[:2], len(rows))
