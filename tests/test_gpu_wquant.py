"""On-device weight quantizer (mq_wquant_sym) against the reference goldens and the
oracle: scales, levels, int4 wire format and W~ bit-exact, on fp32 / fp16 / bf16 weights, with and
without the MSE clip search; ragged shapes and strides; the engine built straight from it."""
import os

import numpy as np
import pytest
import torch

import oracle
from golden_inputs import make_w, make_x
from test_oracle_golden import WQUANT_CASES, wquant_case

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
DT = {0: torch.float32, 1: torch.float16, 2: torch.bfloat16}


@pytest.mark.parametrize("case", WQUANT_CASES)
def test_matches_reference_goldens(golden_dir, case):
    from mquant_amd import ops
    g, w, mode, bits, mse = wquant_case(golden_dir, case)
    wt = torch.from_numpy(w).to(device=DEV, dtype=DT[mode])
    scale, levels, packed, wq = ops.wquant_sym(wt, bits, mse, want_packed=(bits == 4), want_wq=True)
    np.testing.assert_array_equal(scale.cpu().numpy(), g["scale"])
    np.testing.assert_array_equal(wq.float().cpu().numpy(), g["wq"])
    o_scale, o_levels = oracle.wquant_sym(w, bits=bits, mse=mse)
    np.testing.assert_array_equal(levels.cpu().numpy(), o_levels)
    if bits == 4:
        np.testing.assert_array_equal(packed.cpu().numpy(), oracle.pack_i4(o_levels))
        np.testing.assert_array_equal(ops.unpack_i4(packed).cpu().numpy(), o_levels)


@pytest.mark.parametrize("N,K,mse", [(1, 2, True), (7, 130, True), (130, 1000, True), (513, 3584, False),
                                      (64, 19968, True), (5, 33, False)])
def test_matches_oracle_on_ragged_shapes(N, K, mse):
    from mquant_amd import ops
    w = make_w(N * 7 + K, (N, K))
    wt = torch.from_numpy(w).to(DEV).half()
    w16 = wt.float().cpu().numpy()
    scale, levels, _, wq = ops.wquant_sym(wt, 4, mse, want_wq=True)
    o_scale, o_levels = oracle.wquant_sym(w16, bits=4, mse=mse)
    np.testing.assert_array_equal(scale.cpu().numpy(), o_scale)
    np.testing.assert_array_equal(levels.cpu().numpy(), o_levels)
    np.testing.assert_array_equal(wq.float().cpu().numpy(), oracle.round_to(o_scale[:, None] * o_levels.astype(np.float32), 1))


def test_row_stride_grid_and_norm_parameters():
    from mquant_amd import ops
    big = torch.from_numpy(make_w(9, (24, 640))).to(DEV)
    view = big[:, 64:576]                                     # ldw = 640, K = 512
    scale, levels, _, _ = ops.wquant_sym(view, 8, True, norm=2.0, grid=50, maxshrink=0.5)
    o_scale, o_levels = oracle.wquant_sym(view.cpu().numpy(), bits=8, mse=True, norm=2.0, grid=50, maxshrink=0.5)
    np.testing.assert_array_equal(scale.cpu().numpy(), o_scale)
    np.testing.assert_array_equal(levels.cpu().numpy(), o_levels)


def test_errors_are_loud():
    from mquant_amd import ops
    from mquant_amd._lib import MQuantHipError
    w = torch.zeros((4, 7), device=DEV)
    with pytest.raises(MQuantHipError):
        ops.wquant_sym(w, 4, want_packed=True)               # odd K has no int4 wire format
    with pytest.raises(MQuantHipError):
        ops.wquant_sym(w, 9)
    with pytest.raises(MQuantHipError):
        ops.wquant_sym(torch.zeros((4, 8)), 4)                # CPU tensor


def test_weight_quantizer_class_uses_the_kernel_and_matches_torch_path():
    """fake_quant.WeightQuantizer on CUDA tensors == the torch restatement it falls back to for the
    asymmetric / per-tensor modes, scale and W~ bit for bit."""
    from fake_quant import quant_utils as qu
    w = torch.from_numpy(make_w(77, (96, 1024))).to(DEV).half()
    for mse in (False, True):
        a, b = qu.WeightQuantizer(), qu.WeightQuantizer()
        a.configure(4, perchannel=True, sym=True, mse=mse)
        b.configure(4, perchannel=True, sym=True, mse=mse)
        a.find_params(w)
        b.use_kernel = False
        b.find_params(w)
        assert a.scale.shape == b.scale.shape == (96, 1) and a.scale.dtype == b.scale.dtype == torch.float32
        # the torch path sums errors in torch's own order: identical scales except for near-ties
        same = (a.scale == b.scale).float().mean().item()
        assert same >= (0.97 if mse else 1.0)
        if not mse:
            torch.testing.assert_close(a.quantize(w), b.quantize(w), rtol=0, atol=0)


def test_engine_from_float_equals_two_step_construction():
    from mquant_amd import ops
    from mquant_amd.engine import W4A8Linear
    w = torch.from_numpy(make_w(5, (160, 512))).to(DEV).half()
    x = torch.from_numpy(make_x(6, (48, 512))).to(DEV).half()
    lin = W4A8Linear.from_float(w, 4, s_x0=0.05)
    scale, levels, _, _ = ops.wquant_sym(w, 4)
    ref = W4A8Linear(levels, scale, 4, None, 0.05)
    torch.testing.assert_close(lin(x), ref(x), rtol=0, atol=0)
    acc = oracle.gemm_i32(oracle.quant_static(x.float().cpu().numpy(), np.float32(0.05)), levels.cpu().numpy())
    want = oracle.round_to(oracle.epilogue(acc, np.float32(0.05), scale.cpu().numpy()), 1)
    np.testing.assert_array_equal(lin(x).float().cpu().numpy(), want)
