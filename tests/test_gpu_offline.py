"""apply_exact_had_to_linear on the GPU (the fp32 mq_hadamard kernel under
matmul_hadU_cuda) against the reference's outputs in tests/golden/offline_hadamard.npz -- the same
cases and the same tolerance as the CPU test (test_fake_quant_cpu.py::test_offline_hadamard_on_linear)."""
import os

import numpy as np
import pytest
import torch

from golden_inputs import make_w

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ATOL = 2e-7      # fp32 weights of std 0.02: a few ulp of the largest rotated element
torch.set_grad_enabled(False)


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(os.path.join(golden_dir, "offline_hadamard.npz"))


def _lin(n_in, n_out, w_seed, b_seed=None):
    lin = torch.nn.Linear(n_in, n_out, bias=b_seed is not None)
    lin.weight.data = torch.from_numpy(make_w(w_seed, (n_out, n_in)))
    if b_seed is not None:
        lin.bias.data = torch.from_numpy(make_w(b_seed, (n_out,), std=0.1))
    return lin.to(DEV)


@pytest.mark.parametrize("n", [5120, 1280])
def test_full_hadamard_over_inputs(g, n):
    from fake_quant import hadamard_utils as hu
    lin = _lin(n, 24, 700 + n)
    hu.apply_exact_had_to_linear(lin, had_dim=-1, output=False)
    assert lin.weight.is_cuda and lin.weight.dtype == torch.float32
    np.testing.assert_allclose(lin.weight.data.cpu().numpy(), g[f"W_in_{n}"], rtol=0, atol=ATOL)


@pytest.mark.parametrize("n", [5120, 1280])
def test_full_hadamard_over_outputs_with_bias(g, n):
    from fake_quant import hadamard_utils as hu
    lin = _lin(24, n, 710 + n, 711 + n)
    hu.apply_exact_had_to_linear(lin, had_dim=-1, output=True)
    np.testing.assert_allclose(lin.weight.data.cpu().numpy(), g[f"W_out_{n}"], rtol=0, atol=ATOL)
    np.testing.assert_allclose(lin.bias.data.cpu().numpy(), g[f"b_out_{n}"], rtol=0, atol=ATOL)


def test_per_head_hadamard(g):
    from fake_quant import hadamard_utils as hu
    lin = _lin(512, 24, 720)
    hu.apply_exact_had_to_linear(lin, had_dim=128, output=False)
    np.testing.assert_allclose(lin.weight.data.cpu().numpy(), g["W_headin_128"], rtol=0, atol=ATOL)
    lin = _lin(24, 512, 721, 722)
    hu.apply_exact_had_to_linear(lin, had_dim=128, output=True)
    np.testing.assert_allclose(lin.weight.data.cpu().numpy(), g["W_headout_128"], rtol=0, atol=ATOL)
    np.testing.assert_allclose(lin.bias.data.cpu().numpy(), g["b_headout_128"], rtol=0, atol=ATOL)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_half_weights_are_rotated_in_fp32_and_cast_back(g, dtype):
    """Reference :135-191 casts to fp32, rotates, casts back: one rounding of the fp32 result."""
    from fake_quant import hadamard_utils as hu
    n = 1280
    lin = _lin(n, 24, 700 + n).to(dtype)
    w32 = lin.weight.data.float().clone()
    hu.apply_exact_had_to_linear(lin, had_dim=-1, output=False)
    assert lin.weight.dtype == dtype
    ref = torch.nn.Linear(n, 24, bias=False).to(DEV)
    ref.weight.data = w32
    hu.apply_exact_had_to_linear(ref, had_dim=-1, output=False)
    assert torch.equal(lin.weight.data, ref.weight.data.to(dtype))


def test_the_hip_kernel_is_what_ran(monkeypatch):
    """No silent torch path on a GPU box: the rotation must go through matmul_hadU_cuda."""
    from fake_quant import hadamard_utils as hu
    seen = []
    real = hu.matmul_hadU_cuda
    monkeypatch.setattr(hu, "matmul_hadU_cuda", lambda *a, **k: (seen.append(1), real(*a, **k))[1])
    hu.apply_exact_had_to_linear(_lin(1280, 8, 3), had_dim=-1, output=False)
    hu.apply_exact_had_to_linear(_lin(512, 8, 4), had_dim=128, output=False)
    assert len(seen) == 2
