"""the rotation / LayerNorm-fusion passes with the weights on the GPU.

mq_rotate_f64 (sign flip + fast Hadamard per weight row, fp64) stands in for the reference's dense
fp64 ``W @ Q``; the passes must land on the weights the reference's own passes produce
(tests/golden/rotation_*.npz, the <= 1e-6 bar of tests/test_rotation_cpu.py)."""
import functools
import os

import numpy as np
import pytest
import torch

import toy_models
from test_rotation_cpu import _run_passes

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
torch.set_grad_enabled(False)


@functools.lru_cache(maxsize=None)
def _q(n, seed=0):
    from fake_quant import rotation_utils as ru
    torch.manual_seed(seed)
    Q = ru.get_orthogonal_matrix(n, "hadamard", device=DEV)
    assert getattr(Q, "_mq_signs", None) is not None and Q.dtype == torch.float64
    return Q


@pytest.mark.parametrize("n", [64, 40, 80, 1280, 3584, 5120, 17920, 20480])
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32, torch.float16, torch.bfloat16])
def test_kernel_equals_dense_fp64_product(n, dtype):
    from fake_quant import rotation_utils as ru
    Q = _q(n, seed=n)
    g = torch.Generator(device=DEV).manual_seed(n)
    X = (torch.randn(37, n, generator=g, device=DEV, dtype=torch.float64) * 0.05).to(dtype)
    dense = X.double() @ Q
    got = ru.mul_q(X, Q)
    assert got.dtype == dtype and got.shape == X.shape and got.data_ptr() != X.data_ptr()
    if dtype == torch.float64:
        torch.testing.assert_close(got, dense, rtol=0, atol=1e-15 * n)
        return
    want = dense.to(dtype)
    # the two fp64 evaluations differ by ~1e-16 relative: the cast can only differ on a rounding tie
    differ = float((got != want).double().mean())
    assert differ < 1e-4, differ
    ulp = {torch.float32: 2.0 ** -23, torch.float16: 2.0 ** -10, torch.bfloat16: 2.0 ** -7}[dtype]
    assert float(((got.double() - want.double()).abs() / want.double().abs().clamp_min(1e-30)).max()) <= 1.01 * ulp


def test_transposed_grouped_and_vector_forms():
    from fake_quant import rotation_utils as ru
    n = 1280
    Q = _q(n, seed=5)
    dense = Q.clone()                                    # same values, no structure tag: the torch path
    g = torch.Generator(device=DEV).manual_seed(1)
    W = torch.randn(n, 96, generator=g, device=DEV, dtype=torch.float64)
    torch.testing.assert_close(ru.mul_qt(Q, W), ru.mul_qt(dense, W), rtol=0, atol=1e-12)
    b = torch.randn(n, generator=g, device=DEV, dtype=torch.float64)
    torch.testing.assert_close(ru.mul_qt(Q, b), ru.mul_qt(dense, b), rtol=0, atol=1e-12)
    G = torch.randn(24, 4 * n, generator=g, device=DEV, dtype=torch.float64)
    lin_a, lin_b = torch.nn.Linear(4 * n, 24, bias=False), torch.nn.Linear(4 * n, 24, bias=False)
    lin_a.weight.data, lin_b.weight.data = G.clone(), G.clone()
    ru.rotate_grouped_input_(lin_a, Q)
    ru.rotate_grouped_input_(lin_b, dense)
    torch.testing.assert_close(lin_a.weight.data, lin_b.weight.data, rtol=0, atol=1e-12)
    # a bias kept in fp32 beside fp16 weights: rotated in fp64, cast to the weight dtype
    bias32 = torch.randn(n, generator=g, device=DEV)
    got = ru.mul_qt(Q, bias32, torch.float16)
    assert got.dtype == torch.float16
    assert float((got.double() - (dense.T @ bias32.double())).abs().max()) < 2e-3
    # weights that live on the host are rotated on the GPU and come back
    Wh = W.cpu()
    out = ru.mul_qt(Q, Wh)
    assert out.device.type == "cpu"
    torch.testing.assert_close(out, ru.mul_qt(dense, W).cpu(), rtol=0, atol=1e-12)


def test_rows_with_a_stride_and_in_place_use():
    from fake_quant import hadamard_utils as hu
    from mquant_amd import ops
    n = 160
    K = hu.get_hadK(n)[1]
    Q = _q(n, seed=9)
    buf = torch.randn(11, 256, device=DEV, dtype=torch.float64)
    keep = buf.clone()
    view = buf[:, :n]
    want = view @ Q
    ops.rotate_f64_(view, Q._mq_signs.to(DEV), K, hu.had_sign_bits(K, torch.device(DEV), prepared=False))
    torch.testing.assert_close(buf[:, :n], want, rtol=0, atol=1e-13)
    assert torch.equal(buf[:, n:], keep[:, n:])


def test_bad_sizes_are_refused():
    from mquant_amd import ops
    from mquant_amd._lib import MQuantHipError
    with pytest.raises(MQuantHipError):
        ops.rotate_f64_(torch.zeros(2, 24, device=DEV, dtype=torch.float64), None, 5, None)     # 24 % 5
    with pytest.raises(MQuantHipError):
        ops.rotate_f64_(torch.zeros(2, 24, device=DEV, dtype=torch.float64), None, 1, None)     # 24 not 2^m
    with pytest.raises(MQuantHipError):
        ops.rotate_f64_(torch.zeros(1, 32768, device=DEV, dtype=torch.float64), None, 1, None)  # 256 KiB row
    with pytest.raises(MQuantHipError):
        ops.rotate_f64_(torch.zeros(2, 64, dtype=torch.float64), None, 1, None)                 # CPU tensor


@pytest.mark.parametrize("kind", ["qwen2vl", "internvl", "qwenvl", "minicpmv"])
@pytest.mark.parametrize("where", ["cuda", "cpu"])
def test_passes_on_the_gpu_reproduce_the_reference_weights(golden_dir, kind, where, monkeypatch):
    """Weights on the GPU (or on the host with a GPU present: rotated on the GPU, returned)."""
    from mquant_amd import ops
    g = np.load(os.path.join(golden_dir, f"rotation_{kind}.npz"))
    model, pixels, ids = toy_models.build(kind, seed=int(g["seed"]))
    model = model.to(where)
    launches = []
    real = ops.rotate_f64_
    monkeypatch.setattr(ops, "rotate_f64_", lambda *a, **k: (launches.append(a[0].shape), real(*a, **k))[1])
    _run_passes(kind, model, toy_models.rotation_args(), seed=int(g["rot_seed"]), as_upstream=True)
    assert len(launches) >= 10, launches                 # the dense product is not what ran
    sd = model.state_dict()
    keys = [k for k in g.files if k not in ("seed", "rot_seed", "logits")]
    assert sorted(keys) == sorted(sd.keys())
    for k in keys:
        assert sd[k].device.type == where or sd[k].numel() == 1, k      # RMSN's placeholder weight stays where it is made
        np.testing.assert_allclose(sd[k].cpu().numpy(), g[k], rtol=0, atol=1e-6, err_msg=k)
    np.testing.assert_allclose(model(pixels.to(where), ids.to(where)).cpu().numpy(), g["logits"], rtol=0, atol=2e-5)


def test_random_mode_keeps_the_dense_path(monkeypatch):
    from fake_quant import rotation_utils as ru
    from mquant_amd import ops
    monkeypatch.setattr(ops, "rotate_f64_", lambda *a, **k: pytest.fail("structured kernel on an unstructured Q"))
    torch.manual_seed(0)
    Q = ru.get_orthogonal_matrix(48, "random", device=DEV)
    X = torch.randn(5, 48, device=DEV, dtype=torch.float64)
    torch.testing.assert_close(ru.mul_q(X, Q), X @ Q, rtol=0, atol=0)
