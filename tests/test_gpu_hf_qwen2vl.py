"""End to end on REAL HF module classes, on the GPU: the tiny random-config Qwen2-VL of tests/hf_tiny.py taken through the
reference driver's sequence (exam/quant_qwen2vl.py:29-222) in fp16, then the quantized forward on the INTEGER backend (every
wrapped Linear and the patch-embedding Conv3d on the HIP kernels, q|k|v and gate|up fused by model_quant) against the reference's
simulated evaluation of the same calibrated model (quant_utils.py:330-391)."""
import types

import pytest
import torch

pytestmark = pytest.mark.gpu
transformers = pytest.importorskip("transformers")

import hf_tiny  # noqa: E402


def test_integer_backend_equals_the_simulated_forward_on_real_hf_classes():
    from fake_quant import hf_compat, quant_utils
    torch.set_grad_enabled(False)
    dev = "cuda:0"
    hf = hf_tiny.build(dtype=torch.float16, device=dev)
    inp = hf_tiny.inputs(device=dev, dtype=torch.float16)
    ref = hf(**inp).logits.float()
    legacy = hf_compat.legacy_qwen2vl(hf)
    vlm = types.SimpleNamespace(model=legacy)
    args = hf_tiny.driver_args()
    ql, qv = hf_tiny.rotate_and_wrap(vlm, args)
    wrappers = {**{"llm." + k: v for k, v in ql.items()}, **{"visual." + k: v for k, v in qv.items()}}
    rot = hf(**inp).logits.float()
    assert float((rot - ref).norm() / ref.norm()) < 2e-2           # fp16 weights after fp64 surgery: invariance at half precision
    hf_tiny.quantize_and_calibrate(vlm, hf, args, ql, qv, [inp, hf_tiny.inputs(device=dev, dtype=torch.float16, seed=5), inp])
    # integer backend
    with quant_utils.token_type_mask(None):
        y_int = hf(**inp).logits.float()
    backends = {n: w.backend() for n, w in wrappers.items()}
    assert all("integer" in b for b in backends.values()), {n: b for n, b in backends.items() if "integer" not in b}
    fused = {id(w.__dict__["_group"]) for w in wrappers.values() if w.__dict__.get("_group") is not None and w.__dict__["_group"].enabled}
    assert len(fused) == 4, "q|k|v and gate|up of both decoder layers run as one quantize + one GEMM each"
    # the reference's simulated evaluation of the same model
    for w in wrappers.values():
        w.real_quant = False
    y_sim = hf(**inp).logits.float()
    assert all("simulated" in w.backend() for w in wrappers.values())
    for w in wrappers.values():
        w.real_quant = True
    assert torch.isfinite(y_int).all() and torch.isfinite(y_sim).all()
    quant_err = float((y_sim - ref).norm() / ref.norm())
    diff = float((y_int - y_sim).norm() / y_sim.norm())
    # the two evaluations share every scale and level grid; they differ by the fp16 GEMM of the simulated path against exact
    # integer accumulation (<= 1e-3 per Linear, north_star) and the few activation levels that flips downstream
    assert diff < 0.05 and diff < 0.5 * quant_err, (diff, quant_err)
    assert torch.equal(hf(**inp).logits.float(), y_int)            # back on the integer backend: deterministic
