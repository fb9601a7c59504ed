"""End to end on REAL HF module classes (the installed transformers' Qwen2-VL, tiny random config): the reference driver's
sequence fuse -> rotate -> qwen2vl_add_act_qaunt -> RTN -> calibration -> model_quant -> forward
(exam/quant_qwen2vl.py:29-222; passes: fake_quant/qwen2vl_rotation.py:232-332, quant_utils.py:543-720).

The reference pins transformers 4.46.3 / 4.47.1; releases from 4.52 on nest the model differently, which
``fake_quant.hf_compat.legacy_qwen2vl`` bridges (a model in the old layout passes through unchanged)."""
import types

import pytest
import torch

transformers = pytest.importorskip("transformers")

import hf_tiny  # noqa: E402


@pytest.fixture(scope="module")
def prepared():
    from fake_quant import hf_compat
    torch.set_grad_enabled(False)
    hf = hf_tiny.build()
    inp = hf_tiny.inputs()
    ref = hf(**inp).logits
    legacy = hf_compat.legacy_qwen2vl(hf)
    vlm = types.SimpleNamespace(model=legacy)          # what VLMEvalKit's wrapper exposes: .model = the HF module
    args = hf_tiny.driver_args()
    return hf, legacy, vlm, args, inp, ref


def test_layout_view_matches_what_the_reference_walks(prepared):
    from fake_quant import hf_compat
    hf, legacy, *_ = prepared
    if hf_compat.is_new_qwen2vl_layout(hf):
        assert legacy.visual is hf.model.visual and legacy.model is hf.model.language_model and legacy.lm_head is hf.lm_head
        assert legacy.config.hidden_size == hf.config.text_config.hidden_size
    else:                                              # the layout the reference pins: no shell at all
        assert legacy is hf
    for path in ("visual.patch_embed.proj", "visual.blocks", "visual.merger.ln_q", "visual.merger.mlp", "model.layers",
                 "model.embed_tokens", "model.norm", "lm_head"):
        obj = legacy
        for part in path.split("."):
            obj = getattr(obj, part)
    assert legacy.visual.blocks[0].attn.num_heads == 2 and hf_compat.legacy_qwen2vl(legacy) is legacy


def test_fuse_rotate_wrap_quantize_on_real_hf_classes(prepared):
    from fake_quant import quant_utils
    hf, legacy, vlm, args, inp, ref = prepared
    ql, qv = hf_tiny.rotate_and_wrap(vlm, args)
    # the structure factors of the 7B model occur: pad 592 -> 624 with K = 156, K = 40 with the split column
    assert legacy.config.need_pad and legacy.config.intermediate_size == 624
    assert [w.K for n, w in ql.items() if "down_proj" in n] == [156, 156] and [w.pad_to for n, w in ql.items() if "down_proj" in n] == [624, 624]
    assert [w.K for n, w in qv.items() if "fc2" in n] == [40, 40] and all(w.split for n, w in qv.items() if "fc2" in n)
    assert len(ql) == 14 and len(qv) == 11 and isinstance(legacy.visual.patch_embed.proj, quant_utils.ActQuantWrapper)
    assert not isinstance(legacy.lm_head, quant_utils.ActQuantWrapper)       # the head is not wrapped (quant_utils.py:560-564)
    # network-function invariance: LN fusion + rotation + online Hadamards leave the logits alone
    for w in list(ql.values()) + list(qv.values()):
        w.simulate_on_cpu, w.real_quant = True, False
    rot = hf(**inp).logits
    assert float((rot - ref).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max()))
    quantizers = hf_tiny.quantize_and_calibrate(vlm, hf, args, ql, qv, [inp, hf_tiny.inputs(seed=5), inp])
    # keys as upstream (gptq/qwen2vl_gptq_plus.py:406,528)
    assert "model.visual.patch_embed.proj.module" in quantizers and "model.visual.blocks.1.mlp.fc2.L2" in quantizers
    assert "model.visual.merger.mlp.2.module" in quantizers and "model.model.layers.1.mlp.down_proj.module" in quantizers
    assert len(quantizers) == 25 + 2 - 2 + 2            # 14 + 11 wrappers: one quantizer each (fc2: L2 only)
    q = hf(**inp).logits
    rel = float((q - ref).norm() / ref.norm())
    assert 1e-3 < rel < 0.5, rel                        # W4A8 on a random 2-layer model: visibly quantized, not broken
    # every wrapper is calibrated and switched on
    for w in list(ql.values()) + list(qv.values()):
        assert w.quantizer.static and w.quantizer.quant and not w.quantizer.calibrate and w.quantizer.quantizer.scale is not None
