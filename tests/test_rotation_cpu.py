"""LayerNorm fusion + rotation passes on toy models with the HF attribute layout.

1. network function unchanged by fuse + rotate (the property the passes exist for), incl. the
   online Hadamard in front of fc2 / down_proj and the padded down_proj;
2. weights after the passes equal those produced by the reference's own functions on the same
   toy model and RNG seed (tests/golden/rotation_*.npz, written by tools/gen_golden.py).
"""
import os
import types

import numpy as np
import pytest
import torch

import toy_models
from fake_quant import hadamard_utils as hu
from fake_quant import internvl_rotation, minicpmv_rotation, module_util, qwen2vl_rotation, rotation_utils

torch.set_grad_enabled(False)

# Upstream's ViT fusion is an approximation for two of the models; kept as is (drop-in), the
# rotation that follows is exact and is checked against the FUSED model for these:
#  * InternVL2: mlp1[0] is a LayerNorm over FOUR concatenated tokens, whose joint mean is not the
#    per-token mean the ViT fusion removes (reference internvl_rotation.py:197-206);
#  * Qwen-VL: the ViT output enters attn_pool.kv_proj without a norm in between, so removing the
#    per-token mean from the residual stream changes kv_proj's input (reference
#    rotation_utils.py:148-149,190 and model/visual_opt.py:513-517).
APPROXIMATE_FUSION = ("internvl", "qwenvl")


def _run_passes(kind, model, args, seed=123, probe=None, as_upstream=False):
    """fuse, (probe the fused model), rotate.  Returns the probe's logits after fusion only."""
    torch.manual_seed(seed)
    wrapper = types.SimpleNamespace(model=model)
    fuse, rotate = {"qwen2vl": (qwen2vl_rotation.fuse_qwen2vl_layer_norms, qwen2vl_rotation.rotate_qwen2vl_model),
                    "internvl": (internvl_rotation.fuse_internvl_layer_norms, internvl_rotation.rotate_internvl2_model),
                    "qwenvl": (rotation_utils.fuse_qwenvl_layer_norms, rotation_utils.rotate_model),
                    "minicpmv": (minicpmv_rotation.fuse_minicpmv_layer_norms, minicpmv_rotation.rotate_minicpmv_model)}[kind]
    fuse(model if kind in ("qwenvl", "minicpmv") else wrapper, args)       # these passes take the HF module itself
    if kind == "minicpmv" and not args.no_fuse_visual_clip and not as_upstream:
        # upstream folds vpm.post_layernorm's affine into resampler.kv_proj but leaves the module a
        # LayerNorm (minicpmv_rotation.py:54-60 replaces the encoder's norms only); its mean
        # subtraction is not rotation invariant.  With the equivalent RMSN (the stream is zero-mean
        # at that point) the rotation is exact, which is what this test pins down.
        model.vpm.post_layernorm = module_util.RMSN(model.vpm.embed_dim, eps=1e-6)
    fused = model(*probe) if probe is not None else None
    rotate(model, args)
    model.online_visual = bool(args.rotate_visual_clip and args.online_visual_hadamard)
    model.online_llm = bool(args.rotate_llm and args.online_llm_hadamard)
    return fused


@pytest.mark.parametrize("kind", ["qwen2vl", "internvl", "qwenvl", "minicpmv"])
@pytest.mark.parametrize("mode", ["hadamard", "random"])
def test_network_function_is_invariant(kind, mode):
    model, pixels, ids = toy_models.build(kind, seed=7)
    want = model(pixels, ids)
    fused = _run_passes(kind, model, toy_models.rotation_args(rotate_mode=mode), probe=(pixels, ids))
    got = model(pixels, ids)
    if kind in APPROXIMATE_FUSION:
        assert (fused - want).abs().max() > 1e-3
        want = fused
    else:
        torch.testing.assert_close(fused, want, rtol=0, atol=1e-9)
    # the exact-Hadamard steps run in fp32 (as upstream), hence 1e-5 and not 1e-12
    torch.testing.assert_close(got, want, rtol=0, atol=2e-5)
    left = [n for n, m in model.named_modules() if isinstance(m, torch.nn.LayerNorm)]
    assert left == (["transformer.visual.ln_pre"] if kind == "qwenvl" else [])      # ln_pre sits before fc_sub_mean
    assert any(isinstance(m, module_util.RMSN) for m in model.modules())


@pytest.mark.parametrize("kind", ["qwen2vl", "internvl", "qwenvl", "minicpmv"])
def test_partial_passes_are_invariant_too(kind):
    for over in (dict(rotate_visual_clip=False, rotate_visual_cross_attn=False, no_fuse_visual_clip=True,
                      no_fuse_visual_cross_attn=True),
                 dict(rotate_llm=False),
                 dict(online_visual_hadamard=False, online_llm_hadamard=False)):
        model, pixels, ids = toy_models.build(kind, seed=11)
        want = model(pixels, ids)
        fused = _run_passes(kind, model, toy_models.rotation_args(**over), probe=(pixels, ids))
        if kind in APPROXIMATE_FUSION and not over.get("no_fuse_visual_clip"):
            want = fused                               # see test_network_function_is_invariant
        torch.testing.assert_close(model(pixels, ids), want, rtol=0, atol=2e-5)


def test_qwen2vl_down_proj_is_padded_with_zeros():
    model, pixels, ids = toy_models.build("qwen2vl", seed=3, inter=88)      # 88 -> 96 = 12 * 8
    want = model(pixels, ids)
    _run_passes("qwen2vl", model, toy_models.rotation_args())
    assert model.config.need_pad and model.config.intermediate_size == 96 == hu.auto_pad_size(88)
    for layer in model.model.layers:
        assert layer.mlp.down_proj.in_features == 96 and layer.mlp.down_proj.bias is None
    torch.testing.assert_close(model(pixels, ids), want, rtol=0, atol=2e-5)


def test_fuse_ln_linear_matches_definition():
    torch.manual_seed(0)
    ln = torch.nn.LayerNorm(16).double()
    ln.weight.data, ln.bias.data = torch.randn(16).double(), torch.randn(16).double()
    a, b = torch.nn.Linear(16, 8, bias=False).double(), torch.nn.Linear(16, 4).double()
    x = torch.randn(5, 16).double()
    want = [a(ln(x)), b(ln(x))]
    rotation_utils.fuse_ln_linear(ln, [a, b])
    assert a.bias is not None and torch.all(ln.weight == 1) and torch.all(ln.bias == 0)
    torch.testing.assert_close([a(ln(x)), b(ln(x))], want, rtol=0, atol=1e-12)


def test_bake_mean_then_rmsn_equals_layernorm():
    torch.manual_seed(0)
    lin, ln = torch.nn.Linear(12, 16).double(), torch.nn.LayerNorm(16, elementwise_affine=False, eps=1e-6).double()
    x = torch.randn(7, 12).double()
    want = ln(lin(x))
    rotation_utils.bake_mean_into_linear(lin)
    torch.testing.assert_close(module_util.RMSN(16, eps=1e-6)(lin(x)), want, rtol=0, atol=1e-12)


def test_orthogonal_matrices():
    torch.manual_seed(0)
    for mode, n in (("random", 24), ("hadamard", 40), ("hadamard", 64)):
        Q = rotation_utils.get_orthogonal_matrix(n, mode, device="cpu")
        assert Q.dtype == torch.float64
        torch.testing.assert_close(Q @ Q.T, torch.eye(n, dtype=torch.float64), rtol=0, atol=1e-6)
    with pytest.raises(ValueError):
        rotation_utils.get_orthogonal_matrix(8, "dct")


@pytest.mark.parametrize("kind", ["qwen2vl", "internvl", "qwenvl", "minicpmv"])
def test_weights_match_reference_passes(golden_dir, kind):
    g = np.load(os.path.join(golden_dir, f"rotation_{kind}.npz"))
    model, pixels, ids = toy_models.build(kind, seed=int(g["seed"]))
    _run_passes(kind, model, toy_models.rotation_args(), seed=int(g["rot_seed"]), as_upstream=True)
    sd = model.state_dict()
    keys = [k for k in g.files if k not in ("seed", "rot_seed", "logits")]
    assert sorted(keys) == sorted(sd.keys())
    for k in keys:
        np.testing.assert_allclose(sd[k].numpy(), g[k], rtol=0, atol=1e-6, err_msg=k)
    np.testing.assert_allclose(model(pixels, ids).numpy(), g["logits"], rtol=0, atol=2e-5)
