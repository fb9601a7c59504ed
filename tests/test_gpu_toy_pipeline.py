"""The whole MQuant recipe on a toy Qwen2-VL (HF attribute layout) on the GPU, the way
exam/quant_qwen2vl.py strings it together: LayerNorm fusion -> rotation (online Hadamard, padded
down_proj) -> wrap -> online-Hadamard flags / split / pad hook -> RTN weight pass -> static
calibration protocol (MSQ on the LLM) -> quantized forward.  The real W4A8 kernels must reproduce
the simulated (fake-quant, torch) forward of the same wrappers, and both stay near the fp model."""
import functools
import types

import pytest
import torch

import toy_models

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
torch.set_grad_enabled(False)


class Args:
    skip_names = []


class _Dataset:
    def __init__(self, n):
        import pandas as pd
        self.data = pd.DataFrame({"v": list(range(n))})

    def build_prompt(self, record):
        return int(record["v"])


def build(inter, llm_split, visual_split, w_bits, msq, gptq=False):
    from fake_quant import hadamard_utils as hu, quant_utils as qu, qwen2vl_rotation, utils
    from fake_quant.gptq import qwen2vl_gptq_plus
    model, pixels, ids = toy_models.build("qwen2vl", seed=21, inter=inter)
    want = model(pixels, ids)
    rargs = toy_models.rotation_args()
    torch.manual_seed(4)
    vlm = types.SimpleNamespace(model=model)
    qwen2vl_rotation.fuse_qwen2vl_layer_norms(vlm, rargs)
    qwen2vl_rotation.rotate_qwen2vl_model(model, rargs)
    model = model.float().to(DEV)
    vlm.model = model
    # gptq=True: the flags of the documented command lines (docs/qwen2vl.md): GPTQ with --act_order and
    # the MSE clip search on both towers, --visual_split
    qargs = types.SimpleNamespace(quant_llm=True, quant_visual_clip=True, quant_cross_attention=True,
                                  act_per_tensor=False, visual_w_rtn=not gptq, llm_w_rtn=not gptq, visual_w_bits=w_bits,
                                  llm_w_bits=w_bits, w_asym=False, visual_w_clip=gptq, llm_w_clip=gptq,
                                  skip_names=[], nsamples=4, percdamp=0.01, w_groupsize=-1, act_order=gptq,
                                  dataset_name="toy", visual_split=visual_split, llm_split=llm_split)
    pix_dev, ids_dev = pixels.float().to(DEV), ids.to(DEV)

    def generate(message, dataset):
        g = torch.Generator().manual_seed(int(message))
        return model(pix_dev + 0.1 * torch.randn(pix_dev.shape, generator=g).to(DEV), ids_dev)
    vlm.generate = generate
    qu.qwen2vl_add_act_qaunt(vlm, qargs)
    for name, w in qu.find_qlayers(model.model, [qu.ActQuantWrapper]).items():      # exam/quant_qwen2vl.py:107-127
        if "mlp.down_proj" in name:
            w.had_K, w.K = hu.get_hadK(model.config.intermediate_size)
            w.online_full_had = True
            w.split = llm_split
            if llm_split:
                w.split_weights()
            if model.config.need_pad:
                w.register_forward_pre_hook(functools.partial(utils.revise_down_input,
                                                              new_size=model.config.intermediate_size))
    for name, w in qu.find_qlayers(model.visual, [qu.ActQuantWrapper]).items():     # :129-143
        if "mlp.fc2" in name:
            w.had_K, w.K = hu.get_hadK(int(model.visual.blocks[0].mlp.fc2.module.in_features))
            w.online_full_had = True
            w.split = visual_split
            if visual_split:
                w.split_weights()
    quantizers = qwen2vl_gptq_plus.qwen2vl_rtn_gptq_fwrd_plus(vlm, _Dataset(8), DEV, "toy", qargs)
    wrappers = qu.find_qlayers(model, [qu.ActQuantWrapper])
    assert len(quantizers) >= len(wrappers)            # split wrappers contribute ".module" and ".L2"
    for name, w in wrappers.items():
        w.quantizer.configure(bits=8, sym=True, static=True, msq=msq and name.startswith("model."))
    return model, wrappers, pixels.float().to(DEV), ids.to(DEV), want


def calibrate(model, pixels, ids, mask):
    from fake_quant import quant_utils as qu
    g = torch.Generator(device="cpu").manual_seed(0)
    with qu.token_type_mask(mask):
        qu.model_open_calibrate(model, Args())
        for i in range(3):
            if i == 2:
                qu.model_open_last_calibrate(model, Args())
            model(pixels + 0.05 * torch.randn(pixels.shape, generator=g).to(DEV), ids)
        qu.model_close_calibrate(model, Args())
        qu.model_quant(model, Args())


@pytest.mark.parametrize("inter,llm_split,visual_split,w_bits,msq,gptq", [
    (96, False, False, 8, False, False), (88, True, True, 8, True, False), (96, False, True, 4, True, False),
    (96, False, True, 8, False, True), (88, False, True, 4, True, True)])
def test_quantized_toy_model_real_kernels_equal_simulation(inter, llm_split, visual_split, w_bits, msq, gptq):
    from fake_quant import quant_utils as qu
    model, wrappers, pixels, ids, want = build(inter, llm_split, visual_split, w_bits, msq, gptq)
    assert model.config.need_pad == (inter == 88)
    mask = torch.tensor([0, 0, 1, 1, 1, 1, 1], device=DEV)          # 2 merged vision tokens, 5 text tokens
    calibrate(model, pixels, ids, mask)
    with qu.token_type_mask(mask):
        real = model(pixels, ids)
        used = [n for n, w in wrappers.items() if qu.real_engine(w) is not None]
        assert len(used) == len(wrappers), sorted(set(wrappers) - set(used))   # every wrapped layer ran the kernels
        for w in wrappers.values():
            w.real_quant = False
        sim = model(pixels, ids)
    # same integer grids, different evaluation order of the dequantised products
    torch.testing.assert_close(real, sim, rtol=0, atol=2e-3 * float(sim.abs().max()))
    err = float((real.double().cpu() - want).norm() / want.norm())
    assert err < (0.05 if w_bits == 8 else 0.35), err


# ------------------------------------------------------------------ the other model families
FAMILIES = {
    # kind: (fuse, rotate, takes_wrapper_for_fuse, add_actquant, weight pass, LLM out-proj tag, ViT out-proj tag)
    "internvl": ("internvl_rotation.fuse_internvl_layer_norms", "internvl_rotation.rotate_internvl2_model", True,
                 "internvl_add_act_qaunt", "internvl_gptq_plus.internvl_rtn_gptq_fwrd_plus", "feed_forward.w2", "mlp.fc2"),
    "qwenvl": ("rotation_utils.fuse_qwenvl_layer_norms", "rotation_utils.rotate_model", False,
               "qwenvl_add_act_qaunt", "qwenvl_gptq_plus.qwenvl_rtn_gptq_fwrd_plus", "mlp.c_proj", "mlp.c_proj"),
    "minicpmv": ("minicpmv_rotation.fuse_minicpmv_layer_norms", "minicpmv_rotation.rotate_minicpmv_model", False,
                 "minicpmv_add_act_qaunt", "minicpmv_gptq_plus.minicpmv_rtn_gptq_fwrd_plus", "mlp.down_proj", "mlp.fc2"),
}


def _resolve(path):
    import importlib
    mod, fn = path.rsplit(".", 1)
    pkg = "fake_quant.gptq." if mod.endswith("_gptq_plus") else "fake_quant."
    return getattr(importlib.import_module(pkg + mod), fn)


@pytest.mark.parametrize("kind,split", [("internvl", False), ("internvl", True), ("qwenvl", False), ("qwenvl", True),
                                        ("minicpmv", False), ("minicpmv", True)])
def test_other_model_families_run_the_real_kernels(kind, split):
    from fake_quant import hadamard_utils as hu, quant_utils as qu
    fuse, rotate, fuse_takes_wrapper, add_act, weight_pass, llm_tag, vit_tag = FAMILIES[kind]
    model, pixels, ids = toy_models.build(kind, seed=33)
    rargs = toy_models.rotation_args()
    torch.manual_seed(5)
    vlm = types.SimpleNamespace(model=model)
    _resolve(fuse)(vlm if fuse_takes_wrapper else model, rargs)
    if kind == "minicpmv":      # upstream leaves this one a LayerNorm; see tests/test_rotation_cpu.py
        from fake_quant import module_util
        model.vpm.post_layernorm = module_util.RMSN(model.vpm.embed_dim, eps=1e-6)
    want = model(pixels, ids)                                  # fused fp model (fusion is approximate upstream)
    _resolve(rotate)(model, rargs)
    model = model.float().to(DEV)
    vlm.model = model
    qargs = types.SimpleNamespace(quant_llm=True, quant_visual_clip=True, quant_cross_attention=True,
                                  act_per_tensor=False, visual_w_rtn=True, llm_w_rtn=True, visual_w_bits=8,
                                  llm_w_bits=8, w_asym=False, visual_w_clip=False, llm_w_clip=False,
                                  skip_names=[], dataset_name="toy")
    getattr(qu, add_act)(vlm if kind == "internvl" else model, qargs)
    wrappers = qu.find_qlayers(model, [qu.ActQuantWrapper])
    rotated = 0
    for name, w in wrappers.items():
        is_llm = name.startswith(("language_model.", "transformer.h.", "llm."))
        tag = llm_tag if is_llm else vit_tag
        if tag in name and "attn" not in name.split(tag)[0].rsplit(".", 2)[-1]:
            if kind == "qwenvl" and not (("transformer.h" in name) or ("transformer.resblock" in name)):
                continue
            w.had_K, w.K = hu.get_hadK(w.module.in_features)
            w.online_full_had = True
            w.split = split
            if split:
                w.split_weights()
            rotated += 1
    assert rotated == 4                                        # 2 ViT blocks + 2 LLM layers
    from fake_quant import gptq
    entry = getattr(gptq, weight_pass.rsplit(".", 1)[1])      # the name the exam/ drivers dereference
    assert entry is _resolve(weight_pass)
    if kind == "qwenvl":                                       # upstream arities (gptq/qwenvl_gptq_plus.py:620)
        entry(vlm, None, DEV, qargs)
    else:
        entry(vlm, None, DEV, "toy", qargs)
    for name, w in wrappers.items():
        w.quantizer.configure(bits=8, sym=True, static=True)
    pixels, ids = pixels.float().to(DEV), ids.to(DEV)
    qu.model_open_calibrate(model, Args())
    model(pixels, ids)
    qu.model_open_last_calibrate(model, Args())
    model(pixels * 1.01, ids)
    qu.model_close_calibrate(model, Args())
    qu.model_quant(model, Args())
    real = model(pixels, ids)
    missing = [n for n, w in wrappers.items() if qu.real_engine(w) is None]
    assert not missing, missing
    for w in wrappers.values():
        w.real_quant = False
    sim = model(pixels, ids)
    torch.testing.assert_close(real, sim, rtol=0, atol=2e-3 * float(sim.abs().max()))
    assert float((real.double().cpu() - want).norm() / want.norm()) < 0.08


# ------------------------------------------------------------------ sibling fusion (q|k|v, gate|up)
def _groups(wrappers):
    return {id(w.__dict__["_group"]): w.__dict__["_group"] for w in wrappers.values() if w.__dict__.get("_group") is not None}


def test_sibling_fusion_through_model_quant_is_bit_identical_to_per_linear_evaluation():
    """model_quant() groups the q/k/v and gate/up wrappers of every decoder layer (same parent, same input,
    identical static scale sets): one quantize + one GEMM per group, members return views.  The model output
    must not move by a bit against the per-Linear evaluation the reference performs (quant_utils.py:626-662)."""
    from fake_quant import quant_utils as qu
    model, wrappers, pixels, ids, _ = build(96, False, True, 4, True)
    mask = torch.tensor([0, 0, 1, 1, 1, 1, 1], device=DEV)
    calibrate(model, pixels, ids, mask)
    groups = _groups(wrappers)
    names = sorted(tuple(sorted(n for n, _ in g.members)) for g in groups.values())
    assert names == [("gate_proj", "up_proj"), ("gate_proj", "up_proj"), ("k_proj", "q_proj", "v_proj"),
                     ("k_proj", "q_proj", "v_proj")], names
    with qu.token_type_mask(mask):
        fused = model(pixels, ids)
        assert all(g.enabled and g.launches == 1 and g.engine is not None for g in groups.values())
        assert all(g._result is None for g in groups.values())            # every product fully consumed and released
        grouped = [w for w in wrappers.values() if w.__dict__.get("_group") is not None]
        assert grouped and all(w._real is None for w in grouped)          # members never built engines of their own
        fused2 = model(pixels * 1.03, ids)                                  # a second pass: fresh products
        assert all(g.launches == 2 for g in groups.values())
        qu.model_quant(model, types.SimpleNamespace(skip_names=[], no_sibling_fusion=True))
        assert not _groups(wrappers)
        plain = model(pixels, ids)
        plain2 = model(pixels * 1.03, ids)
    assert torch.equal(fused, plain) and torch.equal(fused2, plain2)
    assert not torch.equal(fused, fused2)


def test_sibling_group_never_serves_a_stale_or_foreign_product():
    from fake_quant import quant_utils as qu
    model, wrappers, pixels, ids, _ = build(96, False, False, 8, False)
    calibrate(model, pixels, ids, None)
    attn = model.model.layers[0].self_attn
    q, k, v = attn.q_proj, attn.k_proj, attn.v_proj
    grp = q.__dict__["_group"]
    assert grp is not None and grp is k.__dict__["_group"] is v.__dict__["_group"]
    x = torch.randn(7, q.module.in_features, device=DEV)
    y_q = q(x).clone()
    assert grp.launches == 1
    y_k = k(x).clone()                       # served from the product q computed
    assert grp.launches == 1
    y_q2 = q(x)                              # q took this product already: recomputed, not re-served
    assert grp.launches == 2 and torch.equal(y_q, y_q2)
    x.mul_(1.5)                              # same storage, new contents: the version counter moves
    y_k2 = k(x)
    assert grp.launches == 3 and not torch.equal(y_k, y_k2)
    # reference values from the members' own engines
    qu.model_quant(model, types.SimpleNamespace(skip_names=[], no_sibling_fusion=True))
    assert torch.equal(k(x), y_k2) and torch.equal(q(x), q(x)) and q._real is not None and k._real is not None
    # siblings fed DIFFERENT tensors: every product goes unused, the group dissolves itself
    qu.model_quant(model, Args())
    grp = q.__dict__["_group"]
    for i in range(4):
        for m in (q, k, v):
            m(torch.randn(7, q.module.in_features, device=DEV))
    assert not grp.enabled and q.__dict__.get("_group") is None
    assert torch.equal(k(x), y_k2)


def test_sibling_group_drops_its_cache_at_every_forward_pass_of_the_parent():
    """A partially consumed product must not survive into the next pass when the same buffer was refilled through a path the
    version counter does not see (``.data``, this repository's in-place kernels, inference tensors): the parent module's
    forward-pre-hook empties the cache; the hook pickles and deep-copies as a no-op."""
    import copy
    import pickle
    from fake_quant import quant_utils as qu
    model, wrappers, pixels, ids, _ = build(96, False, False, 8, False)
    calibrate(model, pixels, ids, None)
    attn = model.model.layers[0].self_attn
    q, k = attn.q_proj, attn.k_proj
    grp = q.__dict__["_group"]
    assert grp._hook is not None
    x = torch.randn(7, q.module.in_features, device=DEV)
    y_q = q(x).clone()                       # product computed, only q consumed it
    x.data.copy_(torch.randn_like(x))        # refill through .data: the version counter does not move
    stale_key_would_match = True             # (same storage, same version: without the pass boundary k would get the old product)
    for h in attn._forward_pre_hooks.values():
        h(attn, ())                          # what the parent's next forward pass does first
    launches = grp.launches
    y_k = k(x)
    assert grp.launches == launches + 1 and stale_key_would_match      # recomputed from the refilled buffer
    qu.model_quant(model, types.SimpleNamespace(skip_names=[], no_sibling_fusion=True))
    assert torch.equal(k(x), y_k)
    # inference tensors have no version counter: without a parent hook the group does not share at all
    qu.model_quant(model, Args())
    grp = q.__dict__["_group"]
    grp._hook.remove()
    grp._hook = None
    with torch.inference_mode():
        xi = torch.randn(7, q.module.in_features, device=DEV)
        q(xi)
        n = grp.launches
        k(xi)
        assert grp.launches == n             # no product was cached or served: both ran alone
    # the parent pickles / deep-copies with the hook as a no-op
    qu.model_quant(model, Args())
    clone = copy.deepcopy(attn)
    blob = pickle.dumps(attn)
    for m in (clone, pickle.loads(blob)):
        for h in m._forward_pre_hooks.values():
            assert h(m, ()) is None
