"""Offline weight surgery shared by the per-model rotation passes: LayerNorm fusion, mean
baking and orthogonal rotations, all evaluated in fp64 and cast back to the weight's dtype
(reference: ``fake_quant/rotation_utils.py``).

Conventions (x is a row vector, Linear computes x W^T + b):
    rotate the INPUT  space of a Linear by Q:   W <- W Q            (``rotate_linear_input_``)
    rotate the OUTPUT space of a Linear by Q:   W <- Q^T W, b <- Q^T b  (``rotate_linear_output_``)
so that  (x Q) (W Q)^T = x W^T  and  (x W^T + b) Q = x (Q^T W)^T + Q^T b.
"""
import typing

import torch
import tqdm

from fake_quant import module_util, utils
from fake_quant.hadamard_utils import (  # noqa: F401  (re-exported like upstream)
    apply_exact_had_to_linear,
    is_pow2,
    random_hadamard_matrix,
)


# ------------------------------------------------------------------------------- primitives
def _as64(t: torch.Tensor, like: torch.Tensor = None) -> torch.Tensor:
    t = t.to(torch.float64)
    return t if like is None else t.to(like.device)


def q_to(Q: torch.Tensor, device) -> torch.Tensor:
    """``Q.to(device)`` that keeps the structure tag of a random Hadamard rotation."""
    Qd = Q.to(device)
    if Qd is not Q and getattr(Q, "_mq_signs", None) is not None and _structured_tag_alive(Q):
        Qd._mq_signs = Q._mq_signs
        Qd._mq_tag_version = utils.tensor_version(Qd)
    return Qd


def _structured(Q: torch.Tensor) -> bool:
    """Q = diag(s) H_n / sqrt(n) from ``random_hadamard_matrix`` and a GPU to run on: the product
    is a sign flip plus a fast Hadamard per row (``mq_rotate_f64``) instead of a dense fp64 GEMM."""
    if getattr(Q, "_mq_signs", None) is None or not torch.cuda.is_available():
        return False
    return _tag_verified(Q)


def _structured_tag_alive(Q: torch.Tensor) -> bool:
    """False once Q was written in place after ``random_hadamard_matrix`` tagged it."""
    ver = utils.tensor_version(Q)
    if Q.__dict__.get("_mq_tag_version", ver) != ver:
        Q._mq_signs = None
        return False
    return True


def _tag_verified(Q: torch.Tensor) -> bool:
    """The structure tag is an attribute of the tensor OBJECT, not of its contents: an in-place edit of Q
    (or a reused tagged tensor) would make the fast path apply diag(s) H / sqrt(n) instead of the matrix
    actually passed.  Per (object, version counter): compare a few rows of the dense Q with the structured
    transform of the matching unit vectors; on any mismatch the tag is dropped and the dense product runs."""
    if not _structured_tag_alive(Q):
        return False
    stamp = (utils.tensor_version(Q), Q.data_ptr())
    seen = Q.__dict__.get("_mq_tag_ok")
    if seen is not None and seen[0] == stamp:
        return seen[1]
    from fake_quant import hadamard_utils as hu
    n = Q.shape[0]
    ok = Q.dim() == 2 and Q.shape[1] == n and Q._mq_signs.numel() == n
    if ok:
        rows = sorted({0, 1, n // 3, n // 2, n - 2, n - 1} & set(range(n)))
        eye = torch.zeros((len(rows), n), dtype=torch.float64)
        eye[torch.arange(len(rows)), torch.tensor(rows)] = 1.0
        want = hu.matmul_hadU(eye * Q._mq_signs.to(dtype=torch.float64, device="cpu")[None, :])   # e_r diag(s) H / sqrt(n)
        got = Q[rows].to(device="cpu", dtype=torch.float64)
        ok = bool(torch.allclose(got, want, rtol=0, atol=1e-9))
    if not ok:
        Q._mq_signs = None
    Q.__dict__["_mq_tag_ok"] = (stamp, ok)
    return ok


def _signs_on(Q: torch.Tensor, device) -> torch.Tensor:
    cache = Q.__dict__.setdefault("_mq_signs_dev", {})
    key = (device.type, device.index)
    if key not in cache:
        cache[key] = Q._mq_signs.to(device=device, dtype=torch.float64).contiguous()
    return cache[key]


def mul_q(X: torch.Tensor, Q: torch.Tensor, dtype: torch.dtype = None) -> torch.Tensor:
    """``(X.double() @ Q).to(dtype)`` over the last dim of X (|Q| = X.shape[-1]); dtype defaults to
    X's.  Random-Hadamard Q on a GPU box: one fused fp64 launch; otherwise the dense product."""
    dtype = X.dtype if dtype is None else dtype
    if not _structured(Q):
        return (X.double() @ _as64(Q, X)).to(dtype)
    from fake_quant import hadamard_utils as hu
    from mquant_amd import ops
    work = X.device if X.is_cuda else torch.device("cuda", torch.cuda.current_device())
    n = Q.shape[0]
    assert X.shape[-1] == n
    Xw = X.to(work)
    Xw = Xw.clone(memory_format=torch.contiguous_format) if Xw.data_ptr() == X.data_ptr() or not Xw.is_contiguous() else Xw
    if Xw.dtype != dtype:                                  # the kernel casts back to the dtype it read
        Xw = Xw.double()
    _, K = hu.get_hadK(n)
    words = None if K == 1 else hu.had_sign_bits(K, work, prepared=False)
    ops.rotate_f64_(Xw, _signs_on(Q, work), K, words)
    return Xw.to(device=X.device, dtype=dtype)


def mul_qt(Q: torch.Tensor, W: torch.Tensor, dtype: torch.dtype = None) -> torch.Tensor:
    """``(Q.T.double() @ W.double()).to(dtype)`` for a matrix or a vector W."""
    dtype = W.dtype if dtype is None else dtype
    if not _structured(Q):
        return (_as64(Q, W).T @ W.double()).to(dtype)
    if W.dim() == 1:
        return mul_q(W, Q, dtype)                          # Q^T b == b Q for a vector
    return mul_q(W.t().contiguous(), Q, dtype).t().contiguous()


def rotate_linear_input_(linear, Q: torch.Tensor) -> None:
    linear.weight.data = mul_q(linear.weight.data, Q)


def rotate_linear_output_(linear, Q: torch.Tensor) -> None:
    W = linear.weight.data
    linear.weight.data = mul_qt(Q, W)
    if linear.bias is not None:
        linear.bias.data = mul_qt(Q, linear.bias.data, W.dtype)


def rotate_grouped_input_(linear, Q: torch.Tensor) -> None:
    """Input features are several consecutive vectors of size |Q| (e.g. the 2x2 patch merger):
    every group is rotated by the same Q."""
    W = linear.weight.data
    out_f, in_f = W.shape
    g = Q.shape[0]
    linear.weight.data = mul_q(W.reshape(out_f, -1, g), Q).reshape(out_f, in_f).contiguous()


def rotate_vector_(param: torch.Tensor, Q: torch.Tensor) -> None:
    """param <- param Q for embedding tables / positional tensors whose last dim is the model dim."""
    param.data = mul_q(param.data, Q)


# ------------------------------------------------------------------------------- LayerNorm fusion
def _fold_norm(layernorm, linear_layers, grouped: bool) -> None:
    gamma = layernorm.weight.double()
    has_bias_attr = getattr(layernorm, "bias", None) is not None    # RMSNorm: gamma only
    beta = layernorm.bias.double() if has_bias_attr else None
    for lin in linear_layers:
        dt = lin.weight.dtype
        W = lin.weight.data.double()
        if grouped:
            out_f, in_f = W.shape
            Wg = W.view(out_f, -1, gamma.shape[0])
            lin.weight.data = (Wg * gamma).to(dt).view(out_f, in_f)
            shift = (Wg @ beta).sum(dim=-1) if beta is not None else None
        else:
            lin.weight.data = (W * gamma).to(dt)
            shift = W @ beta if beta is not None else None
        if has_bias_attr:
            if lin.bias is None:
                lin.bias = torch.nn.Parameter(torch.zeros(lin.out_features, dtype=torch.float64).to(W))
            lin.bias.data = (lin.bias.data.double() + shift).to(dt)
    layernorm.weight.data = torch.ones_like(layernorm.weight.data)
    if has_bias_attr:
        layernorm.bias.data = torch.zeros_like(layernorm.bias.data)


def fuse_ln_linear(layernorm: torch.nn.Module, linear_layers: typing.Iterable[torch.nn.Linear]) -> None:
    """Fold a norm's affine part (gamma, beta) into the Linears that consume its output:
    W <- W diag(gamma), b <- b + W beta; the norm is left with gamma = 1, beta = 0."""
    _fold_norm(layernorm, linear_layers, grouped=False)


def fuse_merger_linear(layernorm: torch.nn.Module, linear_layers: typing.Iterable[torch.nn.Linear]) -> None:
    """Same for a Linear whose input concatenates several normalised vectors (Qwen2-VL merger)."""
    _fold_norm(layernorm, linear_layers, grouped=True)


def bake_mean_into_conv(conv) -> None:
    """Make the convolution emit zero-mean features (over output channels), i.e. absorb the mean
    subtraction of the following LayerNorm."""
    dt = conv.weight.dtype
    W = conv.weight.data.double()
    conv.weight.data = (W - W.mean(dim=0, keepdim=True)).to(dt)
    if conv.bias is not None:
        b = conv.bias.data.double()
        conv.bias.data = (b - b.mean()).to(dt)


def bake_mean_into_linear(linear: torch.nn.Linear) -> None:
    dt = linear.weight.dtype
    W = linear.weight.data.double()
    linear.weight.data = (W - W.mean(dim=-2, keepdim=True)).to(dt)
    if linear.bias is not None:
        b = linear.bias.data.double()
        linear.bias.data = (b - b.mean()).to(dt)


# ------------------------------------------------------------------------------- rotations
def random_orthogonal_matrix(size, device):
    """QR of a Gaussian matrix with the sign convention diag(R) > 0 (Haar distributed), fp64."""
    if torch.cuda.is_available():
        torch.cuda.empty_cache()
    q, r = torch.linalg.qr(torch.randn(size, size, dtype=torch.float64).to(device))
    q *= torch.sign(torch.diag(r)).unsqueeze(0)
    return q


def get_orthogonal_matrix(size, mode, device=utils.DEV):
    if mode == "random":
        return random_orthogonal_matrix(size, device)
    if mode == "hadamard":
        return random_hadamard_matrix(size, device)
    raise ValueError(f"Unknown mode {mode}")


def rotate_conv(layer, Q_v, embed_dims) -> None:
    """Rotate the OUTPUT channels of a patch-embedding convolution."""
    W = layer.weight.data
    layer.weight.data = mul_qt(Q_v, W.reshape(embed_dims, -1)).view(W.shape)
    if layer.bias is not None:
        layer.bias.data = mul_q(layer.bias.data, Q_v, W.dtype)


def rotate_value_output_heads_(v_weight, v_bias, o_proj, Q_head, head_num, head_dim):
    """Per-head rotation of the V projection's outputs and the O projection's inputs by the same
    head_dim x head_dim matrix.  Returns the new (v_weight, v_bias)."""
    dt = v_weight.dtype
    Wv = v_weight.T.reshape(-1, head_num, head_dim)
    v_weight = mul_q(Wv, Q_head, dt).reshape(-1, head_num * head_dim).T
    if v_bias is not None:
        v_bias = mul_q(v_bias.reshape(head_num, head_dim), Q_head, dt).reshape(-1)
    Wo = o_proj.weight.data.reshape(-1, head_num, head_dim)
    o_proj.weight.data = mul_q(Wo, Q_head, dt).reshape(-1, head_num * head_dim)
    return v_weight, v_bias


def pad_linear_inputs_(model, match: str, new_in: int) -> int:
    """Replace every nn.Linear whose dotted name contains ``match`` by one with ``new_in`` input
    features.  The original columns are copied, the pad columns are ZERO and the new layer has a
    bias only if the old one had (upstream leaves both at their random init)."""
    count = 0
    modules = dict(model.named_modules())
    for name, mod in list(modules.items()):
        if match in name and isinstance(mod, torch.nn.Linear) and mod.in_features != new_in:
            new = torch.nn.Linear(new_in, mod.out_features, bias=mod.bias is not None,
                                  dtype=mod.weight.dtype, device=mod.weight.device)
            with torch.no_grad():
                new.weight.zero_()
                new.weight[:, : mod.in_features] = mod.weight.data
                if mod.bias is not None:
                    new.bias.copy_(mod.bias.data)
            parent_name, _, leaf = name.rpartition(".")
            setattr(modules[parent_name] if parent_name else model, leaf, new)
            count += 1
    return count


# =============================================================================== Qwen-VL (v1)
# The "opt" Qwen-VL checkpoint the drivers load (reference model/visual_opt.py:452-523,
# modeling_qwen_opt.py) has q/k/v split into three Linears, `visual.fc_sub_mean` (mean removal as
# a Linear behind ln_pre) and `visual.proj_fc` (the old `proj` parameter as a Linear).
# MiniCPM-V shares the helpers through ``is_minicpmv`` (other attribute paths, same algebra).
def _pick(obj, *names):
    for n in names:
        if hasattr(obj, n):
            return getattr(obj, n)
    raise AttributeError(f"none of {names} on {type(obj).__name__}")


def _scale_param_(param, gamma):
    """param <- param / gamma: a positional term added AFTER a norm moves in front of its affine."""
    param.data = (param.data.double() / gamma.data.double()).to(param.data.dtype)


def fuse_qwenvl_layer_norms(model, args):
    print("fuse qwenvl layer norms")
    vis = model.transformer.visual
    if not args.no_fuse_visual_clip:
        for blk in vis.transformer.resblocks:
            fuse_ln_linear(blk.ln_1, [blk.attn.q_proj, blk.attn.k_proj, blk.attn.v_proj])
            fuse_ln_linear(blk.ln_2, [blk.mlp.c_fc])
            bake_mean_into_linear(blk.attn.out_proj)
            bake_mean_into_linear(blk.mlp.c_proj)
        module_util.replace_modules(vis.transformer.resblocks, torch.nn.LayerNorm,
                                    lambda _: module_util.RMSN(model.config.visual["width"], eps=1e-6),
                                    replace_layers=False)
    if not args.no_fuse_visual_cross_attn:
        pool = vis.attn_pool
        _scale_param_(pool.pos_embed_kv, pool.ln_kv.weight)
        fuse_ln_linear(pool.ln_kv, [pool.attn.k_proj, pool.attn.v_proj])
        _scale_param_(pool.pos_embed, pool.ln_q.weight)
        fuse_ln_linear(pool.ln_q, [pool.attn.q_proj])
        pool.query.data = (pool.query.data - pool.query.data.double().mean(dim=-1, keepdim=True)).to(pool.query.data.dtype)
        bake_mean_into_linear(pool.kv_proj)
        module_util.replace_modules(pool, torch.nn.LayerNorm,
                                    lambda _: module_util.RMSN(model.config.visual["output_dim"], eps=1e-6),
                                    replace_layers=False)
        fuse_ln_linear(vis.ln_post, [vis.proj_fc])
        bake_mean_into_linear(pool.attn.out_proj)
        vis.ln_post = module_util.RMSN(model.config.visual["output_dim"], eps=1e-6)
    if not args.no_fuse_llm:
        for layer in model.transformer.h:
            fuse_ln_linear(layer.ln_2, [layer.mlp.w1, layer.mlp.w2])
            fuse_ln_linear(layer.ln_1, [layer.attn.q_proj, layer.attn.k_proj, layer.attn.v_proj])
        fuse_ln_linear(model.transformer.ln_f, [model.lm_head])


def _rotate_bias_row_(linear, Q):
    if linear.bias is not None:
        rotate_vector_(linear.bias, Q)


def rotate_embeddings(model, Q, is_minicpmv=False) -> None:
    table, proj = ((model.llm.model.embed_tokens, model.resampler.proj_fc) if is_minicpmv
                   else (model.transformer.wte, model.transformer.visual.proj_fc))
    rotate_vector_(table.weight, Q)
    rotate_linear_output_(proj, Q)          # the projector writes into the LLM residual stream


def rotate_head(model, Q: torch.Tensor, is_minicpmv=False) -> None:
    rotate_linear_input_(model.llm.lm_head if is_minicpmv else model.lm_head, Q)


def rotate_kv_proj(model, Q: torch.Tensor, is_minicpmv=False) -> None:
    rotate_linear_input_(model.resampler.kv_proj if is_minicpmv else model.transformer.visual.attn_pool.kv_proj, Q)


def rotate_attention_inputs(layer, Q, is_minicpmv=False) -> None:
    att = layer.self_attn if is_minicpmv else layer.attn
    for lin in (att.q_proj, att.k_proj, att.v_proj):
        rotate_linear_input_(lin, Q)


def rotate_cross_attention_inputs(layer, Q_q, Q_kv) -> None:
    rotate_linear_input_(layer.attn.q_proj, Q_q)
    rotate_linear_input_(layer.attn.k_proj, Q_kv)
    rotate_linear_input_(layer.attn.v_proj, Q_kv)


def rotate_cross_embeddings(model, Q_q, Q_kv, is_minicpmv=False):
    pool = model.resampler if is_minicpmv else model.transformer.visual.attn_pool
    rotate_vector_(pool.query, Q_q)
    if not is_minicpmv:
        rotate_vector_(pool.pos_embed, Q_q)
    rotate_linear_output_(pool.kv_proj, Q_kv)
    rotate_vector_(pool.pos_embed if is_minicpmv else pool.pos_embed_kv, Q_kv)


def rotate_attention_output(layer, Q, is_visual=False) -> None:
    att = _pick(layer, "attn", "self_attn")
    rotate_linear_output_(_pick(att, "out_proj") if is_visual else _pick(att, "c_proj", "o_proj"), Q)


def rotate_mlp_input(layer, Q, is_visual=False, is_minicpmv=False) -> None:
    if is_visual:
        targets = [_pick(layer.mlp, "c_fc", "fc1")]
    else:
        targets = [layer.mlp.up_proj, layer.mlp.gate_proj] if is_minicpmv else [layer.mlp.w1, layer.mlp.w2]
    for lin in targets:
        rotate_linear_input_(lin, Q)


def rotate_mlp_output(layer, Q, online_hadamard=False):
    out = _pick(layer.mlp, "c_proj", "down_proj", "fc2")
    rotate_linear_output_(out, Q)               # weight and bias; the Hadamard below is input-side only
    if online_hadamard:
        apply_exact_had_to_linear(out, had_dim=-1, output=False)


def rotate_ov_proj(layer, head_num, head_dim, is_visual=False, is_minicpmv=False):
    att = _pick(layer, "attn", "self_attn")
    v_proj = att.v_proj
    if is_visual:
        o_proj = att.out_proj
        Qh = get_orthogonal_matrix(head_dim, mode="hadamard")
        v_w, v_b = rotate_value_output_heads_(v_proj.weight.data, None if v_proj.bias is None else v_proj.bias.data,
                                              o_proj, Qh, head_num, head_dim)
        v_proj.weight.data = v_w
        if v_b is not None:
            v_proj.bias.data = v_b
    else:
        o_proj = att.o_proj if is_minicpmv else att.c_proj
        apply_exact_had_to_linear(v_proj, had_dim=head_dim, output=True)
        apply_exact_had_to_linear(o_proj, had_dim=head_dim, output=False)


def rotate_o_ln_proj(layer, Q_o):
    """Older layout: ``proj`` is a bare parameter and ln_post keeps its bias."""
    rotate_linear_output_(layer.attn_pool.attn.out_proj, Q_o)
    rotate_vector_(layer.ln_post.bias, Q_o)
    layer.proj.data = mul_qt(Q_o, layer.proj.data)


def rotate_o_ln_proj_fc(layer, Q_o, is_minicpmv=False):
    rotate_linear_output_((layer.attn if is_minicpmv else layer.attn_pool.attn).out_proj, Q_o)
    rotate_linear_input_(layer.proj_fc, Q_o)


@torch.inference_mode()
def rotate_model(model, args):
    """Qwen-VL (v1, "opt" layout) driver."""
    print("rotate model")
    vis = model.transformer.visual
    vcfg = model.config.visual
    if args.rotate_visual_clip:
        heads = vcfg["heads"]
        Q_v = get_orthogonal_matrix(vcfg["width"], args.rotate_mode)
        for blk in tqdm.tqdm(vis.transformer.resblocks, unit="layer", desc="Rotating Visual CLIP"):
            rotate_attention_inputs(blk, Q_v)
            rotate_attention_output(blk, Q_v, is_visual=True)
            rotate_mlp_input(blk, Q_v, is_visual=True)
            rotate_mlp_output(blk, Q_v, args.online_visual_hadamard)
            rotate_ov_proj(blk, heads, vcfg["width"] // heads, is_visual=True)
        rotate_kv_proj(model, Q_v)
        rotate_linear_output_(vis.fc_sub_mean, Q_v)      # the stream enters the ViT through it
        utils.cleanup_memory()

    if args.rotate_visual_cross_attn:
        print("\n Rotating Visual Cross Attention \n")
        pool = vis.attn_pool
        Q_q = get_orthogonal_matrix(vcfg["output_dim"], args.rotate_mode)
        Q_kv = get_orthogonal_matrix(vcfg["output_dim"], args.rotate_mode)
        rotate_cross_embeddings(model, Q_q, Q_kv)
        rotate_cross_attention_inputs(pool, Q_q, Q_kv)
        rotate_ov_proj(pool, pool.num_heads, pool.embed_dim // pool.num_heads, is_visual=True)
        Q_o = get_orthogonal_matrix(vcfg["output_dim"], args.rotate_mode)
        rotate_o_ln_proj_fc(vis, Q_o)
        utils.cleanup_memory()

    if args.rotate_llm:
        cfg = model.config
        if args.online_llm_hadamard:
            cfg.need_pad = False
            from fake_quant.hadamard_utils import auto_pad_size
            padded = auto_pad_size(cfg.intermediate_size)
            if padded != cfg.intermediate_size:
                pad_linear_inputs_(model.transformer.h, "mlp.c_proj", padded)
                cfg.intermediate_size = padded
                cfg.need_pad = True
        Q = get_orthogonal_matrix(cfg.hidden_size, args.rotate_mode)
        head_dim = cfg.hidden_size // cfg.num_attention_heads
        rotate_embeddings(model, Q)
        rotate_head(model, Q)
        utils.cleanup_memory()
        for layer in tqdm.tqdm(model.transformer.h, unit="layer", desc="Rotating"):
            rotate_attention_inputs(layer, Q)
            rotate_attention_output(layer, Q)
            rotate_mlp_input(layer, Q)
            rotate_mlp_output(layer, Q, args.online_llm_hadamard)
            rotate_ov_proj(layer, cfg.num_attention_heads, head_dim)
        utils.cleanup_memory()


__all__ = [n for n in dir() if not n.startswith("_")] + ["module_util"]
