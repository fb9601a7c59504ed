"""LayerNorm fusion and rotation for Qwen2-VL (reference: ``fake_quant/qwen2vl_rotation.py``).

``fuse_qwen2vl_layer_norms(model, args)`` takes the VLMEvalKit wrapper (``model.model`` is the HF
module); ``rotate_qwen2vl_model(model, args)`` takes the HF module itself, as upstream.
"""
import torch
import tqdm

from fake_quant import module_util, utils
from fake_quant.hadamard_utils import apply_exact_had_to_linear, auto_pad_size
from fake_quant.rotation_utils import (
    bake_mean_into_conv,
    bake_mean_into_linear,
    fuse_ln_linear,
    fuse_merger_linear,
    get_orthogonal_matrix,
    mul_q,
    mul_qt,
    pad_linear_inputs_,
    q_to,
    rotate_conv,
    rotate_grouped_input_,
    rotate_linear_input_,
    rotate_linear_output_,
    rotate_value_output_heads_,
    rotate_vector_,
)


def fuse_qwen2vl_layer_norms(model, args):
    print("fuse qwen2vl layer norms")
    hf = model.model
    vis = hf.visual
    if not args.no_fuse_visual_clip:
        bake_mean_into_conv(vis.patch_embed.proj)
        for blk in vis.blocks:
            fuse_ln_linear(blk.norm1, [blk.attn.qkv])
            fuse_ln_linear(blk.norm2, [blk.mlp.fc1])
            bake_mean_into_linear(blk.attn.proj)
            bake_mean_into_linear(blk.mlp.fc2)
        module_util.replace_modules(vis.blocks, torch.nn.LayerNorm,
                                    lambda _: module_util.RMSN(vis.patch_embed.embed_dim, eps=1e-6),
                                    replace_layers=False)
    if not args.no_fuse_visual_cross_attn:
        fuse_merger_linear(vis.merger.ln_q, [vis.merger.mlp[0]])
        module_util.replace_modules(vis.merger, torch.nn.LayerNorm,
                                    lambda _: module_util.RMSN(vis.patch_embed.embed_dim, eps=1e-6),
                                    replace_layers=False)
    if not args.no_fuse_llm:
        for layer in hf.model.layers:
            att = layer.self_attn
            fuse_ln_linear(layer.input_layernorm, [att.q_proj, att.k_proj, att.v_proj])
            fuse_ln_linear(layer.post_attention_layernorm, [layer.mlp.gate_proj, layer.mlp.up_proj])
        fuse_ln_linear(hf.model.norm, [hf.lm_head])      # the final norm folds into the head


# ---- per-layer pieces (names kept: the drivers and notebooks call them directly) ----------------
def rotate_qwen2vl_attention_inputs(layer, Q, is_visual=False) -> None:
    targets = [layer.attn.qkv] if is_visual else [layer.self_attn.q_proj, layer.self_attn.k_proj,
                                                 layer.self_attn.v_proj]
    for lin in targets:
        rotate_linear_input_(lin, Q)


def rotate_qwen2vl_attention_output(layer, Q, is_visual=False) -> None:
    rotate_linear_output_(layer.attn.proj if is_visual else layer.self_attn.o_proj, Q)


def rotate_qwen2vl_mlp_input(layer, Q, is_visual=False) -> None:
    for lin in ([layer.mlp.fc1] if is_visual else [layer.mlp.gate_proj, layer.mlp.up_proj]):
        rotate_linear_input_(lin, Q)


def rotate_qwen2vl_mlp_output(layer, Q, is_visual=False, online_hadamard=False):
    out = layer.mlp.fc2 if is_visual else layer.mlp.down_proj
    bias = out.bias
    out.bias = None                                   # the bias is rotated after the Hadamard step
    rotate_linear_output_(out, Q)
    if online_hadamard:                               # its input gets the online Hadamard at run time
        apply_exact_had_to_linear(out, had_dim=-1, output=False)
    out.bias = bias
    if bias is not None:
        out.bias.data = mul_qt(Q, bias.data, out.weight.data.dtype)


def rotate_qwen2vl_ov_proj(layer, head_num, head_dim, is_visual=False):
    if not is_visual:                                 # separate v_proj / o_proj: exact Hadamard per head
        apply_exact_had_to_linear(layer.self_attn.v_proj, had_dim=head_dim, output=True)
        apply_exact_had_to_linear(layer.self_attn.o_proj, had_dim=head_dim, output=False)
        return
    qkv, o_proj = layer.attn.qkv, layer.attn.proj
    q_w, k_w, v_w = qkv.weight.data.chunk(3)
    Qh = get_orthogonal_matrix(head_dim, mode="hadamard")
    v_b = None
    if qkv.bias is not None:
        q_b, k_b, v_b = qkv.bias.data.chunk(3)
    v_w, v_b = rotate_value_output_heads_(v_w, v_b, o_proj, Qh, head_num, head_dim)
    qkv.weight.data = torch.cat([q_w, k_w, v_w], 0).contiguous()
    if qkv.bias is not None:
        qkv.bias.data = torch.cat([q_b, k_b, v_b], -1).contiguous()


def rotate_visual_merger(model, Q: torch.Tensor) -> None:
    rotate_grouped_input_(model.visual.merger.mlp[0], Q)


def rotate_qwen2vl_embeddings(model, Q) -> None:
    rotate_vector_(model.model.embed_tokens.weight, Q)
    last = model.visual.merger.mlp[2]                 # the merger's output feeds the LLM residual
    W = last.weight.data
    last.weight.data = mul_qt(Q, W)
    if last.bias is not None:
        last.bias.data = mul_q(last.bias.data, Q, W.dtype)


def rotate_qwen2vl_head(model, Q: torch.Tensor) -> None:
    rotate_linear_input_(model.lm_head, Q)


@torch.no_grad()
def rotate_qwen2vl_model(model, args):
    print("rotate model")
    if args.rotate_visual_clip:
        blocks = model.visual.blocks
        width = blocks[0].attn.qkv.in_features
        heads = blocks[0].attn.num_heads
        Q_v = get_orthogonal_matrix(width, args.rotate_mode)
        rotate_conv(model.visual.patch_embed.proj, Q_v, width)
        for layer in tqdm.tqdm(blocks, unit="layer", desc="Rotating Visual CLIP"):
            rotate_qwen2vl_attention_inputs(layer, Q_v, is_visual=True)
            rotate_qwen2vl_attention_output(layer, Q_v, is_visual=True)
            rotate_qwen2vl_mlp_input(layer, Q_v, is_visual=True)
            rotate_qwen2vl_mlp_output(layer, Q_v, True, args.online_visual_hadamard)
            rotate_qwen2vl_ov_proj(layer, heads, width // heads, is_visual=True)
        rotate_visual_merger(model, Q_v)
        utils.cleanup_memory()

    if args.rotate_visual_cross_attn:
        print("\n Rotating Visual Cross Attention \n")     # nothing to do, as upstream

    if args.rotate_llm:
        cfg = model.config
        if args.online_llm_hadamard:
            cfg.need_pad = False
            padded = auto_pad_size(cfg.intermediate_size)
            if padded != cfg.intermediate_size:           # 18944 -> 19968, 29568 -> 30720
                pad_linear_inputs_(model, "down_proj", padded)
                cfg.intermediate_size = padded
                cfg.need_pad = True
        Q = get_orthogonal_matrix(cfg.hidden_size, args.rotate_mode)
        head_dim = cfg.hidden_size // cfg.num_attention_heads
        rotate_qwen2vl_embeddings(model, Q)
        rotate_qwen2vl_head(model, Q)
        utils.cleanup_memory()
        for layer in tqdm.tqdm(model.model.layers, unit="layer", desc="LLM Rotating"):
            Q = q_to(Q, next(layer.parameters()).device)
            rotate_qwen2vl_attention_inputs(layer, Q)
            rotate_qwen2vl_attention_output(layer, Q)
            rotate_qwen2vl_mlp_input(layer, Q)
            rotate_qwen2vl_mlp_output(layer, Q, False, args.online_llm_hadamard)
            rotate_qwen2vl_ov_proj(layer, cfg.num_attention_heads, head_dim, is_visual=False)
        utils.cleanup_memory()
