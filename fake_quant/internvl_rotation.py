"""LayerNorm fusion and rotation for InternVL2 (InternViT + InternLM2); reference:
``fake_quant/internvl_rotation.py``.  InternLM2 packs q/k/v of every KV group into one ``wqkv``
(per group: ``num_key_value_groups`` query heads, then K, then V), which the V/O head rotation
has to respect (``rotate_internvl_ov_proj_v2``)."""
import torch
import tqdm

from fake_quant import module_util, utils
from fake_quant.hadamard_utils import apply_exact_had_to_linear
from fake_quant.rotation_utils import (
    bake_mean_into_conv,
    bake_mean_into_linear,
    fuse_ln_linear,
    get_orthogonal_matrix,
    mul_q,
    mul_qt,
    rotate_conv,
    rotate_grouped_input_,
    rotate_linear_input_,
    rotate_linear_output_,
    rotate_value_output_heads_,
    rotate_vector_,
)


def _center_(param):
    """Remove the mean over the feature dim (the following LayerNorm's mean subtraction)."""
    param.data = (param.data - param.data.double().mean(dim=-1, keepdim=True)).to(param.data.dtype)


def fuse_internvl_layer_norms(model, args):
    print("fuse internvl layer norms")
    hf = model.model
    vm = hf.vision_model
    if not args.no_fuse_visual_clip:
        bake_mean_into_conv(vm.embeddings.patch_embedding)
        _center_(vm.embeddings.class_embedding)
        _center_(vm.embeddings.position_embedding)
        for layer in vm.encoder.layers:
            fuse_ln_linear(layer.norm1, [layer.attn.qkv])
            fuse_ln_linear(layer.norm2, [layer.mlp.fc1])
            bake_mean_into_linear(layer.attn.proj)
            bake_mean_into_linear(layer.mlp.fc2)
        module_util.replace_modules(vm.encoder.layers, torch.nn.LayerNorm,
                                    lambda _: module_util.RMSN(vm.encoder.config.hidden_size, eps=1e-6),
                                    replace_layers=False)
    if not args.no_fuse_visual_cross_attn:
        fuse_ln_linear(hf.mlp1[0], [hf.mlp1[1]])
        width = vm.encoder.config.hidden_size * int(1 / hf.config.downsample_ratio) ** 2
        module_util.replace_modules(hf.mlp1, torch.nn.LayerNorm,
                                    lambda _: module_util.RMSN(width, eps=1e-6), replace_layers=False)
    if not args.no_fuse_llm:
        lm = hf.language_model
        for layer in lm.model.layers:
            fuse_ln_linear(layer.attention_norm, [layer.attention.wqkv])
            fuse_ln_linear(layer.ffn_norm, [layer.feed_forward.w1, layer.feed_forward.w3])
        fuse_ln_linear(lm.model.norm, [lm.output])


def rotate_internvl_attention_inputs(layer, Q, is_visual=False) -> None:
    rotate_linear_input_(layer.attn.qkv if is_visual else layer.attention.wqkv, Q)


def rotate_internvl_attention_output(layer, Q, is_visual=False) -> None:
    rotate_linear_output_(layer.attn.proj if is_visual else layer.attention.wo, Q)


def rotate_internvl_mlp_input(layer, Q, is_visual=False) -> None:
    for lin in ([layer.mlp.fc1] if is_visual else [layer.feed_forward.w1, layer.feed_forward.w3]):
        rotate_linear_input_(lin, Q)


def rotate_internvl_mlp_output(layer, Q, is_visual=False, online_hadamard=False):
    out = layer.mlp.fc2 if is_visual else layer.feed_forward.w2
    bias = out.bias
    out.bias = None
    rotate_linear_output_(out, Q)
    if online_hadamard:
        apply_exact_had_to_linear(out, had_dim=-1, output=False)
    out.bias = bias
    if bias is not None:
        out.bias.data = mul_qt(Q, bias.data, out.weight.data.dtype)


def rotate_internvl_ov_proj(layer, head_num, head_dim, is_visual=False):
    """q/k/v stacked as three equal chunks (InternViT; also MHA language models)."""
    qkv, o_proj = (layer.attn.qkv, layer.attn.proj) if is_visual else (layer.attention.wqkv, layer.attention.wo)
    q_w, k_w, v_w = qkv.weight.data.chunk(3)
    Qh = get_orthogonal_matrix(head_dim, mode="hadamard")
    v_b = None
    if qkv.bias is not None:
        q_b, k_b, v_b = qkv.bias.data.chunk(3)
    v_w, v_b = rotate_value_output_heads_(v_w, v_b, o_proj, Qh, head_num, head_dim)
    qkv.weight.data = torch.cat([q_w, k_w, v_w], 0).contiguous()
    if qkv.bias is not None:
        qkv.bias.data = torch.cat([q_b, k_b, v_b], -1).contiguous()


def rotate_internvl_ov_proj_v2(layer, q_head_num, kv_head_num, head_dim):
    """GQA layout of InternLM2's wqkv: rows grouped per KV head as [q x groups, k, v]."""
    qkv, o_proj = layer.attention.wqkv, layer.attention.wo
    groups = q_head_num // kv_head_num
    W = qkv.weight.data
    out_f, in_f = W.shape
    Wt = W.T.contiguous().reshape(in_f, -1, 2 + groups, head_dim)
    Qh = get_orthogonal_matrix(head_dim, mode="hadamard").to(W.device)
    Wt[..., -1, :] = (Wt[..., -1, :].double() @ Qh).to(W.dtype)
    qkv.weight.data = Wt.reshape(in_f, out_f).T.contiguous()
    Wo = o_proj.weight.data.double().reshape(-1, q_head_num, head_dim)
    o_proj.weight.data = (Wo @ Qh).reshape(-1, q_head_num * head_dim).to(W.dtype)


def rotate_mlp1(model, Q: torch.Tensor) -> None:
    rotate_grouped_input_(model.mlp1[1], Q)


def rotate_internvl_embeddings(model, Q) -> None:
    rotate_vector_(model.language_model.model.tok_embeddings.weight, Q)
    last = model.mlp1[3]
    W = last.weight.data
    last.weight.data = mul_qt(Q, W)
    if last.bias is not None:
        last.bias.data = mul_q(last.bias.data, Q, W.dtype)


def rotate_internvl_head(model, Q: torch.Tensor) -> None:
    rotate_linear_input_(model.language_model.output, Q)


@torch.inference_mode()
def rotate_internvl2_model(model, args):
    print("rotate model")
    if args.rotate_visual_clip:
        vcfg = model.config.vision_config
        heads = vcfg.num_attention_heads
        Q_v = get_orthogonal_matrix(vcfg.hidden_size, args.rotate_mode)
        emb = model.vision_model.embeddings
        rotate_conv(emb.patch_embedding, Q_v, vcfg.hidden_size)
        rotate_vector_(emb.position_embedding, Q_v)
        rotate_vector_(emb.class_embedding, Q_v)
        for layer in tqdm.tqdm(model.vision_model.encoder.layers, unit="layer", desc="Rotating Visual CLIP"):
            rotate_internvl_attention_inputs(layer, Q_v, is_visual=True)
            rotate_internvl_attention_output(layer, Q_v, is_visual=True)
            rotate_internvl_mlp_input(layer, Q_v, is_visual=True)
            rotate_internvl_mlp_output(layer, Q_v, True, args.online_visual_hadamard)
            rotate_internvl_ov_proj(layer, heads, vcfg.hidden_size // heads, is_visual=True)
        rotate_mlp1(model, Q_v)
        utils.cleanup_memory()          # upstream forgets to import utils here (NameError); fixed

    if args.rotate_visual_cross_attn:
        print("\n Rotating Visual Cross Attention \n")

    if args.rotate_llm:
        cfg = model.config.llm_config
        Q = get_orthogonal_matrix(cfg.hidden_size, args.rotate_mode)
        head_dim = cfg.hidden_size // cfg.num_attention_heads
        rotate_internvl_embeddings(model, Q)
        rotate_internvl_head(model, Q)
        utils.cleanup_memory()
        for layer in tqdm.tqdm(model.language_model.model.layers, unit="layer", desc="LLM Rotating"):
            rotate_internvl_attention_inputs(layer, Q)
            rotate_internvl_attention_output(layer, Q)
            rotate_internvl_mlp_input(layer, Q)
            rotate_internvl_mlp_output(layer, Q, False, args.online_llm_hadamard)
            rotate_internvl_ov_proj_v2(layer, cfg.num_attention_heads, cfg.num_key_value_heads, head_dim)
        utils.cleanup_memory()
