"""LayerNorm fusion and rotation for MiniCPM-V (SigLIP tower ``vpm``, ``resampler``, ``llm``);
reference: ``fake_quant/minicpmv_rotation.py``.  Both passes take the HF module itself."""
import torch
import tqdm

from fake_quant import module_util, utils
from fake_quant.hadamard_utils import auto_pad_size
from fake_quant.rotation_utils import (
    bake_mean_into_conv,
    bake_mean_into_linear,
    fuse_ln_linear,
    get_orthogonal_matrix,
    pad_linear_inputs_,
    rotate_attention_inputs,
    rotate_attention_output,
    rotate_conv,
    rotate_cross_attention_inputs,
    rotate_cross_embeddings,
    rotate_embeddings,
    rotate_head,
    rotate_kv_proj,
    rotate_mlp_input,
    rotate_mlp_output,
    rotate_o_ln_proj_fc,
    rotate_ov_proj,
    rotate_vector_,
)


def _center_(param):
    param.data = (param.data - param.data.double().mean(dim=-1, keepdim=True)).to(param.data.dtype)


def fuse_minicpmv_layer_norms(model, args):
    print("fuse minicpmv layer norms")
    vpm, res = model.vpm, model.resampler
    if not args.no_fuse_visual_clip:
        bake_mean_into_conv(vpm.embeddings.patch_embedding)
        _center_(vpm.embeddings.position_embedding.weight)
        for layer in vpm.encoder.layers:
            att = layer.self_attn
            fuse_ln_linear(layer.layer_norm1, [att.q_proj, att.k_proj, att.v_proj])
            fuse_ln_linear(layer.layer_norm2, [layer.mlp.fc1])
            bake_mean_into_linear(att.out_proj)
            bake_mean_into_linear(layer.mlp.fc2)
        fuse_ln_linear(vpm.post_layernorm, [res.kv_proj])        # the tower's last norm feeds the resampler
        module_util.replace_modules(vpm.encoder.layers, torch.nn.LayerNorm,
                                    lambda _: module_util.RMSN(vpm.embed_dim, eps=1e-6), replace_layers=False)
    if not args.no_fuse_visual_cross_attn:
        res.pos_embed.data = (res.pos_embed.data.double() / res.ln_kv.weight.data.double()).to(res.pos_embed.data.dtype)
        fuse_ln_linear(res.ln_kv, [res.attn.k_proj, res.attn.v_proj])
        fuse_ln_linear(res.ln_q, [res.attn.q_proj])
        fuse_ln_linear(res.ln_post, [res.proj_fc])
        _center_(res.query)
        bake_mean_into_linear(res.kv_proj)
        bake_mean_into_linear(res.attn.out_proj)
        module_util.replace_modules(res, torch.nn.LayerNorm,
                                    lambda _: module_util.RMSN(res.embed_dim, eps=1e-6), replace_layers=False)
    if not args.no_fuse_llm:
        for layer in model.llm.model.layers:
            att = layer.self_attn
            fuse_ln_linear(layer.post_attention_layernorm, [layer.mlp.up_proj, layer.mlp.gate_proj])
            fuse_ln_linear(layer.input_layernorm, [att.q_proj, att.k_proj, att.v_proj])
        fuse_ln_linear(model.llm.model.norm, [model.llm.lm_head])


@torch.no_grad()
def rotate_minicpmv_model(model, args):
    print("rotate model")
    if args.rotate_visual_clip:
        emb = model.vpm.embeddings
        Q_v = get_orthogonal_matrix(emb.embed_dim, args.rotate_mode)
        rotate_conv(emb.patch_embedding, Q_v, emb.embed_dim)
        rotate_vector_(emb.position_embedding.weight, Q_v)
        if args.online_visual_hadamard:
            vcfg = model.config.vision_config
            vcfg.need_pad = False
            padded = auto_pad_size(vcfg.intermediate_size)       # SigLIP 4304 -> 4480
            if padded != vcfg.intermediate_size:
                pad_linear_inputs_(model, "mlp.fc2", padded)
                vcfg.intermediate_size = padded
                vcfg.need_pad = True
        for layer in tqdm.tqdm(model.vpm.encoder.layers, unit="layer", desc="Rotating Visual CLIP"):
            rotate_attention_inputs(layer, Q_v, is_minicpmv=True)
            rotate_attention_output(layer, Q_v, is_visual=True)
            rotate_mlp_input(layer, Q_v, is_visual=True)
            rotate_mlp_output(layer, Q_v, args.online_visual_hadamard)
            rotate_ov_proj(layer, layer.self_attn.num_heads, layer.self_attn.head_dim, is_visual=True)
        rotate_kv_proj(model, Q_v, is_minicpmv=True)
        utils.cleanup_memory()

    if args.rotate_visual_cross_attn:
        print("Rotating Visual Cross Attention")
        res = model.resampler
        Q_q = get_orthogonal_matrix(res.embed_dim, args.rotate_mode)
        Q_kv = get_orthogonal_matrix(res.embed_dim, args.rotate_mode)
        rotate_cross_embeddings(model, Q_q, Q_kv, is_minicpmv=True)
        rotate_cross_attention_inputs(res, Q_q, Q_kv)
        rotate_ov_proj(res, res.num_heads, res.embed_dim // res.num_heads, is_visual=True)
        Q_o = get_orthogonal_matrix(model.config.hidden_size, args.rotate_mode)
        rotate_o_ln_proj_fc(res, Q_o, is_minicpmv=True)
        utils.cleanup_memory()

    if args.rotate_llm:
        cfg = model.config
        if args.online_llm_hadamard:
            cfg.need_pad = False
            padded = auto_pad_size(cfg.intermediate_size)
            if padded != cfg.intermediate_size:
                pad_linear_inputs_(model, "down_proj", padded)
                cfg.intermediate_size = padded
                cfg.need_pad = True
        Q = get_orthogonal_matrix(cfg.hidden_size, args.rotate_mode)
        head_dim = cfg.hidden_size // cfg.num_attention_heads
        rotate_embeddings(model, Q, is_minicpmv=True)
        for layer in tqdm.tqdm(model.llm.model.layers, unit="layer", desc="Rotating"):
            rotate_attention_inputs(layer, Q, is_minicpmv=True)
            rotate_attention_output(layer, Q)
            rotate_mlp_input(layer, Q, is_minicpmv=True)
            rotate_mlp_output(layer, Q, args.online_llm_hadamard)
            rotate_ov_proj(layer, cfg.num_attention_heads, head_dim, is_minicpmv=True)
        rotate_head(model, Q, is_minicpmv=True)
