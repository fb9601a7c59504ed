"""Weight pass for Qwen2-VL (reference: ``fake_quant/gptq/qwen2vl_gptq_plus.py``): RTN
(``--visual_w_rtn`` / ``--llm_w_rtn``) or layer-sequential GPTQ over calibration prompts.

Keys of the returned dict follow upstream: ``model.visual.patch_embed.proj.module``,
``model.visual.blocks.<i>.<name>``, ``model.visual.merger.<name>``,
``model.model.layers.<i>.<name>``.
"""
import logging

import torch

from .rtn import rtn_module, rtn_wrapped_conv

from . import sequential as seq


def qwen2vl_visual_clip_rtn(model, dev, args, quantizers):
    sym, mse = not args.w_asym, args.visual_w_clip
    rtn_wrapped_conv(model.visual.patch_embed.proj, "model.visual.patch_embed.proj.module",
                     args.visual_w_bits, sym, mse, quantizers)
    for i, layer in enumerate(model.visual.blocks):
        rtn_module(layer, f"model.visual.blocks.{i}", args.visual_w_bits, sym, mse,
                   args.skip_names, quantizers)


def qwen2vl_visual_cross_attention_rtn(model, dev, args, quantizers):
    print("-----Rtn Quantization visual clip cross attention-----")
    rtn_module(model.visual.merger, "model.visual.merger", args.visual_w_bits, not args.w_asym,
               args.visual_w_clip, [], quantizers)


def qwen2vl_llm_rtn(model, dev, args, quantizers):
    print("-----Rtn Quantization llm---")
    for i, layer in enumerate(model.model.layers):
        rtn_module(layer, f"model.model.layers.{i}", args.llm_w_bits, not args.w_asym,
                   args.llm_w_clip, args.skip_names, quantizers)


def _feed(model, dataset, dataset_name, args):
    return lambda enough: seq.run_calibration_prompts(model, dataset, dataset_name, args, enough)


@torch.no_grad()
def gptq_qwen2vl_fwrd_visual_clip_conv1(model, dataset, dev, dataset_name, args, quantizers):
    """The patch-embedding Conv3d (whole-patch kernel), solved as a Linear over flattened patches."""
    patch_embed = model.model.visual.patch_embed
    samples = seq.capture_inputs(patch_embed, _feed(model, dataset, dataset_name, args), args.nsamples)
    seq.gptq_single(lambda s: patch_embed(*s[0], **s[1]), patch_embed, samples, [["proj.module"]],
                    args.visual_w_bits, not args.w_asym, args.visual_w_clip, args,
                    lambda n: "model.visual.patch_embed." + n, quantizers,
                    layers=(torch.nn.Conv3d, torch.nn.Conv2d))
    print("-----GPTQ Quantization visual clip conv1 Done-----")


@torch.no_grad()
def gptq_qwen2vl_fwrd_visual_clip_resblocks(model, dataset, dev, dataset_name, args, quantizers):
    blocks = model.model.visual.blocks
    samples = seq.capture_inputs(blocks[0], _feed(model, dataset, dataset_name, args), args.nsamples)
    sequential = [["attn.qkv.module"], ["attn.proj.module"], ["mlp.fc1.module"],
                  ["mlp.fc2.L2" if args.visual_split else "mlp.fc2.module"]]
    seq.gptq_blocks(blocks, samples, sequential, args.visual_w_bits, not args.w_asym, args.visual_w_clip,
                    args, "model.visual.blocks.%d.%s", quantizers)
    print("\n-----GPTQ Quantization visual clip resblocks Done-----")


@torch.no_grad()
def gptq_qwen2vl_fwrd_visual_clip_cross_attention(model, dataset, dev, dataset_name, args, quantizers):
    print("-----GPTQ Quantization visual clip cross attention-----")
    merger = model.model.visual.merger
    samples = seq.capture_inputs(merger, _feed(model, dataset, dataset_name, args), args.nsamples)
    seq.gptq_single(lambda s: merger(*s[0], **s[1]), merger, samples, [["mlp.0.module"], ["mlp.2.module"]],
                    args.visual_w_bits, not args.w_asym, args.visual_w_clip, args,
                    lambda n: "model.visual.merger." + n, quantizers)


@torch.no_grad()
def gptq_qwen2vl_fwrd_llm(model, dataset, dev, dataset_name, args, quantizers):
    print("-----GPTQ Quantization LLM-----")
    cfg = model.model.config
    use_cache = getattr(cfg, "use_cache", None)
    cfg.use_cache = False
    layers = model.model.model.layers
    samples = seq.capture_inputs(layers[0], _feed(model, dataset, dataset_name, args), args.nsamples)
    sequential = [["self_attn.q_proj.module", "self_attn.k_proj.module", "self_attn.v_proj.module"],
                  ["self_attn.o_proj.module"],
                  ["mlp.up_proj.module", "mlp.gate_proj.module"],
                  ["mlp.down_proj.L2" if args.llm_split else "mlp.down_proj.module"]]
    seq.gptq_blocks(layers, samples, sequential, args.llm_w_bits, not args.w_asym, args.llm_w_clip, args,
                    "model.model.layers.%d.%s", quantizers)
    cfg.use_cache = use_cache
    print("\n-----GPTQ Quantization LLM Done-----")
    return quantizers


@torch.no_grad()
def qwen2vl_rtn_gptq_fwrd_plus(model, dataset, dev, dataset_name, args):
    logging.info("-----RTN Or GPTQ Quantization-----")
    quantizers = {}
    if args.quant_visual_clip:
        if args.visual_w_rtn:
            qwen2vl_visual_clip_rtn(model.model, dev, args, quantizers)
        else:
            gptq_qwen2vl_fwrd_visual_clip_conv1(model, dataset, dev, dataset_name, args, quantizers)
            gptq_qwen2vl_fwrd_visual_clip_resblocks(model, dataset, dev, dataset_name, args, quantizers)
    if args.quant_cross_attention:
        if args.visual_w_rtn:
            qwen2vl_visual_cross_attention_rtn(model.model, dev, args, quantizers)
        else:
            gptq_qwen2vl_fwrd_visual_clip_cross_attention(model, dataset, dev, dataset_name, args, quantizers)
    if args.quant_llm:
        if args.llm_w_rtn:
            qwen2vl_llm_rtn(model.model, dev, args, quantizers)
        else:
            gptq_qwen2vl_fwrd_llm(model, dataset, dev, dataset_name, args, quantizers)
    return quantizers
