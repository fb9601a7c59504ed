"""Weight pass for Qwen2-VL (reference: ``fake_quant/gptq/qwen2vl_gptq_plus.py``).

Keys of the returned dict follow upstream: ``model.visual.patch_embed.proj.module``,
``model.visual.blocks.<i>.<name>``, ``model.visual.merger.<name>``,
``model.model.layers.<i>.<name>``.
"""
import logging

import torch

from .rtn import rtn_module, rtn_wrapped_conv

_GPTQ_MSG = ("layer-sequential GPTQ capture for %s is not built yet in this tree; pass "
             "--visual_w_rtn / --llm_w_rtn (RTN, optional MSE clipping) or quantize with "
             "fake_quant.gptq.gptq_utils.GPTQ directly")


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


@torch.no_grad()
def qwen2vl_rtn_gptq_fwrd_plus(model, dataset, dev, dataset_name, args):
    logging.info("-----RTN Or GPTQ Quantization-----")
    quantizers = {}
    if args.quant_visual_clip:
        if not args.visual_w_rtn:
            raise NotImplementedError(_GPTQ_MSG % "the vision tower")
        qwen2vl_visual_clip_rtn(model.model, dev, args, quantizers)
    if args.quant_cross_attention:
        if not args.visual_w_rtn:
            raise NotImplementedError(_GPTQ_MSG % "the merger")
        qwen2vl_visual_cross_attention_rtn(model.model, dev, args, quantizers)
    if args.quant_llm:
        if not args.llm_w_rtn:
            raise NotImplementedError(_GPTQ_MSG % "the LLM")
        qwen2vl_llm_rtn(model.model, dev, args, quantizers)
    return quantizers
