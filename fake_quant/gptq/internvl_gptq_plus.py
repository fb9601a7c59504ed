"""Weight pass for InternVL2 (reference: ``fake_quant/gptq/internvl_gptq_plus.py``).  The
returned keys use the module path (``model.vision_model.encoder.layers.<i>.<name>``,
``model.mlp1.<name>``, ``model.language_model.model.layers.<i>.<name>``)."""
import logging

import torch

from .rtn import rtn_module, rtn_wrapped_conv
from .qwen2vl_gptq_plus import _GPTQ_MSG


def internvl_visual_clip_rtn(model, dev, args, quantizers):
    sym, mse = not args.w_asym, args.visual_w_clip
    rtn_wrapped_conv(model.vision_model.embeddings.patch_embedding,
                     "model.vision_model.embeddings.patch_embedding", args.visual_w_bits, sym, mse,
                     quantizers)
    for i, layer in enumerate(model.vision_model.encoder.layers):
        rtn_module(layer, f"model.vision_model.encoder.layers.{i}", args.visual_w_bits, sym, mse,
                   args.skip_names, quantizers)


def internvl_visual_cross_attention_rtn(model, dev, args, quantizers):
    rtn_module(model.mlp1, "model.mlp1", args.visual_w_bits, not args.w_asym, args.visual_w_clip,
               [], quantizers)


def internvl_llm_rtn(model, dev, args, quantizers):
    for i, layer in enumerate(model.language_model.model.layers):
        rtn_module(layer, f"model.language_model.model.layers.{i}", args.llm_w_bits,
                   not args.w_asym, args.llm_w_clip, args.skip_names, quantizers)


@torch.no_grad()
def internvl_rtn_gptq_fwrd_plus(model, dataset, dev, dataset_name, args):
    logging.info("-----RTN Or GPTQ Quantization-----")
    quantizers = {}
    if args.quant_visual_clip:
        if not args.visual_w_rtn:
            raise NotImplementedError(_GPTQ_MSG % "the vision tower")
        internvl_visual_clip_rtn(model.model, dev, args, quantizers)
    if args.quant_cross_attention:
        if not args.visual_w_rtn:
            raise NotImplementedError(_GPTQ_MSG % "mlp1")
        internvl_visual_cross_attention_rtn(model.model, dev, args, quantizers)
    if args.quant_llm:
        if not args.llm_w_rtn:
            raise NotImplementedError(_GPTQ_MSG % "the LLM")
        internvl_llm_rtn(model.model, dev, args, quantizers)
    return quantizers
