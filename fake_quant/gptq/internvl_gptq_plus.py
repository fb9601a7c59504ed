"""Weight pass for InternVL2 (reference: ``fake_quant/gptq/internvl_gptq_plus.py``).  The
returned keys use the module path (``model.vision_model.encoder.layers.<i>.<name>``,
``model.mlp1.<name>``, ``model.language_model.model.layers.<i>.<name>``)."""
import logging

import torch

from . import sequential as seq
from .rtn import rtn_module, rtn_wrapped_conv


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


def _feed(model, dataset, dataset_name, args):
    return lambda enough: seq.run_calibration_prompts(model, dataset, dataset_name, args, enough)


@torch.no_grad()
def gptq_internvl_fwrd_visual_clip_conv1(model, dataset, dev, dataset_name, args, quantizers):
    emb = model.model.vision_model.embeddings
    target = emb.patch_embedding                 # the wrapped Conv2d itself
    samples = seq.capture_inputs(target, _feed(model, dataset, dataset_name, args), args.nsamples)
    seq.gptq_single(lambda s: target(*s[0], **s[1]), target, samples, [["module"]],
                    args.visual_w_bits, not args.w_asym, args.visual_w_clip, args,
                    lambda n: "model.vision_model.embeddings.patch_embedding", quantizers,
                    layers=(torch.nn.Conv2d,))


@torch.no_grad()
def gptq_internvl_fwrd_visual_clip_resblocks(model, dataset, dev, dataset_name, args, quantizers):
    blocks = model.model.vision_model.encoder.layers
    samples = seq.capture_inputs(blocks[0], _feed(model, dataset, dataset_name, args), args.nsamples)
    sequential = [["attn.qkv.module"], ["attn.proj.module"], ["mlp.fc1.module"],
                  ["mlp.fc2.L2" if args.visual_split else "mlp.fc2.module"]]
    seq.gptq_blocks(blocks, samples, sequential, args.visual_w_bits, not args.w_asym, args.visual_w_clip,
                    args, "model.vision_model.encoder.layers.%d.%s", quantizers)


@torch.no_grad()
def gptq_internvl_fwrd_visual_clip_cross_attention(model, dataset, dev, dataset_name, args, quantizers):
    print("-----GPTQ Quantization visual clip cross attention-----")
    mlp1 = model.model.mlp1
    samples = seq.capture_inputs(mlp1, _feed(model, dataset, dataset_name, args), args.nsamples)
    seq.gptq_single(lambda s: mlp1(*s[0], **s[1]), mlp1, samples, [["1.module"], ["3.module"]],
                    args.visual_w_bits, not args.w_asym, args.visual_w_clip, args,
                    lambda n: "model.mlp1." + n, quantizers)


@torch.no_grad()
def gptq_internvl_fwrd_llm(model, dataset, dev, dataset_name, args, quantizers):
    print("-----GPTQ Quantization LLM-----")
    layers = model.model.language_model.model.layers
    samples = seq.capture_inputs(layers[0], _feed(model, dataset, dataset_name, args), args.nsamples)
    sequential = [["attention.wqkv.module"], ["attention.wo.module"],
                  ["feed_forward.w1.module", "feed_forward.w3.module"],
                  ["feed_forward.w2.L2" if args.llm_split else "feed_forward.w2.module"]]
    seq.gptq_blocks(layers, samples, sequential, args.llm_w_bits, not args.w_asym, args.llm_w_clip, args,
                    "model.language_model.model.layers.%d.%s", quantizers)
    return quantizers


@torch.no_grad()
def internvl_rtn_gptq_fwrd_plus(model, dataset, dev, dataset_name, args):
    logging.info("-----RTN Or GPTQ Quantization-----")
    quantizers = {}
    if args.quant_visual_clip:
        if args.visual_w_rtn:
            internvl_visual_clip_rtn(model.model, dev, args, quantizers)
        else:
            gptq_internvl_fwrd_visual_clip_conv1(model, dataset, dev, dataset_name, args, quantizers)
            gptq_internvl_fwrd_visual_clip_resblocks(model, dataset, dev, dataset_name, args, quantizers)
    if args.quant_cross_attention:
        if args.visual_w_rtn:
            internvl_visual_cross_attention_rtn(model.model, dev, args, quantizers)
        else:
            gptq_internvl_fwrd_visual_clip_cross_attention(model, dataset, dev, dataset_name, args, quantizers)
    if args.quant_llm:
        if args.llm_w_rtn:
            internvl_llm_rtn(model.model, dev, args, quantizers)
        else:
            gptq_internvl_fwrd_llm(model, dataset, dev, dataset_name, args, quantizers)
    return quantizers
