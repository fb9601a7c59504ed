"""Weight pass for MiniCPM-V (reference ``fake_quant/gptq/minicpmv_gptq_plus.py``): RTN or
layer-sequential GPTQ over the SigLIP tower (``vpm``), the resampler and the LLM."""
import logging

import torch

from . import sequential as seq
from .rtn import rtn_module, rtn_wrapped_conv


def _feed(model, dataset, args):
    return lambda enough: seq.run_calibration_prompts(model, dataset, args.dataset_name, args, enough)


def minicpmv_visual_clip_rtn(model, dev, args, quantizers):
    sym, mse = not args.w_asym, args.visual_w_clip
    rtn_wrapped_conv(model.vpm.embeddings.patch_embedding, "model.vpm.embeddings.patch_embedding",
                     args.visual_w_bits, sym, mse, quantizers)
    for i, layer in enumerate(model.vpm.encoder.layers):
        rtn_module(layer, f"model.vpm.encoder.layers.{i}", args.visual_w_bits, sym, mse, args.skip_names, quantizers)


def minicpmv_visual_cross_attention_rtn(model, dev, args, quantizers):
    print("-----Rtn Quantization visual clip cross attention-----")
    rtn_module(model.resampler, "model.resampler", args.visual_w_bits, not args.w_asym, args.visual_w_clip, [],
               quantizers)


def minicpmv_llm_rtn(model, dev, args, quantizers):
    for i, layer in enumerate(model.llm.model.layers):
        rtn_module(layer, f"model.llm.model.layers.{i}", args.llm_w_bits, not args.w_asym, args.llm_w_clip,
                   args.skip_names, quantizers)


@torch.no_grad()
def gptq_minicpmv_fwrd_visual_clip_conv1(model, dataset, dev, args, quantizers):
    target = model.model.vpm.embeddings.patch_embedding
    samples = seq.capture_inputs(target, _feed(model, dataset, args), args.nsamples)
    seq.gptq_single(lambda s: target(*s[0], **s[1]), target, samples, [["module"]], args.visual_w_bits,
                    not args.w_asym, args.visual_w_clip, args, lambda n: "model.vpm.embeddings.patch_embedding",
                    quantizers, layers=(torch.nn.Conv2d,))


@torch.no_grad()
def gptq_minicpmv_fwrd_visual_clip_resblocks(model, dataset, dev, args, quantizers):
    blocks = model.model.vpm.encoder.layers
    samples = seq.capture_inputs(blocks[0], _feed(model, dataset, args), args.nsamples)
    sequential = [["self_attn.k_proj.module", "self_attn.v_proj.module", "self_attn.q_proj.module"],
                  ["self_attn.out_proj.module"], ["mlp.fc1.module"],
                  ["mlp.fc2.L2" if args.visual_split else "mlp.fc2.module"]]
    seq.gptq_blocks(blocks, samples, sequential, args.visual_w_bits, not args.w_asym, args.visual_w_clip, args,
                    "model.vpm.encoder.layers.%d.%s", quantizers)


@torch.no_grad()
def gptq_minicpmv_fwrd_visual_clip_cross_attention(model, dataset, dev, args, quantizers):
    print("-----GPTQ Quantization visual clip cross attention-----")
    res = model.model.resampler
    samples = seq.capture_inputs(res, _feed(model, dataset, args), args.nsamples)
    sequential = [["kv_proj.module"], ["attn.k_proj.module", "attn.v_proj.module", "attn.q_proj.module"],
                  ["attn.out_proj.module"], ["proj_fc.module"]]
    seq.gptq_single(lambda s: res(*s[0], **s[1]), res, samples, sequential, args.visual_w_bits, not args.w_asym,
                    args.visual_w_clip, args, lambda n: "model.resampler." + n, quantizers)


@torch.no_grad()
def gptq_minicpmv_fwrd_llm(model, dataset, dev, args, quantizers):
    print("-----GPTQ Quantization LLM-----")
    layers = model.model.llm.model.layers
    samples = seq.capture_inputs(layers[0], _feed(model, dataset, args), args.nsamples)
    sequential = [["self_attn.k_proj.module", "self_attn.v_proj.module", "self_attn.q_proj.module"],
                  ["self_attn.o_proj.module"], ["mlp.up_proj.module", "mlp.gate_proj.module"],
                  ["mlp.down_proj.L2" if args.llm_split else "mlp.down_proj.module"]]
    seq.gptq_blocks(layers, samples, sequential, args.llm_w_bits, not args.w_asym, args.llm_w_clip, args,
                    "model.llm.model.layers.%d.%s", quantizers)
    return quantizers


@torch.no_grad()
def minicpmv_rtn_gptq_fwrd_plus(model, dataset, dev, dataset_name, args):
    logging.info("-----RTN Or GPTQ Quantization-----")
    quantizers = {}
    if args.quant_visual_clip:
        if args.visual_w_rtn:
            minicpmv_visual_clip_rtn(model.model, dev, args, quantizers)
        else:
            gptq_minicpmv_fwrd_visual_clip_conv1(model, dataset, dev, args, quantizers)
            gptq_minicpmv_fwrd_visual_clip_resblocks(model, dataset, dev, args, quantizers)
    if args.quant_cross_attention:
        if args.visual_w_rtn:
            minicpmv_visual_cross_attention_rtn(model.model, dev, args, quantizers)
        else:
            gptq_minicpmv_fwrd_visual_clip_cross_attention(model, dataset, dev, args, quantizers)
    if args.quant_llm:
        if args.llm_w_rtn:
            minicpmv_llm_rtn(model.model, dev, args, quantizers)
        else:
            gptq_minicpmv_fwrd_llm(model, dataset, dev, args, quantizers)
    return quantizers
