"""Weight pass for Qwen-VL (v1, "opt" layout; reference ``fake_quant/gptq/qwenvl_gptq_plus.py``):
RTN or layer-sequential GPTQ.  Dict keys follow upstream, including its ``model.model.`` prefix on
the attn_pool / proj_fc entries (:396,441).  Upstream's RTN functions return nothing; here they
also record (and attach) their quantizers, which the real W4A8 path needs."""
import logging

import torch

from . import sequential as seq
from .rtn import rtn_module, rtn_wrapped_conv

_CONV = (torch.nn.Conv2d, torch.nn.Conv3d, torch.nn.Linear)


def _feed(model, dataset, args):
    return lambda enough: seq.run_calibration_prompts(model, dataset, args.dataset_name, args, enough)


def qwenvl_visual_clip_rtn(model, dev, args, quantizers=None):
    quantizers = {} if quantizers is None else quantizers
    sym, mse = not args.w_asym, args.visual_w_clip
    vis = model.transformer.visual
    rtn_wrapped_conv(vis.conv1, "model.transformer.visual.conv1", args.visual_w_bits, sym, mse, quantizers)
    for i, layer in enumerate(vis.transformer.resblocks):
        rtn_module(layer, f"model.transformer.visual.transformer.resblocks.{i}", args.visual_w_bits, sym, mse,
                   args.skip_names, quantizers)
    return quantizers


def qwenvl_visual_cross_attention_rtn(model, dev, args, quantizers=None):
    quantizers = {} if quantizers is None else quantizers
    print("-----Rtn Quantization visual clip cross attention-----")
    vis = model.transformer.visual
    sym, mse = not args.w_asym, args.visual_w_clip
    rtn_module(vis.attn_pool, "model.model.transformer.visual.attn_pool", args.visual_w_bits, sym, mse, [], quantizers)
    rtn_wrapped_conv(vis.proj_fc, "model.model.transformer.visual.proj_fc", args.visual_w_bits, sym, mse, quantizers)
    return quantizers


def qwenvl_llm_rtn(model, dev, args, quantizers=None):
    quantizers = {} if quantizers is None else quantizers
    for i, layer in enumerate(model.transformer.h):
        rtn_module(layer, f"model.transformer.h.{i}", args.llm_w_bits, not args.w_asym, args.llm_w_clip,
                   args.skip_names, quantizers)
    return quantizers


@torch.no_grad()
def gptq_fwrd_visual_clip_conv1(model, dataset, dev, args, quantizers):
    print("-----GPTQ Quantization visual clip conv1-----")
    conv1 = model.model.transformer.visual.conv1          # the ActQuantWrapper around the patch conv
    samples = seq.capture_inputs(conv1, _feed(model, dataset, args), args.nsamples)
    seq.gptq_single(lambda s: conv1(*s[0], **s[1]), conv1, samples, [["module"]], args.visual_w_bits,
                    not args.w_asym, args.visual_w_clip, args, lambda n: "model.transformer.visual.conv1",
                    quantizers, layers=_CONV)


@torch.no_grad()
def gptq_fwrd_visual_clip_resblocks(model, dataset, dev, args, quantizers):
    print("-----GPTQ Quantization visual clip resblocks-----")
    blocks = model.model.transformer.visual.transformer.resblocks
    samples = seq.capture_inputs(blocks[0], _feed(model, dataset, args), args.nsamples)
    sequential = [["attn.k_proj.module", "attn.v_proj.module", "attn.q_proj.module"], ["attn.out_proj.module"],
                  ["mlp.c_fc.module"], ["mlp.c_proj.L2" if args.visual_split else "mlp.c_proj.module"]]
    seq.gptq_blocks(blocks, samples, sequential, args.visual_w_bits, not args.w_asym, args.visual_w_clip, args,
                    "model.transformer.visual.transformer.resblocks.%d.%s", quantizers)


@torch.no_grad()
def gptq_fwrd_visual_clip_cross_attention(model, dataset, dev, args, quantizers):
    print("-----GPTQ Quantization visual clip cross attention-----")
    vis = model.model.transformer.visual
    pool = vis.attn_pool
    samples = seq.capture_inputs(pool, _feed(model, dataset, args), args.nsamples)
    sequential = [["kv_proj.module"], ["attn.k_proj.module", "attn.v_proj.module", "attn.q_proj.module"],
                  ["attn.out_proj.module"]]
    sym, mse = not args.w_asym, args.visual_w_clip
    seq.gptq_single(lambda s: pool(*s[0], **s[1]), pool, samples, sequential, args.visual_w_bits, sym, mse, args,
                    lambda n: "model.model.transformer.visual.attn_pool." + n, quantizers)
    # proj_fc sees ln_post(attn_pool(x)) of the already quantized pool (:404-441)
    fc_in = [((vis.ln_post(pool(*s[0], **s[1])),), {}) for s in samples]
    seq.gptq_single(lambda s: vis.proj_fc(*s[0]), vis.proj_fc, fc_in, [["module"]], args.visual_w_bits, sym, mse,
                    args, lambda n: "model.model.transformer.visual.proj_fc", quantizers)


@torch.no_grad()
def gptq_fwrd_llm(model, dataset, dev, args, quantizers):
    print("-----GPTQ Quantization LLM-----\\n")
    layers = model.model.transformer.h
    samples = seq.capture_inputs(layers[0], _feed(model, dataset, args), args.nsamples)
    sequential = [["attn.k_proj.module", "attn.v_proj.module", "attn.q_proj.module"], ["attn.c_proj.module"],
                  ["mlp.w1.module", "mlp.w2.module"], ["mlp.c_proj.L2" if args.llm_split else "mlp.c_proj.module"]]
    seq.gptq_blocks(layers, samples, sequential, args.llm_w_bits, not args.w_asym, args.llm_w_clip, args,
                    "model.transformer.h.%d.%s", quantizers)
    return quantizers


@torch.no_grad()
def qwenvl_rtn_gptq_fwrd_plus(model, dataset, dev, args):
    logging.info("-----RTN Or GPTQ Quantization-----")
    quantizers = {}
    if args.quant_visual_clip:
        if args.visual_w_rtn:
            qwenvl_visual_clip_rtn(model.model, dev, args, quantizers)
        else:
            gptq_fwrd_visual_clip_conv1(model, dataset, dev, args, quantizers)
            gptq_fwrd_visual_clip_resblocks(model, dataset, dev, args, quantizers)
    if args.quant_cross_attention:
        if args.visual_w_rtn:
            qwenvl_visual_cross_attention_rtn(model.model, dev, args, quantizers)
        else:
            gptq_fwrd_visual_clip_cross_attention(model, dataset, dev, args, quantizers)
    if args.quant_llm:
        if args.llm_w_rtn:
            qwenvl_llm_rtn(model.model, dev, args, quantizers)
        else:
            gptq_fwrd_llm(model, dataset, dev, args, quantizers)
    return quantizers
