"""A tiny random-config Qwen2-VL built from the INSTALLED ``transformers`` (real HF module classes, no checkpoint) and the
reference driver's sequence of calls on it (exam/quant_qwen2vl.py:29-222), shared by the CPU and GPU end-to-end tests.

Sizes keep the structure of the 7B model: LLM hidden 224 = 28 x 8 (7B: 3584 = 28 x 128), head_dim 32, intermediate 592 -> padded to
624 = 156 x 4 by ``auto_pad_size`` (7B: 18944 -> 19968 = 156 x 128, the K = 156 online Hadamard + pad hook); ViT width 80 = 20 x 4
(7B: 1280 = 20 x 64), 2 heads of 40, fc2 input 320 = 40 x 8 (7B: 5120 = 40 x 128, the K = 40 online Hadamard + split column)."""
import functools
import types

import torch

IMG, VID, VSTART, VEND = 300, 301, 302, 303


def build(dtype=torch.float32, device="cpu", seed=0):
    from transformers import Qwen2VLConfig, Qwen2VLForConditionalGeneration
    torch.manual_seed(seed)
    cfg = Qwen2VLConfig(
        text_config=dict(hidden_size=224, intermediate_size=592, num_hidden_layers=2, num_attention_heads=7, num_key_value_heads=1,
                         vocab_size=320, max_position_embeddings=256, rope_scaling={"type": "mrope", "mrope_section": [4, 6, 6]},
                         bos_token_id=1, eos_token_id=2, pad_token_id=0),
        vision_config=dict(depth=2, embed_dim=80, hidden_size=224, num_heads=2, mlp_ratio=4, patch_size=14, spatial_merge_size=2,
                           temporal_patch_size=2, in_channels=3),
        image_token_id=IMG, video_token_id=VID, vision_start_token_id=VSTART, vision_end_token_id=VEND)
    m = Qwen2VLForConditionalGeneration(cfg).eval()
    g = torch.Generator().manual_seed(seed + 1)
    for _, p in m.named_parameters():          # non-trivial norm weights / biases: the LayerNorm fusion has something to fold
        if p.dim() == 1:
            p.data = p.data + 0.1 * torch.randn(p.shape, generator=g)
    return m.to(device=device, dtype=dtype)


def inputs(device="cpu", dtype=torch.float32, grid=(1, 8, 8), n_text=24, seed=2):
    """One image of grid[1] x grid[2] patches (a quarter as many merged vision tokens) followed by n_text text tokens."""
    t, h, w = grid
    n_patch = t * h * w
    g = torch.Generator().manual_seed(seed)
    pix = torch.randn((n_patch, 3 * 2 * 14 * 14), generator=g)
    ids = torch.cat([torch.tensor([VSTART]), torch.full((n_patch // 4,), IMG), torch.tensor([VEND]),
                     torch.randint(3, 290, (n_text,), generator=g)])[None]
    out = dict(input_ids=ids, attention_mask=torch.ones_like(ids), pixel_values=pix.to(dtype),
               image_grid_thw=torch.tensor([list(grid)]), mm_token_type_ids=(ids == IMG).int())
    return {k: v.to(device) for k, v in out.items()}


def driver_args(**over):
    """The flags of the reference's canonical W4A8 + W4A8 command (docs/qwen2vl.md:26) that the passes read, RTN instead of GPTQ."""
    a = types.SimpleNamespace(
        no_fuse_visual_clip=False, no_fuse_visual_cross_attn=False, no_fuse_llm=False,
        rotate_visual_clip=True, rotate_visual_cross_attn=True, rotate_llm=True, rotate_mode="hadamard",
        online_llm_hadamard=True, online_visual_hadamard=True, fp32_had=False,
        quant_llm=True, quant_visual_clip=True, quant_cross_attention=True, act_per_tensor=False,
        visual_w_bits=4, llm_w_bits=4, visual_a_bits=8, llm_a_bits=8, w_asym=False, a_asym=False, a_groupsize=-1, a_clip_ratio=1.0,
        visual_w_clip=False, llm_w_clip=False, visual_w_rtn=True, llm_w_rtn=True, visual_static=True, llm_static=True,
        visual_split=True, llm_split=False, skip_names=[], no_sibling_fusion=False)
    for k, v in over.items():
        setattr(a, k, v)
    return a


def rotate_and_wrap(vlm, args):
    """exam/quant_qwen2vl.py:52-143: fuse -> rotate -> add_act_qaunt -> online-Hadamard / split / pad-hook flags.
    ``vlm.model`` is the HF module in the layout the reference walks (``fake_quant.hf_compat.legacy_qwen2vl``)."""
    from fake_quant import hadamard_utils, quant_utils, utils
    from fake_quant.qwen2vl_rotation import fuse_qwen2vl_layer_norms, rotate_qwen2vl_model
    utils.seed_everything(42)
    fuse_qwen2vl_layer_norms(vlm, args)
    rotate_qwen2vl_model(vlm.model, args)
    quant_utils.qwen2vl_add_act_qaunt(vlm, args)
    ql = quant_utils.find_qlayers(vlm.model.model, layers=[quant_utils.ActQuantWrapper])
    for name in ql:
        if "mlp.down_proj" in name:
            had_K, K = hadamard_utils.get_hadK(vlm.model.config.intermediate_size)
            ql[name].online_full_had, ql[name].had_K, ql[name].K, ql[name].fp32_had = True, had_K, K, args.fp32_had
            ql[name].split = args.llm_split
            if args.llm_split:
                ql[name].split_weights()
            if vlm.model.config.need_pad:
                ql[name].register_forward_pre_hook(functools.partial(utils.revise_down_input,
                                                                     new_size=vlm.model.config.intermediate_size))
    qv = quant_utils.find_qlayers(vlm.model.visual, layers=[quant_utils.ActQuantWrapper])
    for name in qv:
        if "mlp.fc2" in name:
            had_K, K = hadamard_utils.get_hadK(int(vlm.model.visual.blocks[0].mlp.fc2.module.in_features))
            qv[name].online_full_had, qv[name].had_K, qv[name].K, qv[name].fp32_had = True, had_K, K, args.fp32_had
            qv[name].split = args.visual_split
            if args.visual_split:
                qv[name].split_weights()
    return ql, qv


def quantize_and_calibrate(vlm, hf, args, ql, qv, batches):
    """exam/quant_qwen2vl.py:146-218: weight pass, activation quantizers, then the calibration protocol
    (fake_quant/quant_utils.py calib_qwen2vl_plus: open -> forwards -> last -> close -> model_quant) on ``batches``."""
    from fake_quant import quant_utils
    from fake_quant.gptq import qwen2vl_gptq_plus as gq
    quantizers = gq.qwen2vl_rtn_gptq_fwrd_plus(vlm, None, next(hf.parameters()).device, "none", args)
    for grp, bits, static in ((qv, args.visual_a_bits, args.visual_static), (ql, args.llm_a_bits, args.llm_static)):
        for name in grp:
            grp[name].quantizer.configure(bits=bits, groupsize=args.a_groupsize, sym=not args.a_asym, clip_ratio=args.a_clip_ratio,
                                          act_per_tensor=args.act_per_tensor, static=static, observer_type="minmax")
    quant_utils.model_open_calibrate(vlm.model, args)
    with torch.no_grad():
        for i, b in enumerate(batches):
            if i == len(batches) - 1:
                quant_utils.model_open_last_calibrate(vlm.model, args)
            hf(**b)
    quant_utils.model_close_calibrate(vlm.model, args)
    quant_utils.model_quant(vlm.model, args)
    return quantizers
