"""Synthetic Qwen2-VL-7B prefill workload for the hot path.

1 x 448^2 image -> 1024 vision tokens -> 256 merged tokens, + 512 text tokens => M_llm = 768.
Dimensions are the public HF config; weights are random with the real
shapes, W4 RTN (symmetric, per output channel); activation scales come from the min/max
observer kernels run on the synthetic inputs.  No checkpoint and no dataset are involved.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional

import numpy as np
import torch

from . import ops
from .engine import WORKSPACE, HadamardSpec, W4A8Linear


@dataclass
class LinearSpec:
    name: str
    M: int
    k_in: int           # features fed by the caller
    k: int              # reduction dim of the weight (k_in padded to a Hadamard-able size)
    n: int
    count: int          # instances per prefill
    bias: bool = False
    had_K: int = 0      # 0 = no online Hadamard, else the special factor (1 = pure 2^p)
    split: bool = False
    msq: bool = False   # two activation scale sets selected by the token-type mask
    group: str = ""     # Linears sharing one input tensor (quantized once)


M_VIS, M_MERGED, M_TXT = 1024, 256, 512
M_LLM = M_MERGED + M_TXT


def qwen2vl_7b_specs(msq: bool = True, batch: int = 1) -> List[LinearSpec]:
    """``batch`` image+prompt samples stacked along the row dimension of every Linear (the
    benchmark configuration is batch = 1; larger values show how the GEMMs scale with M)."""
    v, l = 32, 28
    specs = _qwen2vl_7b_specs(msq, v, l)
    for sp in specs:
        sp.M *= batch
    return specs


def _qwen2vl_7b_specs(msq: bool, v: int, l: int) -> List[LinearSpec]:
    return [
        LinearSpec("vis.patch_embed", M_VIS, 1176, 1176, 1280, 1),
        LinearSpec("vis.attn.qkv", M_VIS, 1280, 1280, 3840, v, bias=True),
        LinearSpec("vis.attn.proj", M_VIS, 1280, 1280, 1280, v, bias=True),
        LinearSpec("vis.mlp.fc1", M_VIS, 1280, 1280, 5120, v, bias=True),
        LinearSpec("vis.mlp.fc2", M_VIS, 5120, 5120, 1280, v, bias=True, had_K=40, split=True),
        LinearSpec("merger.mlp.0", M_MERGED, 5120, 5120, 5120, 1, bias=True),
        LinearSpec("merger.mlp.2", M_MERGED, 5120, 5120, 3584, 1, bias=True),
        LinearSpec("llm.q_proj", M_LLM, 3584, 3584, 3584, l, bias=True, msq=msq, group="qkv"),
        LinearSpec("llm.k_proj", M_LLM, 3584, 3584, 512, l, bias=True, msq=msq, group="qkv"),
        LinearSpec("llm.v_proj", M_LLM, 3584, 3584, 512, l, bias=True, msq=msq, group="qkv"),
        LinearSpec("llm.o_proj", M_LLM, 3584, 3584, 3584, l, msq=msq),
        LinearSpec("llm.gate_proj", M_LLM, 3584, 3584, 18944, l, msq=msq, group="gate_up"),
        LinearSpec("llm.up_proj", M_LLM, 3584, 3584, 18944, l, msq=msq, group="gate_up"),
        LinearSpec("llm.down_proj", M_LLM, 18944, 19968, 3584, l, had_K=156, msq=msq),
    ]


def qwenvl_7b_specs(batch: int = 1) -> List[LinearSpec]:
    """BASELINE config 2: Qwen-VL-7B (v1), one 448^2 image (1024 patches -> 256 resampled tokens) + 512 text
    tokens.  Public config: LLM hidden 4096, 32 layers, ff 11008 (reference model/modeling_qwen_opt.py:374-380:
    w1 / w2 read the block input, c_proj the gated product, online Hadamard 172 x 64); ViT width 1664, 48 layers,
    mlp 8192 (pure power-of-two Hadamard), output 4096.  No modality-specific scales in this configuration."""
    v, l, mv, ml = 48, 32, M_VIS * batch, M_LLM * batch
    return [
        LinearSpec("vis.conv1", mv, 588, 588, 1664, 1),
        LinearSpec("vis.attn.in_proj", mv, 1664, 1664, 4992, v, bias=True),
        LinearSpec("vis.attn.out_proj", mv, 1664, 1664, 1664, v, bias=True),
        LinearSpec("vis.mlp.c_fc", mv, 1664, 1664, 8192, v, bias=True),
        LinearSpec("vis.mlp.c_proj", mv, 8192, 8192, 1664, v, bias=True, had_K=1),
        LinearSpec("pool.kv_proj", mv, 1664, 1664, 4096, 1),
        LinearSpec("llm.attn.c_attn", ml, 4096, 4096, 12288, l, bias=True),
        LinearSpec("llm.attn.c_proj", ml, 4096, 4096, 4096, l),
        LinearSpec("llm.mlp.w1", ml, 4096, 4096, 11008, l, group="gate_up"),
        LinearSpec("llm.mlp.w2", ml, 4096, 4096, 11008, l, group="gate_up"),
        LinearSpec("llm.mlp.c_proj", ml, 11008, 11008, 4096, l, had_K=172),
    ]


def internvl2_8b_specs(batch: int = 1, msq: bool = False) -> List[LinearSpec]:
    """BASELINE config 4: InternVL2-8B, one 448^2 tile (1024 patches + class token -> 256 tokens after the
    pixel shuffle) + 512 text tokens per sample.  InternLM2-7B: hidden 4096, 32 layers, 32 heads / 8 KV heads
    (fused wqkv 4096 -> 6144), ff 14336 (online Hadamard 28 x 512 on w2); InternViT-300M: 1024 / 24 layers /
    mlp 4096 (power-of-two Hadamard on fc2); mlp1 4096 -> 4096 -> 4096."""
    v, l, mv, ml = 24, 32, (M_VIS + 1) * batch, M_LLM * batch
    return [
        LinearSpec("vis.patch_embedding", M_VIS * batch, 588, 588, 1024, 1, bias=True),
        LinearSpec("vis.attn.qkv", mv, 1024, 1024, 3072, v, bias=True),
        LinearSpec("vis.attn.proj", mv, 1024, 1024, 1024, v, bias=True),
        LinearSpec("vis.mlp.fc1", mv, 1024, 1024, 4096, v, bias=True),
        LinearSpec("vis.mlp.fc2", mv, 4096, 4096, 1024, v, bias=True, had_K=1),
        LinearSpec("mlp1.fc1", M_MERGED * batch, 4096, 4096, 4096, 1, bias=True),
        LinearSpec("mlp1.fc3", M_MERGED * batch, 4096, 4096, 4096, 1, bias=True),
        LinearSpec("llm.attention.wqkv", ml, 4096, 4096, 6144, l, msq=msq),
        LinearSpec("llm.attention.wo", ml, 4096, 4096, 4096, l, msq=msq),
        LinearSpec("llm.feed_forward.w1", ml, 4096, 4096, 14336, l, msq=msq, group="gate_up"),
        LinearSpec("llm.feed_forward.w3", ml, 4096, 4096, 14336, l, msq=msq, group="gate_up"),
        LinearSpec("llm.feed_forward.w2", ml, 14336, 14336, 4096, l, had_K=28, msq=msq),
    ]


def qwen2vl_72b_specs(batch: int = 1, msq: bool = True, v: int = 32, l: int = 80) -> List[LinearSpec]:
    """BASELINE config 5: Qwen2-VL-72B.  LLM hidden 8192, 80 layers, 64 heads / 8 KV heads, ff 29568 padded to
    30720 (online Hadamard 60 x 512); the ViT of the 7B model with a 5120 -> 5120 -> 8192 merger.  W4 image
    ~35 GB: one full replica per 288 GB GPU."""
    mv, ml = M_VIS * batch, M_LLM * batch
    return [
        LinearSpec("vis.patch_embed", mv, 1176, 1176, 1280, 1),
        LinearSpec("vis.attn.qkv", mv, 1280, 1280, 3840, v, bias=True),
        LinearSpec("vis.attn.proj", mv, 1280, 1280, 1280, v, bias=True),
        LinearSpec("vis.mlp.fc1", mv, 1280, 1280, 5120, v, bias=True),
        LinearSpec("vis.mlp.fc2", mv, 5120, 5120, 1280, v, bias=True, had_K=40, split=True),
        LinearSpec("merger.mlp.0", M_MERGED * batch, 5120, 5120, 5120, 1, bias=True),
        LinearSpec("merger.mlp.2", M_MERGED * batch, 5120, 5120, 8192, 1, bias=True),
        LinearSpec("llm.q_proj", ml, 8192, 8192, 8192, l, bias=True, msq=msq, group="qkv"),
        LinearSpec("llm.k_proj", ml, 8192, 8192, 1024, l, bias=True, msq=msq, group="qkv"),
        LinearSpec("llm.v_proj", ml, 8192, 8192, 1024, l, bias=True, msq=msq, group="qkv"),
        LinearSpec("llm.o_proj", ml, 8192, 8192, 8192, l, msq=msq),
        LinearSpec("llm.gate_proj", ml, 8192, 8192, 29568, l, msq=msq, group="gate_up"),
        LinearSpec("llm.up_proj", ml, 8192, 8192, 29568, l, msq=msq, group="gate_up"),
        LinearSpec("llm.down_proj", ml, 29568, 30720, 8192, l, had_K=60, msq=msq),
    ]


#: bench.py --workload: name -> (spec builder, description, LLM rows per sample)
WORKLOADS = {
    "qwen2vl_7b": (lambda batch: qwen2vl_7b_specs(msq=True, batch=batch),
                   "Qwen2-VL-7B W4A8 MSQ prefill, 1x448^2 image (1024 vision tokens) + 512 text tokens, 327 wrapped Linears"),
    "qwenvl_7b": (lambda batch: qwenvl_7b_specs(batch=batch),
                  "Qwen-VL-7B W4A8 prefill (BASELINE config 2), 1x448^2 image (1024 patches -> 256 tokens) + 512 text tokens, 354 wrapped Linears"),
    "internvl2_8b": (lambda batch: internvl2_8b_specs(batch=batch),
                     "InternVL2-8B W4A8 prefill (BASELINE config 4), one 448^2 tile (1025 ViT tokens -> 256) + 512 text tokens per sample, 259 wrapped Linears"),
    "qwen2vl_72b": (lambda batch: qwen2vl_72b_specs(batch=batch),
                    "Qwen2-VL-72B W4A8 MSQ prefill (BASELINE config 5), 1x448^2 image + 512 text tokens, 691 wrapped Linears, ~35 GB of W4 weights"),
}


def tiny_specs() -> List[LinearSpec]:
    """Same structure, small sizes: used by smoke tests."""
    return [
        LinearSpec("vis.mlp.fc2", 64, 1280, 1280, 96, 2, bias=True, had_K=40, split=True),
        LinearSpec("llm.q_proj", 48, 256, 256, 128, 2, bias=True, msq=True, group="qkv"),
        LinearSpec("llm.k_proj", 48, 256, 256, 32, 2, bias=True, msq=True, group="qkv"),
        LinearSpec("llm.v_proj", 48, 256, 256, 32, 2, bias=True, msq=True, group="qkv"),
        LinearSpec("llm.o_proj", 48, 256, 256, 256, 2, msq=True),
        LinearSpec("llm.gate_proj", 48, 256, 256, 704, 2, msq=True, group="gate_up"),
        LinearSpec("llm.up_proj", 48, 256, 256, 704, 2, msq=True, group="gate_up"),
        LinearSpec("llm.down_proj", 48, 700, 768, 256, 2, had_K=12, msq=True),
    ]


def tower_of(spec_name: str) -> str:
    """"llm" for the language model's Linears, "visual" for everything in front of it (vision tower, patch embedding, merger /
    resampler / mlp1): the split the reference's --visual_w_bits / --llm_w_bits make (gptq/qwen2vl_gptq_plus.py:14-100, 268-300
    use args.visual_w_bits for visual.patch_embed, visual.blocks and visual.merger; :394, :490 args.llm_w_bits)."""
    return "llm" if spec_name.split(".")[0] == "llm" else "visual"


def load_had_bits(K: int, device) -> Optional[torch.Tensor]:
    if K <= 1:
        return None
    from fake_quant import hadamard_utils
    return hadamard_utils.had_sign_bits(K, device)


def rtn_levels(w: torch.Tensor, bits: int = 4):
    """Symmetric per-output-channel RTN (reference quant_utils.py:446-518, mse off) in one launch
    of ``mq_wquant_sym``: offline preparation, not the timed path."""
    scale, levels, _, _ = ops.wquant_sym(w, bits)
    return levels, scale


def minmax_scale(mn: float, mx: float, dtype: torch.dtype = torch.float32) -> float:
    """observer/minmax.py:40-46, symmetric int8, including the zero-inclusion rule -- evaluated in the
    ACTIVATION's dtype like the reference's observer (min / max, their quotients by qmin / qmax and the
    maximum are tensors of x's dtype there, so an fp16 model gets an fp16-rounded scale)."""
    lo = torch.tensor(min(float(mn), 0.0), dtype=dtype)
    hi = torch.tensor(max(float(mx), 0.0), dtype=dtype)
    s = torch.max(torch.abs(lo / torch.tensor(-128.0, dtype=dtype)), torch.abs(hi / torch.tensor(127.0, dtype=dtype)))
    s.clamp_(float(np.finfo(np.float32).eps))
    return float(s)


def synth_inputs(specs: List[LinearSpec], device, dtype) -> Dict[tuple, torch.Tensor]:
    """One synthetic activation tensor per (rows, features): N(0, 1) with 0.1 % outlier channels x20."""
    gen = torch.Generator(device=device)
    inputs: Dict[tuple, torch.Tensor] = {}
    for spec in specs:
        key = (spec.M, spec.k_in)
        if key in inputs:
            continue
        gen.manual_seed(42 + len(inputs))
        x = torch.randn((spec.M, spec.k_in), generator=gen, device=device, dtype=torch.float32)
        n_out = max(1, int(round(spec.k_in * 0.001)))
        idx = torch.randperm(spec.k_in, generator=gen, device=device)[:n_out]
        x[:, idx] *= 20.0
        inputs[key] = x.to(dtype)
    return inputs


def synth_weight(spec: LinearSpec, li: int, seed: int, device, dtype):
    """Weight N(0, 0.02^2) [n, k] and bias N(0, 0.1^2) of the li-th Linear instance of the workload."""
    gen = torch.Generator(device=device)
    gen.manual_seed(seed + li)
    w = (torch.randn((spec.n, spec.k), generator=gen, device=device, dtype=torch.float32) * 0.02).to(dtype)
    # a bias of the model's dtype (what an fp16 checkpoint holds), handed on as fp32
    bias = (torch.randn((spec.n,), generator=gen, device=device) * 0.1).to(dtype).float() if spec.bias else None
    return w, bias


def vision_text_mask(M: int, device) -> torch.Tensor:
    """Token-type mask of the LLM rows: vision rows first (0), then text rows (1)."""
    s = torch.zeros((M,), dtype=torch.uint8, device=device)
    s[int(M * M_MERGED / M_LLM):] = 1
    return s


def execution_order(specs: List[LinearSpec]) -> List[tuple]:
    """Model order as (spec name, instance) pairs: consecutive specs of one tower (first name token) with
    the same instance count form a block that runs back to back, block after block."""
    order, i = [], 0
    while i < len(specs):
        j = i
        while j < len(specs) and specs[j].count == specs[i].count and \
                specs[j].name.split(".")[0] == specs[i].name.split(".")[0]:
            j += 1
        for c in range(specs[i].count):
            order.extend((sp.name, c) for sp in specs[i:j])
        i = j
    return order


class Layer:
    """One Linear instance of the prefill plus the synthetic input it is fed."""

    def __init__(self, spec: LinearSpec, idx: int, lin: W4A8Linear, x: torch.Tensor,
                 row_sel: Optional[torch.Tensor], out: torch.Tensor):
        self.spec, self.idx, self.lin, self.x, self.row_sel, self.out = spec, idx, lin, x, row_sel, out


class _HotPath:
    """step / kernel attribution / accounting over ``self.layers`` (engine, input, mask, output)."""

    def _calibrate(self, x, spec, had, row_sel):
        """Static scales from the observer kernels on the synthetic input (rotated if needed)."""
        src = x
        if had is not None:
            src = ops.hadamard(x, had.n, had.K, had.bits)
        cb = 1 if spec.split else 0
        if row_sel is None:
            mm = ops.minmax_tensor(src, cb).cpu()
            return minmax_scale(mm[0], mm[1], src.dtype), None
        nv = int((row_sel == 0).sum().item())
        mv = ops.minmax_tensor(src[:nv], cb).cpu()
        mt = ops.minmax_tensor(src[nv:], cb).cpu()
        return minmax_scale(mv[0], mv[1], src.dtype), minmax_scale(mt[0], mt[1], src.dtype)
    def step(self):
        """One pass of the hot path over the whole prefill; returns the last Linear's output."""
        y = None
        for L in self.layers:
            a, x0 = L.lin.quantize(L.x, L.row_sel)
            y = L.lin.gemm(a, x0, self.dtype, L.row_sel, L.out)
        return y

    def step_gemm_only(self):
        """Only the GEMM launches of step() (stale int8 activations): kernel attribution."""
        for L in self.layers:
            a = WORKSPACE.act(self.device, L.spec.M, L.lin.K_pad)
            x0 = WORKSPACE.x0(self.device, L.spec.M) if L.lin.split else None
            L.lin.gemm(a, x0, self.dtype, L.row_sel, L.out)

    def step_quant_only(self):
        for L in self.layers:
            L.lin.quantize(L.x, L.row_sel)

    def set_had_fast(self, on: bool) -> None:
        """NON-DEFAULT: the online rotations of THIS model's layers take (or stop taking) the fast K x K stage -- the
        per-call flag MQ_HAD_FAST, carried by each layer's HadamardSpec (engine.py).  Nothing process-wide changes; a
        captured hipGraph keeps the flags it was captured with."""
        for L in self.layers:
            if L.lin.had is not None:
                L.lin.had.fast = bool(on)

    # -- accounting --------------------------------------------------------------------
    def gemm_launches(self) -> int:
        return len(self.layers)

    def gemm_ops(self) -> int:
        return sum(L.lin.gemm_ops(L.spec.M) for L in self.layers)

    def gemm_bytes(self) -> int:
        return sum(L.lin.gemm_bytes(L.spec.M) for L in self.layers)

    def quant_bytes(self) -> int:
        return sum(L.lin.quant_bytes(L.spec.M) for L in self.layers)

    def weight_bytes(self) -> int:
        return sum(L.lin.w_img.numel() for L in self.layers)


class Prefill(_HotPath):
    """All wrapped Linears of one image+prompt prefill, in execution order."""

    def __init__(self, specs: List[LinearSpec], device="cuda:0", dtype=torch.float16,
                 w_bits: int = 4, seed: int = 1234, share_groups: bool = True, vis_w_bits: Optional[int] = None):
        self.device = torch.device(device)
        self.dtype = dtype
        self.specs = specs
        self.share_groups = share_groups
        self.layers: List[Layer] = []
        llm_w_bits, vis_w_bits = w_bits, (w_bits if vis_w_bits is None else vis_w_bits)
        inputs = synth_inputs(specs, self.device, dtype)
        outs: Dict[tuple, torch.Tensor] = {}
        sels: Dict[int, torch.Tensor] = {}
        scales: Dict[tuple, tuple] = {}
        pending: Dict[tuple, list] = {}
        li = 0
        for spec in specs:
            x = inputs[(spec.M, spec.k_in)]
            okey = (spec.M, spec.n)
            if okey not in outs:
                outs[okey] = torch.empty((spec.M, spec.n), dtype=dtype, device=self.device)
            row_sel = None
            if spec.msq:
                if spec.M not in sels:
                    sels[spec.M] = vision_text_mask(spec.M, self.device)
                row_sel = sels[spec.M]
            had = None
            if spec.had_K:
                had = HadamardSpec(spec.k, spec.had_K, load_had_bits(spec.had_K, self.device))
            skey = (spec.M, spec.k_in, spec.k, spec.had_K, spec.split, spec.msq)
            if skey not in scales:
                scales[skey] = self._calibrate(x, spec, had, row_sel)
            s0, s1 = scales[skey]
            w_bits = llm_w_bits if tower_of(spec.name) == "llm" else vis_w_bits      # --llm_w_bits / --visual_w_bits
            for c in range(spec.count):
                w, bias = synth_weight(spec, li, seed, self.device, dtype)
                li += 1
                if spec.split:
                    # the split column stays in floating point (L1); L2 = columns 1.. is quantized on its own
                    # (reference quant_utils.py:318-329, gptq pass over "L2")
                    q, s_w = rtn_levels(w[:, 1:].contiguous(), w_bits)
                    q = torch.cat([torch.zeros_like(q[:, :1]), q], dim=1).contiguous()
                else:
                    q, s_w = rtn_levels(w, w_bits)
                w0 = w[:, 0].float() if spec.split else None
                del w
                if share_groups and spec.group:
                    # Linears fed by the same tensor (q/k/v, gate/up) carry identical static scales
                    # (their observers saw the same activations): quantize once, ONE GEMM over the
                    # concatenated output channels.
                    pending.setdefault((spec.group, c), []).append((spec, q, s_w, bias, s0, s1, x, row_sel, w_bits))
                    continue
                lin = W4A8Linear(q, s_w, w_bits, bias, s0, s1 if spec.msq else None, had=had, w0=w0,
                                 in_features=spec.k_in)
                self.layers.append(Layer(spec, c, lin, x, row_sel, outs[okey]))
                del q
        for (gname, c), members in pending.items():
            spec0 = members[0][0]
            assert all(abs(mb[4] - members[0][4]) == 0 and mb[6] is members[0][6] for mb in members)
            q = torch.cat([mb[1] for mb in members], dim=0)
            s_w = torch.cat([mb[2] for mb in members], dim=0)
            bias = None
            if members[0][3] is not None:
                bias = torch.cat([mb[3] for mb in members], dim=0)
            n_total = q.shape[0]
            fused = LinearSpec(spec0.name.rsplit(".", 1)[0] + "." + gname + "[" +
                               "+".join(mb[0].name.rsplit(".", 1)[1] for mb in members) + "]",
                               spec0.M, spec0.k_in, spec0.k, n_total, spec0.count, bias is not None,
                               msq=spec0.msq, group=gname)
            okey = (fused.M, n_total)
            if okey not in outs:
                outs[okey] = torch.empty((fused.M, n_total), dtype=dtype, device=self.device)
            lin = W4A8Linear(q, s_w, members[0][8], bias, members[0][4],
                             members[0][5] if spec0.msq else None, in_features=spec0.k_in)
            L = Layer(fused, c, lin, members[0][6], members[0][7], outs[okey])
            L.order_name = spec0.name
            self.layers.append(L)
            del q
        pending.clear()
        self.layers = self._execution_order(self.layers)
        torch.cuda.synchronize(self.device)

    def _execution_order(self, layers):
        """Model order: a tower's per-block Linears run back to back, block after block."""
        by_key = {(getattr(L, "order_name", L.spec.name), L.idx): L for L in layers}
        ordered = [by_key[k] for k in execution_order(self.specs) if k in by_key]
        assert len(ordered) == len(layers)
        return ordered


class _Box(torch.nn.Module):
    """Plain container: the synthetic module tree only needs parents, names and children."""


class WrapperPrefill(_HotPath):
    """The same prefill built THROUGH the drop-in API instead of around it: the 327 Linears as a module
    tree (``<tower>.<block>.<leaf>``, q/k/v and gate/up under one parent like the HF layers), then exactly
    what exam/quant_qwen2vl.py does to a model -- ``add_actquant`` (reference quant_utils.py:626-662), the
    online-Hadamard / split / pad-hook flags (exam/quant_qwen2vl.py:107-143), the RTN weight pass
    (``fake_quant.gptq.rtn``), static int8 activation quantizers (MSQ on the LLM), the calibration
    protocol open -> forwards -> last -> close, ``model_quant``.  ``step()`` then calls every
    ``ActQuantWrapper.forward`` in model order on the same synthetic inputs ``Prefill`` uses, so the two
    objects must produce the same bits (tests/test_gpu_prefill_objects.py) and the same launches:
    ``model_quant`` groups q/k/v and gate/up into one quantize + one GEMM (``quant_utils.SiblingGroup``)."""

    def __init__(self, specs: List[LinearSpec], device="cuda:0", dtype=torch.float16, w_bits: int = 4,
                 seed: int = 1234, fuse_siblings: bool = True, calib_passes: int = 2, w_groupsize: int = -1,
                 vis_w_bits: Optional[int] = None):
        import functools
        import types
        from fake_quant import hadamard_utils as hu, quant_utils as qu, utils as fq_utils
        from fake_quant.gptq.rtn import rtn_module
        self.device = torch.device(device)
        self.dtype = dtype
        self.specs = specs
        self.share_groups = fuse_siblings
        self.qu = qu
        inputs = synth_inputs(specs, self.device, dtype)
        root = _Box()
        where: Dict[tuple, tuple] = {}          # (spec name, instance) -> (parent module, leaf)
        li = 0
        for spec in specs:
            tower, _, path = spec.name.partition(".")
            if not hasattr(root, tower):
                setattr(root, tower, torch.nn.ModuleList())
            blocks = getattr(root, tower)
            for c in range(spec.count):
                while len(blocks) <= c:
                    blocks.append(_Box())
                parent = blocks[c]
                parts = path.split(".")
                for part in parts[:-1]:
                    if not hasattr(parent, part):
                        setattr(parent, part, _Box())
                    parent = getattr(parent, part)
                w, bias = synth_weight(spec, li, seed, self.device, dtype)
                li += 1
                lin = torch.nn.Linear(spec.k, spec.n, bias=spec.bias, device="meta")
                lin.weight = torch.nn.Parameter(w, requires_grad=False)
                if spec.bias:
                    lin.bias = torch.nn.Parameter(bias.to(dtype), requires_grad=False)
                leaf = parts[-1] if not parts[-1].isdigit() else "fc" + parts[-1]
                setattr(parent, leaf, lin)
                where[(spec.name, c)] = (parent, leaf)
        self.root = root
        qu.add_actquant(root)
        by_name = {sp.name: sp for sp in specs}
        self.calls = []                          # (wrapper, input) in model order
        for name, c in execution_order(specs):
            parent, leaf = where[(name, c)]
            wrap, spec = getattr(parent, leaf), by_name[name]
            assert isinstance(wrap, qu.ActQuantWrapper)
            if spec.had_K:
                wrap.had_K, wrap.K = hu.get_hadK(spec.k)
                wrap.online_full_had = True
            if spec.split:
                wrap.split = True
                wrap.split_weights()
            if spec.k != spec.k_in:
                wrap.register_forward_pre_hook(functools.partial(fq_utils.revise_down_input, new_size=spec.k))
            wrap.quantizer.configure(bits=8, sym=True, static=True, observer_type="minmax", msq=spec.msq)
            self.calls.append((wrap, inputs[(spec.M, spec.k_in)], spec))
        self.quantizers: Dict[str, object] = {}
        # w_groupsize > 0: group-wise weight scales as a --w_groupsize GPTQ run leaves them (fake_quant.gptq.rtn, mq_gemm_w4a8_wgroupscale)
        # per tower like the reference's RTN / GPTQ drivers: --visual_w_bits for everything in front of the language model,
        # --llm_w_bits for its layers (docs/qwen2vl.md:19 is W8A8 vision + W4A8 LLM, :26 W4A8 + W4A8)
        for tower in sorted({sp.name.partition(".")[0] for sp in specs}):
            bits = w_bits if tower_of(tower) == "llm" else (w_bits if vis_w_bits is None else vis_w_bits)
            rtn_module(getattr(root, tower), "model." + tower, bits, True, False, [], self.quantizers, groupsize=w_groupsize)
        self.args = types.SimpleNamespace(skip_names=[], no_sibling_fusion=not fuse_siblings)
        m_llm = max([sp.M for sp in specs if sp.msq] + [0])
        self.mask = vision_text_mask(m_llm, self.device) if m_llm else None
        qu.set_token_type_mask(self.mask)
        qu.model_open_calibrate(root, self.args)
        for i in range(calib_passes):
            if i == calib_passes - 1:
                qu.model_open_last_calibrate(root, self.args)
            self._forward_all()
        qu.model_close_calibrate(root, self.args)
        qu.model_quant(root, self.args)
        self._forward_all()                      # freezes every wrapper / group into its integer engine
        torch.cuda.synchronize(self.device)
        # the engines the forwards above built, in launch order: kernel attribution and accounting
        self.layers: List[Layer] = []
        seen = set()
        for wrap, x, spec in self.calls:
            grp = wrap.__dict__.get("_group")
            lin = grp.engine if grp is not None and grp.enabled else wrap._real
            if id(lin) in seen:
                continue
            seen.add(id(lin))
            sel = self.mask if spec.msq else None
            self.layers.append(Layer(spec, 0, lin, x, sel, None))

    def _forward_all(self):
        out = None
        for wrap, x, _ in self.calls:
            out = wrap(x)
        return out

    def step(self):
        """One pass of the hot path: every ``ActQuantWrapper.forward`` of the prefill, in model order."""
        self.qu.set_token_type_mask(self.mask)
        return self._forward_all()

    def outputs(self) -> List[torch.Tensor]:
        """Every wrapper's output of one pass (tests)."""
        self.qu.set_token_type_mask(self.mask)
        return [wrap(x) for wrap, x, _ in self.calls]
