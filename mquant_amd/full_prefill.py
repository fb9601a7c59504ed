"""Whole synthetic Qwen2-VL-7B prefill (SURVEY 8(d) "(ii) full synthetic prefill"): the W4A8
Linears of ``workload.Prefill`` chained with the operators that sit between them in the model
-- weight-less RMS norms (LayerNorms are fused away by the rotation pass), rotary embedding,
PyTorch-ROCm SDPA, GELU / SiLU-gate, residual adds, fp16 ``lm_head`` on the last position
(it is not wrapped: reference quant_utils.py:560-564).  Those glue operators are torch; the Linears
are this repository's kernels, and with ``fused_glue`` also norm -> quantize, activation ->
Hadamard -> quantize and the residual adds (GEMM epilogue).  Used for the TTFT report in bench.py, never for parity.

Static activation scales are re-calibrated on the chained activations (min/max observer kernels,
one calibration pass) so that the int8 grids are the ones this dataflow would get.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

from . import ops
from .workload import M_MERGED, M_TXT, M_VIS, Layer, Prefill

VOCAB = 152064
VIS_DIM, VIS_HEADS = 1280, 16
LLM_DIM, LLM_HEADS, LLM_KV_HEADS, HEAD_DIM = 3584, 28, 4, 128


@dataclass
class Geometry:
    """Widths the glue operators need (the Linears carry their own shapes): Qwen2-VL family."""
    vis_dim: int = VIS_DIM
    vis_heads: int = VIS_HEADS
    llm_dim: int = LLM_DIM
    heads: int = LLM_HEADS
    kv_heads: int = LLM_KV_HEADS
    head_dim: int = HEAD_DIM
    vocab: int = VOCAB


QWEN2VL_7B = Geometry()
QWEN2VL_72B = Geometry(llm_dim=8192, heads=64, kv_heads=8, head_dim=128)     # BASELINE configuration 5


def _rope_tables(rows: int, dim: int, device, dtype, base: float = 10000.0):
    inv = 1.0 / (base ** (torch.arange(0, dim, 2, device=device, dtype=torch.float32) / dim))
    ang = torch.arange(rows, device=device, dtype=torch.float32)[:, None] * inv[None, :]
    ang = torch.cat([ang, ang], dim=-1)
    return ang.cos().to(dtype)[:, None, :], ang.sin().to(dtype)[:, None, :]


def _rope(x, cos, sin):
    """x [T, heads, d]: rotate-half convention."""
    h = x.shape[-1] // 2
    rot = torch.cat([-x[..., h:], x[..., :h]], dim=-1)
    return x * cos + rot * sin


class FullPrefill:
    def __init__(self, pf: Prefill, seed: int = 7, fused_glue: bool = True, sample: int = 0,
                 geometry: Optional[Geometry] = None, kv_fp8: bool = False, attn_fp8: bool = False,
                 attn_kernel: Optional[bool] = None):
        #: norm -> quantize and activation -> Hadamard -> quantize as single launches
        self.fused_glue = fused_glue
        self.g = geometry or QWEN2VL_7B
        #: fp8 (e4m3fn) KV cache, SURVEY 8(f4): the K|V columns of the fused q|k|v GEMM output are quantized on write
        #: with one static scale per KV head (calibrated with the activation scales), and the attention of the same
        #: step consumes the cache contents read back by the same launch (mq_kv_quant_fp8_readback)
        self.kv_fp8 = kv_fp8
        #: with kv_fp8: the attention of the prefill reads the e4m3 cache itself (mq_attn_prefill_fp8kv) instead of a
        #: half-precision read-back through torch SDPA -- no fp16 copy of K / V exists after the q|k|v GEMM output
        self.attn_fp8 = attn_fp8
        assert kv_fp8 or not attn_fp8, "attn_fp8 reads the fp8 cache: it needs kv_fp8"
        #: decoder attention over 16-bit K / V with this repository's kernel (mq_attn_prefill: reads the q / k / v column
        #: slices of the fused GEMM output in place and writes the [T, heads * head_dim] layout o_proj consumes) instead
        #: of torch SDPA; part of the fused glue by default
        own = fused_glue if attn_kernel is None else attn_kernel
        #: with the attention kernels: emit the int8 levels of the o_proj / proj input straight from the attention store
        #: (mq_attn_prefill_quant_i8) instead of a 16-bit tensor plus a quantize launch
        self.attn_quant = own
        self.attn_kernel = own and self.g.head_dim == 128
        self.vis_attn_kernel = own and self.g.vis_dim // self.g.vis_heads in (80, 128)      # the vision tower's (non-causal) attention
        #: the decoder's RoPE rides in the q|k|v GEMM's store (mq_gemm_w4a8_rope_ws) instead of its own launch (round 5)
        self.rope_fused = fused_glue
        #: silu(gate) * up and the vision tower's QuickGELU ride in the store of the PRODUCING GEMM (mq_gemm_w4a8_act_ws, round 6):
        #: one [T, 18944] tensor leaves gate|up instead of [T, 37888], and down_proj / fc2 run their plain Hadamard + quantize
        #: launch; False = round 5's form, the activation in the Hadamard kernel's prologue (mq_act_hadamard_quant_i8)
        self.act_in_gemm = fused_glue
        self.kv_cache: List[torch.Tensor] = []
        self.kv_scales: List[torch.Tensor] = []
        assert pf.share_groups, "the chained prefill uses the fused q/k/v and gate/up GEMMs"
        self.pf, self.dev, self.dtype = pf, pf.device, pf.dtype
        by: Dict[str, List[Layer]] = {}
        for L in pf.layers:
            by.setdefault(getattr(L, "order_name", L.spec.name), []).append(L)
        self.by = by
        g = torch.Generator(device=self.dev).manual_seed(seed)
        rnd = lambda *shape, std=1.0: (torch.randn(shape, generator=g, device=self.dev) * std).to(self.dtype)
        self.patches = rnd(M_VIS, 1176)
        self.text_embeds = rnd(M_TXT, self.g.llm_dim)
        self.lm_head = rnd(self.g.vocab, self.g.llm_dim, std=0.02)             # a weight: the same for every sample
        self.sample = sample
        if sample:
            self.set_sample(sample)
        self.vcos, self.vsin = _rope_tables(M_VIS, self.g.vis_dim // self.g.vis_heads, self.dev, self.dtype)
        self.lcos, self.lsin = _rope_tables(M_MERGED + M_TXT, self.g.head_dim, self.dev, self.dtype, 1e6)
        self.vcos2, self.vsin2 = self.vcos[:, 0].contiguous(), self.vsin[:, 0].contiguous()     # [T, d] for the kernel
        self.lcos2, self.lsin2 = self.lcos[:, 0].contiguous(), self.lsin[:, 0].contiguous()
        self.calibrating = False
        self.logits = None

    def set_sample(self, sample: int) -> None:
        """Inputs of image+prompt ``sample`` (batch sharding: sample i runs on rank i % world).  Sample 0
        is the calibration sample; weights and static scales do not depend on the sample."""
        g = torch.Generator(device=self.dev).manual_seed(104729 + 7919 * sample)
        rnd = lambda *shape: torch.randn(shape, generator=g, device=self.dev).to(self.dtype)
        if sample == 0:
            g0 = torch.Generator(device=self.dev).manual_seed(7)
            rnd = lambda *shape: torch.randn(shape, generator=g0, device=self.dev).to(self.dtype)
        self.patches = rnd(M_VIS, 1176)
        self.text_embeds = rnd(M_TXT, self.g.llm_dim)
        self.sample = sample

    # -- one wrapped Linear (or fused group) -----------------------------------------------
    def _lin(self, L: Layer, x: torch.Tensor, residual: torch.Tensor = None) -> torch.Tensor:
        """Linear(x) [+ residual, folded into the GEMM epilogue when the glue is fused]."""
        if self.calibrating:
            s0, s1 = self.pf._calibrate(x, L.spec, L.lin.had, L.row_sel)
            L.lin.s_x0 = s0
            if L.lin.s_x1 is not None:
                L.lin.s_x1 = s1
        if residual is None:
            return L.lin.forward(x, L.row_sel)
        if self.calibrating or not self.fused_glue or L.lin.col_perm is not None or L.lin.w_groups is not None:
            return residual + L.lin.forward(x, L.row_sel)
        a, x0 = L.lin.quantize(x, L.row_sel)
        return L.lin.gemm_residual(a, x0, residual, L.row_sel)

    def _norm_lin(self, L: Layer, x: torch.Tensor, dim: int) -> torch.Tensor:
        """Linear(RMSN(x))."""
        if self.calibrating or not self.fused_glue or L.lin.col_perm is not None:      # (--act_order engines gather their columns: quantize() only)
            return self._lin(L, F.rms_norm(x, (dim,), eps=1e-6))
        a, _ = L.lin.quantize_rmsn(x, dim, 1e-6, L.row_sel)
        return L.lin.gemm(a, None, self.dtype, L.row_sel)

    def _norm_lin_act(self, L: Layer, x: torch.Tensor, dim: int, act: int) -> Optional[torch.Tensor]:
        """act(Linear(RMSN(x))) with the activation in the GEMM's store, or None when this layer / mode cannot take it."""
        if (self.calibrating or not (self.fused_glue and self.act_in_gemm) or not L.lin.act_in_store_ok(act) or L.lin.had is not None
                or x.shape[0] <= 64):                                      # (a few rows: the weight-streaming kernels have no act epilogue)
            return None
        a, _ = L.lin.quantize_rmsn(x, dim, 1e-6, L.row_sel)
        return L.lin.gemm_act(a, act, self.dtype, L.row_sel)

    def _act_lin(self, L: Layer, x: torch.Tensor, x2, act: int, residual: torch.Tensor) -> torch.Tensor:
        """residual + Linear(act(x[, x2])) for a layer with an online Hadamard."""
        if self.calibrating or not self.fused_glue or L.lin.col_perm is not None:
            h = F.silu(x) * x2 if act == ops.ACT_SILU_MUL else x * torch.sigmoid(1.702 * x)
            return self._lin(L, h, residual)
        a, x0 = L.lin.quantize_act(x, x2, act, L.row_sel)
        return L.lin.gemm_residual(a, x0, residual, L.row_sel)

    def _vis_mlp(self, i: int, x: torch.Tensor) -> torch.Tensor:
        """x + fc2(quick_gelu(fc1(RMSN(x)))) of vision block i (hidden_act = quick_gelu)."""
        VD = self.g.vis_dim
        f = self._norm_lin_act(self.by["vis.mlp.fc1"][i], x, VD, ops.ACT_QUICK_GELU)     # fc1 storing quick_gelu(fc1(.))
        if f is not None:
            return self._lin(self.by["vis.mlp.fc2"][i], f, residual=x)
        f = self._norm_lin(self.by["vis.mlp.fc1"][i], x, VD)
        return self._act_lin(self.by["vis.mlp.fc2"][i], f, None, ops.ACT_QUICK_GELU, residual=x)

    def calibrate(self):
        """One calibration pass over the CHAINED activations; the hot path's own scales (calibrated on its
        synthetic per-layer inputs) are remembered so that the two sets can be swapped."""
        self._hot_scales = [(L.lin.s_x0, L.lin.s_x1) for L in self.pf.layers]
        self.calibrating = True
        self.step()
        self.calibrating = False
        self._full_scales = [(L.lin.s_x0, L.lin.s_x1) for L in self.pf.layers]
        torch.cuda.synchronize(self.dev)

    def restore_hot_path_scales(self):
        for L, (s0, s1) in zip(self.pf.layers, self._hot_scales):
            L.lin.s_x0, L.lin.s_x1 = s0, s1

    def apply_full_prefill_scales(self):
        for L, (s0, s1) in zip(self.pf.layers, self._full_scales):
            L.lin.s_x0, L.lin.s_x1 = s0, s1

    # -- the prefill -----------------------------------------------------------------------
    def kv_cache_bytes(self) -> int:
        """Bytes the K / V of this prefill occupy in the cache (all layers)."""
        layers = len(self.by["llm.q_proj"])
        per_elem = 1 if self.kv_fp8 else torch.empty((), dtype=self.dtype).element_size()
        return layers * (M_MERGED + M_TXT) * 2 * self.g.kv_heads * self.g.head_dim * per_elem

    def step(self) -> torch.Tensor:
        by, g = self.by, self.g
        VD, VH = g.vis_dim, g.vis_heads
        D, H, KVH, HD = g.llm_dim, g.heads, g.kv_heads, g.head_dim
        # vision tower
        x = self._lin(by["vis.patch_embed"][0], self.patches)
        for i in range(len(by["vis.attn.qkv"])):
            qkv = self._norm_lin(by["vis.attn.qkv"][i], x, VD)
            if self.fused_glue:                               # q and k rotated in place, one launch
                ops.rope_inplace(qkv[:, :2 * VD], 2 * VH, VD // VH, self.vcos2, self.vsin2)
                q, k, v = qkv.view(M_VIS, 3, VH, -1).unbind(1)
            else:
                q, k, v = qkv.view(M_VIS, 3, VH, -1).unbind(1)
                q, k = _rope(q, self.vcos, self.vsin), _rope(k, self.vcos, self.vsin)
            Lp = by["vis.attn.proj"][i]
            if (self.vis_attn_kernel and self.fused_glue and self.attn_quant and not self.calibrating and i > 0
                    and Lp.lin.had is None and not Lp.lin.split and Lp.lin.col_perm is None):
                qa = Lp.lin.act_buffer(M_VIS)
                ops.attn_prefill_quant_i8(q, Lp.lin.s_x0, Lp.lin.s_x1, k=k, v=v, causal=False, row_sel=Lp.row_sel, out=qa)
                x = Lp.lin.gemm_residual(qa, None, x, Lp.row_sel)
                x = self._vis_mlp(i, x)
                continue
            if self.vis_attn_kernel:
                flat = ops.attn_prefill(q, k, v, causal=False)
            else:
                a = F.scaled_dot_product_attention(q.transpose(0, 1)[None], k.transpose(0, 1)[None],
                                                   v.transpose(0, 1)[None])[0]
                flat = a.transpose(0, 1).reshape(M_VIS, VD)
            if i == 0:
                self.vis_attn_first = flat     # [patches, vis_dim] of the first vision block (tests)
            x = self._lin(by["vis.attn.proj"][i], flat, residual=x)
            x = self._vis_mlp(i, x)
        m = F.rms_norm(x, (VD,), eps=1e-6).view(M_MERGED, 4 * VD)
        m = self._lin(by["merger.mlp.2"][0], F.gelu(self._lin(by["merger.mlp.0"][0], m)))
        # language model: [vision tokens | text tokens]
        hdn = torch.cat([m, self.text_embeds], dim=0)
        T = hdn.shape[0]
        kv = KVH * HD
        for i in range(len(by["llm.q_proj"])):
            Lq = by["llm.q_proj"][i]
            rope_in_gemm = (self.fused_glue and self.rope_fused and not self.calibrating and HD == 128 and Lq.lin.had is None
                            and not Lq.lin.split and Lq.lin.dynamic is None and Lq.lin.w_shift is None and Lq.lin.w_groups is None and Lq.lin.col_perm is None
                            and self.dtype in (torch.float16, torch.bfloat16))
            if rope_in_gemm:
                # RMS norm -> quantize (one launch), then the fused q|k|v GEMM whose q | k heads leave the store rotated
                a, _ = Lq.lin.quantize_rmsn(hdn, D, 1e-6, Lq.row_sel)
                qkv = Lq.lin.gemm_rope(a, self.lcos2, self.lsin2, D + kv, self.dtype, Lq.row_sel)
            else:
                qkv = self._norm_lin(Lq, hdn, D)                   # fused q|k|v GEMM
            if self.fused_glue:
                if not rope_in_gemm:
                    ops.rope_inplace(qkv[:, :D + kv], H + KVH, HD, self.lcos2, self.lsin2)
                q = qkv[:, :D].view(T, H, HD)
                k = qkv[:, D:D + kv].view(T, KVH, HD)
                v = qkv[:, D + kv:].view(T, KVH, HD)
                kv_cols = qkv[:, D:].view(T, 2 * KVH, HD)          # K | V side by side, K already rotated
            else:
                q = _rope(qkv[:, :D].view(T, H, HD), self.lcos, self.lsin)
                k = _rope(qkv[:, D:D + kv].view(T, KVH, HD), self.lcos, self.lsin)
                v = qkv[:, D + kv:].view(T, KVH, HD)
                kv_cols = torch.cat([k, v], dim=1) if self.kv_fp8 else None
            if self.kv_fp8:
                if self.calibrating:
                    if len(self.kv_scales) <= i:
                        self.kv_scales.append(ops.kv_scale_from_absmax(kv_cols))
                        self.kv_cache.append(torch.empty((T, 2 * KVH, HD), dtype=torch.float8_e4m3fn, device=self.dev))
                    else:
                        self.kv_scales[i] = ops.kv_scale_from_absmax(kv_cols)
            Lo = by["llm.o_proj"][i]
            # attention -> o_proj's static quantizer in ONE launch (the int8 levels of the attention output, tiled): fused
            # glue, outside calibration, o_proj without an online Hadamard / split
            quant_out = (self.fused_glue and self.attn_quant and not self.calibrating and i > 0
                         and Lo.lin.had is None and not Lo.lin.split and Lo.lin.col_perm is None)
            if self.kv_fp8 and self.attn_fp8 and not self.calibrating:
                # write the cache (e4m3, static per-head scales); the attention kernel reads those bytes
                ops.kv_quant_fp8(kv_cols, self.kv_scales[i], out=self.kv_cache[i])
                if quant_out:
                    qa = Lo.lin.act_buffer(T)
                    ops.attn_prefill_quant_i8(q, Lo.lin.s_x0, Lo.lin.s_x1, kv_cache=self.kv_cache[i], kv_scale=self.kv_scales[i],
                                              causal=True, row_sel=Lo.row_sel, out=qa)
                    hdn = Lo.lin.gemm_residual(qa, None, hdn, Lo.row_sel)
                else:
                    flat = ops.attn_prefill_fp8kv(q, self.kv_cache[i], self.kv_scales[i], causal=True)      # [T, heads * head_dim]
                    a = flat.view(T, H, HD).transpose(0, 1)
            elif quant_out and self.attn_kernel:
                if self.kv_fp8:
                    _, hat = ops.kv_quant_fp8_readback(kv_cols, self.kv_scales[i], out=self.kv_cache[i])
                    k, v = hat[:, :KVH], hat[:, KVH:]
                qa = Lo.lin.act_buffer(T)
                ops.attn_prefill_quant_i8(q, Lo.lin.s_x0, Lo.lin.s_x1, k=k, v=v, causal=True, row_sel=Lo.row_sel, out=qa)
                hdn = Lo.lin.gemm_residual(qa, None, hdn, Lo.row_sel)
            else:
                quant_out = False
                if self.kv_fp8:
                    # write the cache and attend over what was written, read back by the same launch
                    _, hat = ops.kv_quant_fp8_readback(kv_cols, self.kv_scales[i], out=self.kv_cache[i])
                    k, v = hat[:, :KVH], hat[:, KVH:]
                if self.attn_kernel:
                    flat = ops.attn_prefill(q, k, v, causal=True)
                    a = flat.view(T, H, HD).transpose(0, 1)
                else:
                    a = F.scaled_dot_product_attention(q.transpose(0, 1)[None], k.transpose(0, 1)[None],
                                                       v.transpose(0, 1)[None], is_causal=True, enable_gqa=True)[0]
                    flat = a.transpose(0, 1).reshape(T, D)
            if i == 0:
                self.attn_first = a            # [heads, T, head_dim] of the first decoder layer (tests; layer 0 keeps the 16-bit output)
            if not quant_out:
                hdn = self._lin(Lo, flat, residual=hdn)
            h = self._norm_lin_act(by["llm.gate_proj"][i], hdn, D, ops.ACT_SILU_MUL)     # fused gate|up GEMM storing silu(gate) * up
            if h is not None:
                hdn = self._lin(by["llm.down_proj"][i], h, residual=hdn)
            else:
                gu = self._norm_lin(by["llm.gate_proj"][i], hdn, D)    # fused gate|up GEMM
                half = gu.shape[1] // 2
                hdn = self._act_lin(by["llm.down_proj"][i], gu[:, :half], gu[:, half:], ops.ACT_SILU_MUL, residual=hdn)
        last = F.rms_norm(hdn[-1:], (D,), eps=1e-6)
        self.logits = ops.gemv_f16(last, self.lm_head)           # the 16-bit lm_head on one row: an HBM stream (mq_gemv_f16)
        return self.logits
