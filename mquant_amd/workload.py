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


def tiny_specs() -> List[LinearSpec]:
    """Same structure, small sizes: used by smoke tests."""
    return [
        LinearSpec("vis.mlp.fc2", 64, 1280, 1280, 96, 2, bias=True, had_K=40, split=True),
        LinearSpec("llm.q_proj", 48, 256, 256, 128, 2, bias=True, msq=True, group="qkv"),
        LinearSpec("llm.k_proj", 48, 256, 256, 32, 2, bias=True, msq=True, group="qkv"),
        LinearSpec("llm.down_proj", 48, 700, 768, 64, 2, had_K=12, msq=True),
    ]


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


def minmax_scale(mn: float, mx: float) -> float:
    """observer/minmax.py:40-46, symmetric int8, including the zero-inclusion rule."""
    mn = min(float(mn), 0.0)
    mx = max(float(mx), 0.0)
    s = max(abs(np.float32(mn) / np.float32(-128.0)), abs(np.float32(mx) / np.float32(127.0)))
    return float(max(np.float32(s), np.float32(np.finfo(np.float32).eps)))


class Layer:
    """One Linear instance of the prefill plus the synthetic input it is fed."""

    def __init__(self, spec: LinearSpec, idx: int, lin: W4A8Linear, x: torch.Tensor,
                 row_sel: Optional[torch.Tensor], out: torch.Tensor):
        self.spec, self.idx, self.lin, self.x, self.row_sel, self.out = spec, idx, lin, x, row_sel, out


class Prefill:
    """All wrapped Linears of one image+prompt prefill, in execution order."""

    def __init__(self, specs: List[LinearSpec], device="cuda:0", dtype=torch.float16,
                 w_bits: int = 4, seed: int = 1234, share_groups: bool = True):
        self.device = torch.device(device)
        self.dtype = dtype
        self.specs = specs
        self.share_groups = share_groups
        self.layers: List[Layer] = []
        gen = torch.Generator(device=self.device)
        inputs: Dict[tuple, torch.Tensor] = {}
        outs: Dict[tuple, torch.Tensor] = {}
        sels: Dict[int, torch.Tensor] = {}
        scales: Dict[tuple, tuple] = {}
        pending: Dict[tuple, list] = {}
        li = 0
        for spec in specs:
            key = (spec.M, spec.k_in)
            if key not in inputs:
                gen.manual_seed(42 + len(inputs))
                x = torch.randn((spec.M, spec.k_in), generator=gen, device=self.device, dtype=torch.float32)
                n_out = max(1, int(round(spec.k_in * 0.001)))
                idx = torch.randperm(spec.k_in, generator=gen, device=self.device)[:n_out]
                x[:, idx] *= 20.0
                inputs[key] = x.to(dtype)
            x = inputs[key]
            okey = (spec.M, spec.n)
            if okey not in outs:
                outs[okey] = torch.empty((spec.M, spec.n), dtype=dtype, device=self.device)
            row_sel = None
            if spec.msq:
                if spec.M not in sels:
                    s = torch.zeros((spec.M,), dtype=torch.uint8, device=self.device)
                    s[int(spec.M * M_MERGED / M_LLM):] = 1   # vision rows first, then text rows
                    sels[spec.M] = s
                row_sel = sels[spec.M]
            had = None
            if spec.had_K:
                had = HadamardSpec(spec.k, spec.had_K, load_had_bits(spec.had_K, self.device))
            skey = (spec.M, spec.k_in, spec.k, spec.had_K, spec.split, spec.msq)
            if skey not in scales:
                scales[skey] = self._calibrate(x, spec, had, row_sel)
            s0, s1 = scales[skey]
            for c in range(spec.count):
                gen.manual_seed(seed + li)
                li += 1
                w = (torch.randn((spec.n, spec.k), generator=gen, device=self.device,
                                 dtype=torch.float32) * 0.02).to(dtype)
                q, s_w = rtn_levels(w, w_bits)
                w0 = w[:, 0].float() if spec.split else None
                bias = None
                if spec.bias:
                    bias = (torch.randn((spec.n,), generator=gen, device=self.device) * 0.1).float()
                del w
                if share_groups and spec.group:
                    # Linears fed by the same tensor (q/k/v, gate/up) carry identical static scales
                    # (their observers saw the same activations): quantize once, ONE GEMM over the
                    # concatenated output channels.
                    pending.setdefault((spec.group, c), []).append((spec, q, s_w, bias, s0, s1, x, row_sel))
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
            lin = W4A8Linear(q, s_w, w_bits, bias, members[0][4],
                             members[0][5] if spec0.msq else None, in_features=spec0.k_in)
            L = Layer(fused, c, lin, members[0][6], members[0][7], outs[okey])
            L.order_name = spec0.name
            self.layers.append(L)
            del q
        pending.clear()
        self.layers = self._execution_order(self.layers)
        torch.cuda.synchronize(self.device)

    @staticmethod
    def _execution_order(layers):
        """Model order: a tower's per-block Linears run back to back, block after block."""
        by_name: Dict[str, List[Layer]] = {}
        for L in layers:
            by_name.setdefault(getattr(L, "order_name", L.spec.name), []).append(L)
        ordered: List[Layer] = []

        def take(names, reps):
            for i in range(reps):
                for nm in names:
                    if nm in by_name and i < len(by_name[nm]):
                        ordered.append(by_name[nm][i])

        towers = [["vis.patch_embed"],
                  ["vis.attn.qkv", "vis.attn.proj", "vis.mlp.fc1", "vis.mlp.fc2"],
                  ["merger.mlp.0", "merger.mlp.2"],
                  ["llm.q_proj", "llm.k_proj", "llm.v_proj", "llm.o_proj", "llm.gate_proj",
                   "llm.up_proj", "llm.down_proj"]]
        for names in towers:
            reps = max([len(by_name.get(nm, [])) for nm in names] + [0])
            take(names, reps)
        assert len(ordered) == len(layers)
        return ordered

    def _calibrate(self, x, spec, had, row_sel):
        """Static scales from the observer kernels on the synthetic input (rotated if needed)."""
        src = x
        if had is not None:
            src = ops.hadamard(x, had.n, had.K, had.bits)
        cb = 1 if spec.split else 0
        if row_sel is None:
            mm = ops.minmax_tensor(src, cb).cpu()
            return minmax_scale(mm[0], mm[1]), None
        nv = int((row_sel == 0).sum().item())
        mv = ops.minmax_tensor(src[:nv], cb).cpu()
        mt = ops.minmax_tensor(src[nv:], cb).cpu()
        return minmax_scale(mv[0], mv[1]), minmax_scale(mt[0], mt[1])

    # ---------------------------------------------------------------------------------
    def step(self):
        """One pass of the hot path over the whole prefill."""
        for L in self.layers:
            a, x0 = L.lin.quantize(L.x, L.row_sel)
            L.lin.gemm(a, x0, self.dtype, L.row_sel, L.out)

    def step_gemm_only(self):
        """Only the GEMM launches of step() (stale int8 activations): kernel attribution."""
        for L in self.layers:
            a = WORKSPACE.act(self.device, L.spec.M, L.lin.K_pad)
            x0 = WORKSPACE.x0(self.device, L.spec.M) if L.lin.split else None
            L.lin.gemm(a, x0, self.dtype, L.row_sel, L.out)

    def step_quant_only(self):
        for L in self.layers:
            L.lin.quantize(L.x, L.row_sel)

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
