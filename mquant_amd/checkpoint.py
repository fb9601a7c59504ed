"""Flat export of a quantized model: packed int4 weights in the reference wire
format + scales, instead of pickling whole modules (reference exam/quant_qwen2vl.py:147-160 dumps
``torch.save(model)`` with fake-quantized fp16 weights -- 4x the bytes and tied to class paths).

One entry group per ``ActQuantWrapper`` (dotted module name ``<n>``):

    <n>.qweight    uint8 [N, K/2]  reference pack_i4 bytes (quant_utils.py:61-69: two's-complement
                                   nibbles, even k -> low nibble)            (w_bits == 4)
                   int8  [N, K]                                              (w_bits == 8)
                   K includes the zeroed column 0 of a ``split`` layer and the Hadamard padding
    <n>.w_scale    fp32 [N]        per-output-channel weight scale
    <n>.act_scale  fp32 [2]        static activation scales: [vision | all tokens, text tokens]
    <n>.bias       fp32 [N]        (optional)
    <n>.w0         fp32 [N]        (optional) the fp32 column of a ``split`` layer (L1)
    <n>.meta       int64 [12]      see META below

Everything else of the model (norms, embeddings, lm_head) stays in its ordinary ``state_dict``.
Files are safetensors.  ``load_quantized`` rebuilds the frozen ``W4A8Linear`` of every wrapper
without touching (or needing) floating-point weights.
"""
from __future__ import annotations

from typing import Dict

import torch

META = ("version", "w_bits", "a_bits", "N", "K", "in_features", "had_K", "fp32_had", "split", "msq",
        "conv_rank", "reserved")
VERSION = 1


def _meta(**kw) -> torch.Tensor:
    return torch.tensor([int(kw.get(k, 0)) for k in META], dtype=torch.int64)


def read_meta(t: torch.Tensor) -> Dict[str, int]:
    vals = [int(v) for v in t.reshape(-1).tolist()]
    if len(vals) != len(META) or vals[0] != VERSION:
        raise ValueError(f"unsupported quantized-layer record (meta = {vals})")
    return dict(zip(META, vals))


# ------------------------------------------------------------------------------------------ export
def export_wrapper(wrapper, device=None) -> Dict[str, torch.Tensor]:
    """Tensors of one calibrated, weight-quantized ``ActQuantWrapper`` (CPU, contiguous)."""
    from . import ops
    name, wmod = wrapper._weight_module()
    wq = wrapper.weight_quantizers.get(name)
    qz = wrapper.quantizer
    if wq is None or not getattr(qz, "static", False) or qz.quantizer.scale is None:
        raise ValueError("export needs a wrapper with a static calibrated activation quantizer and an "
                         "attached symmetric WeightQuantizer")
    if not getattr(wq, "sym", False):
        raise ValueError("the flat checkpoint stores symmetric weight levels only; a wrapper whose weights were "
                         "quantized asymmetrically (--w_asym) runs from the in-memory engine, not from this format")
    device = device or wmod.weight.device
    W = wmod.weight.data.to(device)
    W2 = W.reshape(W.shape[0], -1)
    scale = wq.scale.reshape(-1).to(device=device, dtype=torch.float32)
    if scale.numel() == 1:
        scale = scale.expand(W2.shape[0]).contiguous()
    levels = ops.weight_levels(W2, scale, wq.bits)
    out: Dict[str, torch.Tensor] = {}
    if wrapper.split:
        levels = torch.cat((torch.zeros_like(levels[:, :1]), levels), dim=1).contiguous()
        out["w0"] = wrapper.L1.weight.data.reshape(-1).float().cpu()
    N, K = levels.shape
    if wq.bits == 4:
        if K % 2:
            levels = torch.cat((levels, torch.zeros_like(levels[:, :1])), dim=1).contiguous()
        out["qweight"] = ops.pack_i4(levels).cpu()
    else:
        out["qweight"] = levels.cpu()
    out["w_scale"] = scale.cpu()
    s0 = float(qz.quantizer.scale)
    s1 = s0
    if getattr(qz, "msq", False) and qz.quantizer_text.scale is not None:
        s1 = float(qz.quantizer_text.scale)
    out["act_scale"] = torch.tensor([s0, s1], dtype=torch.float32)
    if wmod.bias is not None:
        out["bias"] = wmod.bias.data.float().cpu()
    mod = wrapper.module
    out["meta"] = _meta(version=VERSION, w_bits=wq.bits, a_bits=qz.bits, N=N, K=K,
                        in_features=K,
                        had_K=(wrapper.K if wrapper.online_full_had else 0), fp32_had=bool(wrapper.fp32_had),
                        split=bool(wrapper.split), msq=bool(getattr(qz, "msq", False)),
                        conv_rank=0 if isinstance(mod, torch.nn.Linear) else mod.weight.dim() - 2)
    return {k: v.contiguous() for k, v in out.items()}


def export_quantized(model, prefix: str = "") -> Dict[str, torch.Tensor]:
    from fake_quant import quant_utils as qu
    tensors: Dict[str, torch.Tensor] = {}
    for name, wrapper in qu.find_qlayers(model, layers=[qu.ActQuantWrapper]).items():
        for key, val in export_wrapper(wrapper).items():
            tensors[f"{prefix}{name}.{key}"] = val
    return tensors


def save_quantized(model, path: str, prefix: str = "") -> Dict[str, torch.Tensor]:
    from safetensors.torch import save_file
    tensors = export_quantized(model, prefix)
    save_file(tensors, path, metadata={"format": "mquant-w4a8", "version": str(VERSION)})
    return tensors


# ------------------------------------------------------------------------------------------ import
def build_linear(rec: Dict[str, torch.Tensor], device):
    """One record (keys without the module prefix) -> frozen ``W4A8Linear`` on ``device``."""
    from fake_quant import hadamard_utils
    from . import ops
    from .engine import HadamardSpec, W4A8Linear
    m = read_meta(rec["meta"])
    q = rec["qweight"].to(device)
    levels = ops.unpack_i4(q)[:, :m["K"]].contiguous() if m["w_bits"] == 4 else q.to(torch.int8)
    had = None
    if m["had_K"]:
        had = HadamardSpec(m["K"], m["had_K"], hadamard_utils.had_sign_bits(m["had_K"], device) if m["had_K"] > 1 else None,
                           bool(m["fp32_had"]))
    s0, s1 = [float(v) for v in rec["act_scale"].tolist()]
    bias = rec["bias"].to(device) if "bias" in rec else None
    w0 = rec["w0"].to(device) if "w0" in rec else None
    return W4A8Linear(levels, rec["w_scale"].to(device), m["w_bits"], bias, s0, s1 if m["msq"] else None,
                      had=had, w0=w0, in_features=m["in_features"])


def split_records(tensors: Dict[str, torch.Tensor], prefix: str = "") -> Dict[str, Dict[str, torch.Tensor]]:
    recs: Dict[str, Dict[str, torch.Tensor]] = {}
    for key, val in tensors.items():
        if not key.startswith(prefix):
            continue
        mod, _, leaf = key[len(prefix):].rpartition(".")
        recs.setdefault(mod, {})[leaf] = val
    return {k: v for k, v in recs.items() if "meta" in v and "qweight" in v}


def load_linears(path_or_tensors, device, prefix: str = ""):
    """name -> ``W4A8Linear`` for every record of a file or tensor dict (no model needed)."""
    tensors = path_or_tensors
    if isinstance(path_or_tensors, str):
        from safetensors.torch import load_file
        tensors = load_file(path_or_tensors)
    return {name: build_linear(rec, device) for name, rec in split_records(tensors, prefix).items()}


def load_quantized(model, path_or_tensors, device, prefix: str = "", strict: bool = True) -> int:
    """Install the frozen engines into the ``ActQuantWrapper``s of ``model`` (same module names as
    at export).  The wrappers then run the real W4A8 path regardless of their float weights."""
    from fake_quant import quant_utils as qu
    engines = load_linears(path_or_tensors, device, prefix)
    wrappers = qu.find_qlayers(model, layers=[qu.ActQuantWrapper])
    missing = sorted(set(wrappers) - set(engines))
    extra = sorted(set(engines) - set(wrappers))
    if strict and (missing or extra):
        raise KeyError(f"quantized checkpoint does not match the model: missing {missing[:5]}, unexpected {extra[:5]}")
    for name, eng in engines.items():
        if name in wrappers:
            wrappers[name].install_real(eng)
    return len(engines)
