"""Flat export of a quantized model: packed int4 weights in the reference wire
format + scales, instead of pickling whole modules (reference exam/quant_qwen2vl.py:147-160 dumps
``torch.save(model)`` with fake-quantized fp16 weights -- 4x the bytes and tied to class paths).

One entry group per ``ActQuantWrapper`` (dotted module name ``<n>``), format version 2 (round 5: every
configuration the integer backend runs -- ``ActQuantWrapper._real_ready`` -- can be exported; version-1
files, static + symmetric only, still load):

    <n>.qweight         uint8 [N, K/2]  reference pack_i4 bytes (quant_utils.py:61-69: two's-complement
                                        nibbles, even k -> low nibble)            (w_bits == 4)
                        int8  [N, K]                                              (w_bits == 8)
                        K includes the zeroed column 0 of a ``split`` layer and the Hadamard padding.
                        Asymmetric weights (--w_asym) are stored as level - 2^(bits-1) (what the kernels multiply)
    <n>.w_scale         fp32 [N]        per-output-channel weight scale
    <n>.act_scale       fp32 [2]        static activation scales: [vision | all tokens, text tokens]
    <n>.bias            fp32 [N]        (optional)
    <n>.w0              fp32 [N]        (optional) the fp32 column of a ``split`` layer (L1)
    <n>.w_shift         fp32 [N]        (optional, --w_asym) s_w (2^(bits-1) - z_w): the rank-1 epilogue factor of the zero points
    <n>.w_group_scales  fp32 [K/g, N]   (optional, --w_groupsize g) one scale per (group, channel)
    <n>.col_perm        int64 [K]       (optional, --act_order with --w_groupsize) image column j = input channel col_perm[j]
    <n>.act_clip        fp32 [1]        (dynamic activation modes) clip_ratio
    <n>.meta            int64 [20]      see META below (dynamic modes: act_mode = 1 and a_sym / a_per_tensor / a_groupsize)

Everything else of the model (norms, embeddings, lm_head) stays in its ordinary ``state_dict``.
Files are safetensors.  ``load_quantized`` rebuilds the frozen ``W4A8Linear`` of every wrapper
without touching (or needing) floating-point weights.
"""
from __future__ import annotations

from typing import Dict

import torch

META_V1 = ("version", "w_bits", "a_bits", "N", "K", "in_features", "had_K", "fp32_had", "split", "msq",
           "conv_rank", "reserved")
META = META_V1[:-1] + ("act_mode", "a_sym", "a_per_tensor", "a_groupsize", "split_slice", "w_groupsize", "w_asym",
                       "reserved0", "reserved1")
VERSION = 2
assert len(META) == 20


def _meta(**kw) -> torch.Tensor:
    names = META_V1 if int(kw.get("version", VERSION)) == 1 else META
    return torch.tensor([int(kw.get(k, 0)) for k in names], dtype=torch.int64)


def read_meta(t: torch.Tensor) -> Dict[str, int]:
    vals = [int(v) for v in t.reshape(-1).tolist()]
    if len(vals) == len(META_V1) and vals[0] == 1:          # version 1: static activations, symmetric per-channel weights
        m = dict(zip(META_V1, vals))
        m.update({k: 0 for k in META if k not in m})
        m["a_groupsize"] = -1
        m["a_sym"] = 1
        return m
    if len(vals) != len(META) or vals[0] != VERSION:
        raise ValueError(f"unsupported quantized-layer record (meta = {vals})")
    return dict(zip(META, vals))


# ------------------------------------------------------------------------------------------ export
def export_wrapper(wrapper, device=None) -> Dict[str, torch.Tensor]:
    """Tensors of one weight-quantized ``ActQuantWrapper`` whose configuration the integer backend runs (CPU, contiguous):
    exactly the parts ``W4A8Linear`` is built from (``ActQuantWrapper._real_parts``), so what loads is what ran."""
    from . import ops
    why = "activation quantizer not configured" if wrapper.quantizer.bits >= 16 else wrapper._simulated_because()
    if why:
        raise ValueError("export needs a wrapper the integer backend runs (calibrated static or dynamic activation quantizer, "
                         f"attached WeightQuantizer); this one simulates: {why}")
    name, wmod = wrapper._weight_module()
    wq = wrapper.weight_quantizers[name]
    qz = wrapper.quantizer
    device = device or wmod.weight.device
    p = wrapper._real_parts(torch.device(device))
    levels = p["levels"]
    out: Dict[str, torch.Tensor] = {}
    if p["w0"] is not None:
        out["w0"] = p["w0"].float().cpu()
    N, K = levels.shape
    if wq.bits == 4:
        if K % 2:
            levels = torch.cat((levels, torch.zeros_like(levels[:, :1])), dim=1).contiguous()
        out["qweight"] = ops.pack_i4(levels.contiguous()).cpu()
    else:
        out["qweight"] = levels.cpu()
    out["w_scale"] = p["scale"].reshape(-1).float().cpu()
    s0 = float(p["s0"])
    out["act_scale"] = torch.tensor([s0, s0 if p["s1"] is None else float(p["s1"])], dtype=torch.float32)
    if p["bias"] is not None:
        out["bias"] = p["bias"].float().cpu()
    if p["w_shift"] is not None:
        out["w_shift"] = p["w_shift"].float().cpu()
    if p["w_groups"] is not None:
        out["w_group_scales"] = p["w_groups"][0].float().cpu()
    if p.get("col_perm") is not None:
        out["col_perm"] = p["col_perm"].to(torch.int64).cpu()
    dyn = p["dynamic"]
    if dyn is not None:
        out["act_clip"] = torch.tensor([float(dyn["clip_ratio"])], dtype=torch.float32)
    mod = wrapper.module
    out["meta"] = _meta(version=VERSION, w_bits=wq.bits, a_bits=(qz.bits if dyn is None else dyn["bits"]), N=N, K=K,
                        in_features=K + (1 if p["split_slice"] else 0),
                        had_K=(wrapper.K if wrapper.online_full_had else 0), fp32_had=bool(wrapper.fp32_had),
                        split=bool(wrapper.split), msq=bool(p["s1"] is not None),
                        conv_rank=0 if isinstance(mod, torch.nn.Linear) else mod.weight.dim() - 2,
                        act_mode=0 if dyn is None else 1, a_sym=1 if dyn is None else int(bool(dyn["sym"])),
                        a_per_tensor=0 if dyn is None else int(bool(dyn["per_tensor"])),
                        a_groupsize=-1 if dyn is None else int(dyn["groupsize"]),
                        split_slice=bool(p["split_slice"]), w_groupsize=0 if p["w_groups"] is None else int(p["w_groups"][1]),
                        w_asym=p["w_shift"] is not None)
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
        n = m["K"] + (1 if m["split_slice"] else 0)
        had = HadamardSpec(n, m["had_K"], hadamard_utils.had_sign_bits(m["had_K"], device) if m["had_K"] > 1 else None,
                           bool(m["fp32_had"]))
    s0, s1 = [float(v) for v in rec["act_scale"].tolist()]
    bias = rec["bias"].to(device) if "bias" in rec else None
    w0 = rec["w0"].to(device) if "w0" in rec else None
    dynamic = None
    if m["act_mode"] == 1:
        dynamic = dict(bits=m["a_bits"], clip_ratio=float(rec["act_clip"][0]), sym=bool(m["a_sym"]),
                       per_tensor=bool(m["a_per_tensor"]), groupsize=m["a_groupsize"])
    w_shift = rec["w_shift"].to(device) if "w_shift" in rec else None
    w_groups = (rec["w_group_scales"].to(device).contiguous(), m["w_groupsize"]) if m["w_groupsize"] > 0 else None
    col_perm = rec["col_perm"].to(device=device, dtype=torch.long) if "col_perm" in rec else None
    return W4A8Linear(levels, rec["w_scale"].to(device), m["w_bits"], bias, s0, s1 if m["msq"] else None,
                      had=had, w0=w0, in_features=m["in_features"], dynamic=dynamic, w_shift=w_shift,
                      split_slice=bool(m["split_slice"]), w_groups=w_groups, col_perm=col_perm)


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
