"""Offline weight surgery shared by the per-model rotation passes: LayerNorm fusion, mean
baking and orthogonal rotations, all evaluated in fp64 and cast back to the weight's dtype
(reference: ``fake_quant/rotation_utils.py``).

Conventions (x is a row vector, Linear computes x W^T + b):
    rotate the INPUT  space of a Linear by Q:   W <- W Q            (``rotate_linear_input_``)
    rotate the OUTPUT space of a Linear by Q:   W <- Q^T W, b <- Q^T b  (``rotate_linear_output_``)
so that  (x Q) (W Q)^T = x W^T  and  (x W^T + b) Q = x (Q^T W)^T + Q^T b.
"""
import typing

import torch

from fake_quant import module_util, utils
from fake_quant.hadamard_utils import (  # noqa: F401  (re-exported like upstream)
    apply_exact_had_to_linear,
    is_pow2,
    random_hadamard_matrix,
)


# ------------------------------------------------------------------------------- primitives
def _as64(t: torch.Tensor, like: torch.Tensor = None) -> torch.Tensor:
    t = t.to(torch.float64)
    return t if like is None else t.to(like.device)


def rotate_linear_input_(linear, Q: torch.Tensor) -> None:
    W = linear.weight.data
    linear.weight.data = (W.double() @ _as64(Q, W)).to(W.dtype)


def rotate_linear_output_(linear, Q: torch.Tensor) -> None:
    W = linear.weight.data
    Qt = _as64(Q, W).T
    linear.weight.data = (Qt @ W.double()).to(W.dtype)
    if linear.bias is not None:
        linear.bias.data = (Qt @ linear.bias.data.double()).to(W.dtype)


def rotate_grouped_input_(linear, Q: torch.Tensor) -> None:
    """Input features are several consecutive vectors of size |Q| (e.g. the 2x2 patch merger):
    every group is rotated by the same Q."""
    W = linear.weight.data
    out_f, in_f = W.shape
    g = Q.shape[0]
    linear.weight.data = (W.double().reshape(out_f, -1, g) @ _as64(Q, W)).to(W.dtype).reshape(out_f, in_f).contiguous()


def rotate_vector_(param: torch.Tensor, Q: torch.Tensor) -> None:
    """param <- param Q for embedding tables / positional tensors whose last dim is the model dim."""
    param.data = (param.data.double() @ _as64(Q, param.data)).to(param.data.dtype)


# ------------------------------------------------------------------------------- LayerNorm fusion
def _fold_norm(layernorm, linear_layers, grouped: bool) -> None:
    gamma = layernorm.weight.double()
    has_bias_attr = getattr(layernorm, "bias", None) is not None    # RMSNorm: gamma only
    beta = layernorm.bias.double() if has_bias_attr else None
    for lin in linear_layers:
        dt = lin.weight.dtype
        W = lin.weight.data.double()
        if grouped:
            out_f, in_f = W.shape
            Wg = W.view(out_f, -1, gamma.shape[0])
            lin.weight.data = (Wg * gamma).to(dt).view(out_f, in_f)
            shift = (Wg @ beta).sum(dim=-1) if beta is not None else None
        else:
            lin.weight.data = (W * gamma).to(dt)
            shift = W @ beta if beta is not None else None
        if has_bias_attr:
            if lin.bias is None:
                lin.bias = torch.nn.Parameter(torch.zeros(lin.out_features, dtype=torch.float64).to(W))
            lin.bias.data = (lin.bias.data.double() + shift).to(dt)
    layernorm.weight.data = torch.ones_like(layernorm.weight.data)
    if has_bias_attr:
        layernorm.bias.data = torch.zeros_like(layernorm.bias.data)


def fuse_ln_linear(layernorm: torch.nn.Module, linear_layers: typing.Iterable[torch.nn.Linear]) -> None:
    """Fold a norm's affine part (gamma, beta) into the Linears that consume its output:
    W <- W diag(gamma), b <- b + W beta; the norm is left with gamma = 1, beta = 0."""
    _fold_norm(layernorm, linear_layers, grouped=False)


def fuse_merger_linear(layernorm: torch.nn.Module, linear_layers: typing.Iterable[torch.nn.Linear]) -> None:
    """Same for a Linear whose input concatenates several normalised vectors (Qwen2-VL merger)."""
    _fold_norm(layernorm, linear_layers, grouped=True)


def bake_mean_into_conv(conv) -> None:
    """Make the convolution emit zero-mean features (over output channels), i.e. absorb the mean
    subtraction of the following LayerNorm."""
    dt = conv.weight.dtype
    W = conv.weight.data.double()
    conv.weight.data = (W - W.mean(dim=0, keepdim=True)).to(dt)
    if conv.bias is not None:
        b = conv.bias.data.double()
        conv.bias.data = (b - b.mean()).to(dt)


def bake_mean_into_linear(linear: torch.nn.Linear) -> None:
    dt = linear.weight.dtype
    W = linear.weight.data.double()
    linear.weight.data = (W - W.mean(dim=-2, keepdim=True)).to(dt)
    if linear.bias is not None:
        b = linear.bias.data.double()
        linear.bias.data = (b - b.mean()).to(dt)


# ------------------------------------------------------------------------------- rotations
def random_orthogonal_matrix(size, device):
    """QR of a Gaussian matrix with the sign convention diag(R) > 0 (Haar distributed), fp64."""
    if torch.cuda.is_available():
        torch.cuda.empty_cache()
    q, r = torch.linalg.qr(torch.randn(size, size, dtype=torch.float64).to(device))
    q *= torch.sign(torch.diag(r)).unsqueeze(0)
    return q


def get_orthogonal_matrix(size, mode, device=utils.DEV):
    if mode == "random":
        return random_orthogonal_matrix(size, device)
    if mode == "hadamard":
        return random_hadamard_matrix(size, device)
    raise ValueError(f"Unknown mode {mode}")


def rotate_conv(layer, Q_v, embed_dims) -> None:
    """Rotate the OUTPUT channels of a patch-embedding convolution."""
    W = layer.weight.data
    layer.weight.data = (_as64(Q_v, W).T @ W.double().view(embed_dims, -1)).to(W.dtype).view(W.shape)
    if layer.bias is not None:
        layer.bias.data = (layer.bias.data.double() @ _as64(Q_v, W)).to(W.dtype)


def rotate_value_output_heads_(v_weight, v_bias, o_proj, Q_head, head_num, head_dim):
    """Per-head rotation of the V projection's outputs and the O projection's inputs by the same
    head_dim x head_dim matrix.  Returns the new (v_weight, v_bias)."""
    dt = v_weight.dtype
    Qh = _as64(Q_head, v_weight)
    Wv = v_weight.double().T.reshape(-1, head_num, head_dim)
    v_weight = (Wv @ Qh).reshape(-1, head_num * head_dim).T.to(dt)
    if v_bias is not None:
        v_bias = (v_bias.double().reshape(head_num, head_dim) @ Qh).to(dt).reshape(-1)
    Wo = o_proj.weight.data.double().reshape(-1, head_num, head_dim)
    o_proj.weight.data = (Wo @ Qh).reshape(-1, head_num * head_dim).to(dt)
    return v_weight, v_bias


def pad_linear_inputs_(model, match: str, new_in: int) -> int:
    """Replace every nn.Linear whose dotted name contains ``match`` by one with ``new_in`` input
    features.  The original columns are copied, the pad columns are ZERO and the new layer has a
    bias only if the old one had (upstream leaves both at their random init)."""
    count = 0
    modules = dict(model.named_modules())
    for name, mod in list(modules.items()):
        if match in name and isinstance(mod, torch.nn.Linear) and mod.in_features != new_in:
            new = torch.nn.Linear(new_in, mod.out_features, bias=mod.bias is not None,
                                  dtype=mod.weight.dtype, device=mod.weight.device)
            with torch.no_grad():
                new.weight.zero_()
                new.weight[:, : mod.in_features] = mod.weight.data
                if mod.bias is not None:
                    new.bias.copy_(mod.bias.data)
            parent_name, _, leaf = name.rpartition(".")
            setattr(modules[parent_name] if parent_name else model, leaf, new)
            count += 1
    return count


__all__ = [n for n in dir() if not n.startswith("_")] + ["module_util"]
