"""Operator layer of the W4A8 static-quant path, with the reference's public surface
(``fake_quant/quant_utils.py``): ``ActQuantizer``, ``ActQuantWrapper``, ``WeightQuantizer``,
``add_actquant``, ``find_qlayers``, the ``model_*`` calibration toggles and the ``calib_*``
drivers.  Callers (``exam/quant_*.py``, the GPTQ/RTN passes) use these names unchanged.

What is different underneath: once a wrapper is calibrated (``model_quant``) and its weights
carry integer levels (the RTN/GPTQ passes attach their ``WeightQuantizer``), ``forward`` no
longer simulates quantization in floating point.  It runs two hand-written gfx950 kernels
through the C ABI in ``include/mquant_hip.h``:

    [zero-pad + online Hadamard +] static int8 quantize  ->  int8 x int4 MFMA GEMM with
    fused per-channel dequant (+ bias, + the fp32 rank-1 ``split`` term)

There is no CPU fallback for that path: a CPU tensor or a missing ``libmquant_hip.so``
raises ``MQuantHipError``.  Calibration, module surgery and the dynamic (per-token)
quantizers -- which no canonical command line uses -- stay torch code.

Extension: Modality-Specific Static Quantization.  ``ActQuantizer.configure(..., msq=True)``
keeps TWO static scale sets per layer (vision tokens / text tokens); the token-type mask of
the running batch is published with ``token_type_mask(mask)`` and consumed inside the
kernels (``row_sel``).
"""
from __future__ import annotations

import contextlib
import functools
import math
from collections import OrderedDict
from typing import Optional

import torch

from fake_quant import hadamard_utils, utils
from fake_quant.bit_type import BIT_TYPE_DICT
from fake_quant.observer import build_observer
from fake_quant.quantizer import build_quantizer


# =============================================================================== integer grids
def get_minq_maxq(bits, sym):
    """(minq, maxq): symmetric -> [-2^(b-1), 2^(b-1)-1]; asymmetric -> [0, 2^b-1]."""
    if sym:
        maxq = torch.tensor(2 ** (bits - 1) - 1)
        return -maxq - 1, maxq
    return 0, torch.tensor(2 ** bits - 1)


def asym_quant(x, scale, zero, maxq):
    scale, zero = scale.to(x.device), zero.to(x.device)
    return torch.clamp(torch.round(x / scale) + zero, 0, maxq), scale, zero


def asym_dequant(q, scale, zero):
    return scale * (q - zero)


def asym_quant_dequant(x, scale, zero, maxq):
    return asym_dequant(*asym_quant(x, scale, zero, maxq))


def sym_quant(x, scale, maxq):
    scale = scale.to(x.device)
    return torch.clamp(torch.round(x / scale), -(maxq + 1), maxq), scale


def sym_dequant(q, scale):
    return scale * q


def sym_quant_dequant(x, scale, maxq):
    return sym_dequant(*sym_quant(x, scale, maxq))


def two_compl(x, bits: int):
    return torch.where(x < 0, 2 ** bits + x, x)


def pack_i4(q):
    """Signed int4 levels -> uint8, two per byte along the last dim: even index in the low
    nibble, odd index in the high nibble (two's complement).  The wire format."""
    assert torch.is_signed(q), "The tensor to be packed should be signed int"
    minq, maxq = get_minq_maxq(4, True)
    assert torch.all(torch.logical_and(q >= minq, q <= maxq))
    if q.is_cuda:
        from mquant_amd import ops
        return ops.pack_i4(q)
    nib = two_compl(q.to(torch.int8), 4).to(torch.uint8)
    return nib[..., 0::2] | (nib[..., 1::2] << 4)


def unpack_i4(x: torch.Tensor):
    """Inverse of ``pack_i4``; returns int32 like upstream."""
    assert x.dtype == torch.uint8, "The tensor to be unpacked should be stored in uint8"
    if x.is_cuda:
        from mquant_amd import ops
        return ops.unpack_i4(x).to(torch.int32)
    lo = (x & 0x0F).to(torch.int32)
    hi = (x >> 4).to(torch.int32)
    both = torch.stack((lo, hi), dim=-1)
    both = torch.where(both >= 8, both - 16, both)
    return both.reshape(*x.shape[:-1], x.shape[-1] * 2)


# =============================================================================== MSQ mask
_MSQ_STATE = {"mask": None}
_ALL_TEXT_MASKS = {}


@contextlib.contextmanager
def token_type_mask(mask: Optional[torch.Tensor]):
    """Publish the token-type mask of the running batch: one entry per flattened token,
    0 = vision token (scale set 0), 1 = text token (scale set 1)."""
    prev = _MSQ_STATE["mask"]
    _MSQ_STATE["mask"] = None if mask is None else mask.reshape(-1).to(torch.uint8).contiguous()
    try:
        yield
    finally:
        _MSQ_STATE["mask"] = prev


def set_token_type_mask(mask: Optional[torch.Tensor]) -> None:
    _MSQ_STATE["mask"] = None if mask is None else mask.reshape(-1).to(torch.uint8).contiguous()


def _row_mask(rows: int, device) -> torch.Tensor:
    """Mask for a [rows, C] activation.  Without a published mask of matching length every
    row counts as text: decode steps append text tokens only."""
    m = _MSQ_STATE["mask"]
    if m is not None and m.numel() == rows:
        return m.to(device)
    key = (rows, str(device))
    ones = _ALL_TEXT_MASKS.get(key)
    if ones is None:                 # one tensor per (rows, device): sibling wrappers see the SAME mask object
        if len(_ALL_TEXT_MASKS) > 64:
            _ALL_TEXT_MASKS.clear()
        ones = _ALL_TEXT_MASKS[key] = torch.ones(rows, dtype=torch.uint8, device=device)
    return ones


# =============================================================================== ActQuantizer
class ActQuantizer(torch.nn.Module):
    """Activation quantizer.

    static=True  : calibrated scale(s) from an observer; ``calibrate`` / ``last_calibrate`` /
                   ``quant`` flags drive the protocol (open -> N forwards -> last -> close ->
                   quant).  During calibration activations pass through unquantized.
    static=False : dynamic per-token (default), per-tensor (``act_per_tensor``) or group-wise
                   (``groupsize``) ranges found by ``find_params`` on every call.
    bits == 16   : identity.
    """

    def __init__(self, act_per_tensor=False):
        super().__init__()
        self.register_buffer("maxq", torch.tensor(0))
        self.register_buffer("scale", torch.zeros(1))
        self.register_buffer("zero", torch.zeros(1))
        self.bits = 16
        self.act_per_tensor = act_per_tensor
        self.static = False
        self.msq = False

    def __setattr__(self, name, value):
        # every attribute write bumps a version counter: ActQuantWrapper.forward keeps its "the integer backend runs this
        # configuration" decision only while the counter stands still (flags are flipped from outside: model_quant & co.)
        super().__setattr__(name, value)
        self.__dict__["_ver"] = self.__dict__.get("_ver", 0) + 1

    def free(self):
        self.zero = None
        self.scale = None

    def configure(self, bits, groupsize=-1, sym=False, clip_ratio=1.0, act_per_tensor=False,
                  static=False, observer_type="minmax", calibration_mode="layer_wise",
                  msq=False):
        _, self.maxq = get_minq_maxq(bits, sym)
        self.bits = bits
        self.groupsize = groupsize
        self.sym = sym
        self.clip_ratio = clip_ratio
        self.act_per_tensor = act_per_tensor
        assert 0 < self.clip_ratio <= 1, "Clip ratio should be in (0, 1]"
        self.static = static
        self.msq = bool(msq) and static
        if static:
            bit_type = BIT_TYPE_DICT[f"int{bits}"]       # KeyError for bits=4, as upstream
            if observer_type == "percentile":
                print("Using percentile observer for activations")
            self.observer = build_observer(observer_type, "activation", bit_type, calibration_mode)
            self.quantizer = build_quantizer("uniform", bit_type, self.observer, "activation")
            if self.msq:
                self.observer_text = build_observer(observer_type, "activation", bit_type,
                                                    calibration_mode)
                self.quantizer_text = build_quantizer("uniform", bit_type, self.observer_text,
                                                      "activation")
            self.calibrate = False
            self.last_calibrate = False
            self.quant = False

    # ---- static path ---------------------------------------------------------------------
    def _observe(self, x):
        if not self.msq:
            self.quantizer.observer.update(x)
            if self.last_calibrate:
                self.quantizer.update_quantization_params(x)
            return
        rows = x.reshape(-1, x.shape[-1])
        text = _row_mask(rows.shape[0], x.device).bool()
        for qz, part in ((self.quantizer, rows[~text]), (self.quantizer_text, rows[text])):
            if part.shape[0]:
                qz.observer.update(part)
            if self.last_calibrate and qz.observer.max_val is not None:
                qz.update_quantization_params(part)

    def _static_fakequant(self, x):
        if not self.msq:
            return self.quantizer(x)
        from mquant_amd import ops
        s0 = self.quantizer.scale if self.quantizer.scale is not None else self.quantizer_text.scale
        s1 = self.quantizer_text.scale if self.quantizer_text.scale is not None else s0
        rows = x.reshape(-1, x.shape[-1])
        return ops.fakequant_act(rows, float(s0), float(s1),
                                 row_sel=_row_mask(rows.shape[0], x.device)).reshape(x.shape)

    def forward(self, x):
        if self.static:
            if self.calibrate:
                self._observe(x)
                return x
            return self._static_fakequant(x) if self.quant else x
        if self.bits == 16:
            return x
        x_dtype = x.dtype
        if self.sym:
            return sym_quant_dequant(x, self.scale, self.maxq).to(x_dtype)
        return asym_quant_dequant(x, self.scale, self.zero, self.maxq).to(x_dtype)

    def quantize(self, x):
        """Integers + scale (+ zero) instead of the dequantized tensor."""
        if self.sym:
            return sym_quant(x, self.scale, self.maxq)
        return asym_quant(x, self.scale, self.zero, self.maxq)

    # ---- dynamic range search ------------------------------------------------------------
    def _ranges(self, xmin, xmax):
        """scale / zero from clipped min & max tensors (any shape); degenerate ranges -> 1."""
        if self.sym:
            amax = torch.maximum(torch.abs(xmin), xmax)
            scale = amax / self.maxq
            scale = torch.where(amax == 0, torch.ones_like(scale), scale)
            return scale, torch.zeros_like(scale)
        dead = (xmin == 0) & (xmax == 0)
        xmin = torch.where(dead, -torch.ones_like(xmin), xmin)
        xmax = torch.where(dead, torch.ones_like(xmax), xmax)
        scale = (xmax - xmin) / self.maxq
        return scale, torch.round(-xmin / scale)

    def find_params_per_token_groupwise(self, x):
        shape = x.shape
        g = x.reshape(-1, x.shape[-2], x.shape[-1] // self.groupsize, self.groupsize)
        xmax = torch.amax(g, dim=3, keepdim=True) * self.clip_ratio
        xmin = torch.amin(g, dim=3, keepdim=True) * self.clip_ratio
        scale, zero = self._ranges(xmin, xmax)
        self.scale = scale.expand(-1, -1, -1, self.groupsize).reshape(shape)
        self.zero = zero.expand(-1, -1, -1, self.groupsize).reshape(shape)

    def find_params(self, x):
        if self.bits == 16:
            return
        self.maxq = self.maxq.to(x.device)
        shape = x.shape
        if self.act_per_tensor:
            z = torch.zeros((), dtype=x.dtype, device=x.device)
            xmin = torch.minimum(x.min(), z) * self.clip_ratio
            xmax = torch.maximum(x.max(), z) * self.clip_ratio
            if self.sym:
                amax = torch.maximum(torch.abs(xmin), xmax)
                self.scale = 1 if amax == 0 else amax / self.maxq
                self.zero = torch.zeros_like(amax)
            else:
                xmin = -1 if xmin == 0 else xmin     # each end is patched on its own upstream
                xmax = 1 if xmax == 0 else xmax
                self.scale = (xmax - xmin) / self.maxq
                self.zero = torch.round(-xmin / self.scale)
            return
        if self.groupsize > 0:
            self.find_params_per_token_groupwise(x)
            utils.cleanup_memory(verbos=False)
            return
        rows = x.reshape(-1, shape[-1])
        z = torch.zeros(rows.shape[0], device=x.device)
        xmin = torch.minimum(rows.min(1)[0], z) * self.clip_ratio
        xmax = torch.maximum(rows.max(1)[0], z) * self.clip_ratio
        scale, zero = self._ranges(xmin, xmax)
        self.scale = scale.unsqueeze(1).expand(-1, shape[-1]).reshape(shape)
        self.zero = zero.unsqueeze(1).expand(-1, shape[-1]).reshape(shape)


# =============================================================================== ActQuantWrapper
_REAL_DTYPES = (torch.float16, torch.bfloat16, torch.float32)


class _NullHandle:
    def remove(self):
        pass


class ActQuantWrapper(torch.nn.Module):
    """Wraps ``nn.Linear | Conv2d | Conv3d``: [pad ->] online Hadamard -> activation
    quantizer -> the wrapped module.  ``split`` keeps channel 0 in floating point (L1) and
    quantizes the rest (L2).

    Attribute and sub-module names (``module``, ``L1``, ``L2``, ``quantizer``,
    ``out_quantizer``, ``had_K``, ``K``, ``online_full_had``, ``fp32_had``, ``split`` ...)
    are part of the contract: GPTQ matches them by string and pickled checkpoints bake in
    the import path ``fake_quant.quant_utils.ActQuantWrapper``.
    """

    def __init__(self, module: torch.nn.Linear, act_per_tensor=False):
        super().__init__()
        assert isinstance(module, (torch.nn.Linear, torch.nn.Conv2d, torch.nn.Conv3d))
        self.module = module
        self.weight = module.weight
        self.bias = module.bias
        self.quantizer = ActQuantizer(act_per_tensor)
        self.out_quantizer = ActQuantizer(act_per_tensor)
        self.register_buffer("had_K", torch.tensor(0))
        self._buffers["had_K"] = None
        self.K = 1
        self.online_full_had = False
        self.online_partial_had = False
        self.had_dim = 0
        self.fp32_had = False
        self.split = False
        # real-integer backend state (not part of the upstream surface)
        self.real_quant = True          # set False to force the simulated path
        self.weight_quantizers = {}     # sub-module name ("module" / "L2") -> WeightQuantizer
        self.pad_to = None              # folded revise_down_input hook
        self._real = None
        self._real_frozen = False       # engine installed from a flat checkpoint (no float weights)
        self._group = None              # SiblingGroup: q/k/v, gate/up ... sharing one quantize + one GEMM
        self._fast = None               # cached "integer backend, plain nn.Linear" decision of forward (see _remember_fast)

    def __setattr__(self, name, value):
        # any attribute write (flags, sub-modules, engine, group) voids the cached forward decision
        super().__setattr__(name, value)
        if name != "_fast":
            self.__dict__["_fast"] = None

    # pickled checkpoints must not drag device handles along
    def __getstate__(self):
        state = self.__dict__.copy()
        state["_real"] = None
        state["_real_frozen"] = False
        state["_group"] = None
        state["_fast"] = None
        return state

    def extra_repr(self) -> str:
        def kind(qz):
            if qz.bits >= 16:
                return ""
            if getattr(qz, "static", False):
                return " (Static Per-Tensor%s)" % (", MSQ" if getattr(qz, "msq", False) else "")
            return " (Symmetric Per-Token)" if qz.sym else " (Asymmetric Per-Token)"
        return (f"Input Quantizer Bits: {self.quantizer.bits}{kind(self.quantizer)}\n"
                f"Output Quantizer Bits: {self.out_quantizer.bits}{kind(self.out_quantizer)}\n"
                f"Backend: {self.backend()}")

    def backend(self) -> str:
        """Which evaluation a quantized forward of this wrapper takes: the hand-written integer kernels ("W4A8 integer", with
        the entry point family) or the reference's simulated torch ops -- and then why (a model that falls back must
        say so; ``print(model)`` shows this line for every wrapper)."""
        qz = self.quantizer
        if getattr(self, "_real_frozen", False) and self._real is not None:
            if not self.real_quant:
                return "simulated (torch ops): real_quant switched off"
            return "W%dA%d integer (engine from a flat checkpoint)" % (self._real.w_bits, 8 if self._real.dynamic is None else self._real.dynamic["bits"])
        if qz.bits >= 16:
            return "float (activation quantizer not configured)"
        why = self._simulated_because()
        if why:
            return "simulated (torch ops): " + why
        name, _ = self._weight_module()
        wq = self.weight_quantizers[name]
        mode = "static" if qz.static else "dynamic"
        if getattr(wq, "group_scales", None) is not None:
            mode += ", weight groups of %d" % wq.groupsize
        return "W%dA%d integer (%s)" % (wq.bits, qz.bits, mode)

    def _simulated_because(self) -> str:
        """'' when the integer backend can run this wrapper's configuration, else the first reason it cannot (dtype and
        convolution-geometry checks that need the input are ``_real_ready``'s)."""
        qz = self.quantizer
        if not self.real_quant:
            return "real_quant switched off"
        if self.out_quantizer.bits < 16:
            return "output quantizer configured"
        if self.online_partial_had:
            return "online partial Hadamard"
        if qz.static:
            if qz.bits != 8:
                return "static activations other than int8"
            if qz.calibrate or not qz.quant:
                return "static quantizer not calibrated / not switched on (model_quant)"
            if qz.quantizer.scale is None or qz.quantizer.scale.numel() != 1:
                return "channel_wise static scales"
        elif not self._dynamic_real_ok():
            return "dynamic activation mode outside the kernels (bits, group size, split or asymmetric weights with groups)"
        name, _ = self._weight_module()
        wq = self.weight_quantizers.get(name)
        if wq is None:
            return "no weight quantizer attached (run the RTN / GPTQ pass of fake_quant.gptq)"
        if wq.bits not in (4, 8):
            return "weight bits %d" % wq.bits
        # (per-tensor weight quantizers -- perchannel=False, no driver uses them -- repeat their one scale / zero point per output
        #  channel, quant_utils.py:507-509 upstream: they take the per-channel path unchanged, symmetric or not)
        if getattr(wq, "groupsize", -1) and getattr(wq, "groupsize", -1) > 0:
            return self._weight_groups_because(wq)
        return ""

    def _weight_groups_because(self, wq) -> str:
        """--w_groupsize on the integer path (mq_gemm_w4a8_wgroupscale): contiguous groups of 64 or a multiple of 128 input
        channels, symmetric levels, no split column; static, dynamic per-token symmetric or same-size group-wise activations."""
        qz = self.quantizer
        g = int(wq.groupsize)
        if getattr(wq, "group_scales", None) is None:
            return "weight group scales not recorded"
        if getattr(wq, "group_permuted", False):
            # --act_order: the groups are runs of PERMUTED columns; the engine gathers the activation columns the same way
            if getattr(wq, "group_perm", None) is None:
                return "weight groups over --act_order's permuted columns, permutation not recorded"
            if (not qz.static) and getattr(qz, "groupsize", -1) > 0:
                return "--act_order weight groups with group-wise activations (the activation groups would follow the permutation)"
        if not getattr(wq, "sym", False):
            return "asymmetric weights with --w_groupsize"
        if self.split:
            return "--w_groupsize with the split column"
        _, wmod = self._weight_module()
        width = wmod.weight.shape[1] if wmod.weight.dim() == 2 else 0
        if not ((g == 64 or g % 128 == 0) and g <= 1024 and (g & (g - 1)) == 0 and width > 0 and width % g == 0):
            return "weight group size %d (64 or a power of two from 128 to 1024 that divides the input width)" % g
        if not qz.static:
            ag = getattr(qz, "groupsize", -1)
            if not getattr(qz, "sym", False) or getattr(qz, "act_per_tensor", False):
                return "--w_groupsize with asymmetric or per-tensor dynamic activations"
            if ag > 0 and ag != g:
                return "--a_groupsize %d differs from --w_groupsize %d" % (ag, g)
        return ""

    def register_forward_pre_hook(self, hook, *args, **kwargs):
        """The drivers pad ``down_proj`` inputs with a ``revise_down_input`` pre-hook.  That
        hook is folded into the wrapper (``pad_to``) so the fused kernel can read the unpadded
        activations; any other hook is registered normally."""
        if isinstance(hook, functools.partial) and hook.func is utils.revise_down_input:
            self.pad_to = hook.keywords.get("new_size", hook.args[0] if hook.args else None)
            return _NullHandle()
        return super().register_forward_pre_hook(hook, *args, **kwargs)

    def split_weights(self):
        """L1 = column 0, L2 = columns 1: of the wrapped weight (views until a weight pass
        rebinds ``.data``); L2 inherits the bias."""
        mod = self.module
        dev = mod.weight.device
        has_bias = mod.bias is not None
        self.L1 = torch.nn.Linear(1, mod.out_features, bias=False).to(dev)
        self.L2 = torch.nn.Linear(mod.in_features - 1, mod.out_features, bias=has_bias).to(dev)
        self.L1.weight.data = mod.weight.data[:, 0:1]
        self.L2.weight.data = mod.weight.data[:, 1:]
        if has_bias:
            self.L2.bias.data = mod.bias.data
        self._real = None

    # ------------------------------------------------------------------ real-integer backend
    def invalidate_real(self):
        self.__dict__["_fast"] = None
        if not getattr(self, "_real_frozen", False):
            self._real = None
        grp = self.__dict__.get("_group")
        if grp is not None:
            grp.reset()

    def install_real(self, engine) -> None:
        """Adopt an engine rebuilt from a flat checkpoint (mquant_amd.checkpoint.load_quantized):
        from now on ``forward`` runs it whatever the calibration flags or float weights say."""
        self._real = engine
        self._real_frozen = True
        self.quantizer.msq = engine.s_x1 is not None

    def _weight_module(self):
        return ("L2", self.L2) if self.split else ("module", self.module)

    def _real_ready(self, x) -> bool:
        qz = self.quantizer
        if getattr(self, "_real_frozen", False) and self._real is not None:
            return self.real_quant
        if qz.bits >= 16 or self._simulated_because():
            return False
        if x.dtype not in _REAL_DTYPES:
            return False
        mod = self.module
        if isinstance(mod, torch.nn.Linear):
            return True
        # a convolution is a GEMM when the kernel covers the whole (un-padded) input patch
        return (not self.split and not self.online_full_had and x.dim() == mod.weight.dim()
                and tuple(x.shape[2:]) == tuple(mod.kernel_size)
                and all(p == 0 for p in mod.padding) and mod.groups == 1)

    def _dynamic_real_ok(self, x_dtype=torch.float32) -> bool:
        """Dynamic per-token int8 (the reference's default activation mode, quant_utils.py:205-268) also
        has real-integer kernels: symmetric, and asymmetric (``--a_asym``; the zero point travels through a rank-1
        epilogue term like the split column -- two such terms fit, ``_real_ready`` counts them); per token,
        per tensor (``act_per_tensor``) or group-wise (``groupsize``: symmetric, groups of 64 / 128 / 256 ... channels)."""
        qz = self.quantizer
        if not (2 <= qz.bits <= 8):
            return False
        g = getattr(qz, "groupsize", -1)
        if g > 0 and not getattr(qz, "act_per_tensor", False):
            # group-wise scales: symmetric or asymmetric levels, a group = 64 or a multiple of 128 consecutive channels (the k-steps of
            # the GEMM), whole groups in the (padded) input width, no split column (the reference's reshape of the
            # K - 1 quantized columns fails there too) and symmetric weights (no rank-1 epilogue term in this kernel)
            name, wmod = self._weight_module()
            wq = self.weight_quantizers.get(name)
            width = wmod.weight.shape[1] if wmod.weight.dim() == 2 else 0
            return (not self.split and (g == 64 or g % 128 == 0) and g <= 1024
                    and (g & (g - 1)) == 0 and width > 0 and width % g == 0 and wq is not None and bool(getattr(wq, "sym", False)))
        # (per-tensor ranges on half / bf16 activations: the reference keeps range, scale, zero point, x / scale and the
        #  level sum in x's dtype -- quant_utils.py:214-231, ``torch.tensor(0).to(x)``, the int64 maxq does not promote -- and
        #  mq_quantize_act_range_i8 rounds exactly there; goldens wrapper_dynpt16_* from the reference on fp16 / bf16)
        return True

    def _real_parts(self, device) -> dict:
        """Everything the integer backend needs from this wrapper, on ``device``: weight levels and
        scales recovered from the fake-quantized weight and its attached ``WeightQuantizer``, bias, the
        split column, the Hadamard descriptor and the activation-quantizer parameters."""
        from mquant_amd import ops
        from mquant_amd.engine import HadamardSpec
        name, wmod = self._weight_module()
        wq = self.weight_quantizers[name]
        W = wmod.weight.data.to(device)
        W2 = W.reshape(W.shape[0], -1)
        scale = wq.scale.reshape(-1).to(device=device, dtype=torch.float32)
        if scale.numel() == 1:
            scale = scale.expand(W2.shape[0]).contiguous()
        w_shift = None
        w_groups = None
        col_perm = None
        if getattr(wq, "group_scales", None) is not None:
            # --w_groupsize: one scale per (channel, group of g consecutive input channels); levels on each group's own grid
            g = int(wq.groupsize)
            gs = wq.group_scales.to(device=device, dtype=torch.float32)                  # [N, K / g]
            assert gs.shape == (W2.shape[0], W2.shape[1] // g), "group scales do not match the weight"
            if getattr(wq, "group_permuted", False):
                # --act_order: group j holds columns perm[j g .. (j + 1) g - 1]; the image keeps the solver's column order
                col_perm = wq.group_perm.to(device=device, dtype=torch.long).contiguous()
                assert col_perm.numel() == W2.shape[1]
                W2 = W2.index_select(1, col_perm)
            half = 1 << (wq.bits - 1)
            levels = torch.round(W2.float().reshape(W2.shape[0], -1, g) / gs[:, :, None]).clamp(-half, half - 1)
            levels = levels.reshape(W2.shape).to(torch.int8)
            w_groups = (gs.t().contiguous(), g)                                          # [K / g, N] as the kernel reads it
            scale = gs.repeat_interleave(g, dim=1)                                       # per element, for the grid check below
        elif getattr(wq, "sym", False):
            levels = ops.weight_levels(W2, scale, wq.bits)
        else:
            # W~ = s_w (q - z_w), q in 0 .. 2^b - 1: rint(W~ / s_w) = q - z_w; stored level = q - 2^(b-1)
            half = 1 << (wq.bits - 1)
            zero = wq.zero.reshape(-1).to(device=device, dtype=torch.float32)
            if zero.numel() == 1:
                zero = zero.expand(W2.shape[0]).contiguous()
            diff = torch.round(W2.float() / scale[:, None])        # |q - z_w| <= 255: exact also for half-precision W~
            levels = (diff + (zero - float(half))[:, None]).clamp(-half, half - 1).to(torch.int8)
            w_shift = scale * (float(half) - zero)
        # the attached quantizer must be the one that produced these weights (one scale per output
        # channel): levels * scale has to give the stored fake-quantized weight back, else the wrapper
        # would silently run on a different integer grid (e.g. a group-wise GPTQ quantizer that only
        # remembers its last column group)
        # (any weight is within half a step of SOME level, so the test is the MEAN distance: ~0.25 steps for
        # a foreign grid, at most a few 1e-2 for half-precision roundings of scale * level, 0 in fp32)
        sc2 = scale if scale.dim() == 2 else scale[:, None]
        grid_w = levels.float() * sc2 if w_shift is None else levels.float() * sc2 + w_shift[:, None]
        dev_steps = ((grid_w - W2.float()).abs() / sc2).mean()
        if float(dev_steps) > 0.12:
            raise RuntimeError(f"ActQuantWrapper: weights of '{name}' are not on the attached quantizer's grid "
                               f"(mean distance {float(dev_steps):.3g} quantization steps); refusing to build the integer backend")
        w0 = None
        bias = wmod.bias
        qz = self.quantizer
        # asymmetric dynamic activations with the split column: the asymmetric quantizer kernels have no column to skip, so
        # the engine quantizes the view x[:, 1:] and keeps the levels of columns 1.. only (engine.W4A8Linear.split_slice)
        split_slice = bool(self.split) and (not qz.static) and not getattr(qz, "sym", False)
        if self.split:
            if not split_slice:
                levels = torch.cat((torch.zeros_like(levels[:, :1]), levels), dim=1).contiguous()
            w0 = self.L1.weight.data.to(device).reshape(-1).float()
        had = None
        if self.online_full_had:
            n = levels.shape[1] + (1 if split_slice else 0)
            had = HadamardSpec(n, self.K, hadamard_utils._bits_for(self.had_K, self.K, device),
                               bool(self.fp32_had))
        dynamic = None
        s0, s1 = 1.0, None
        if qz.static:
            s0 = float(qz.quantizer.scale)
            if qz.msq:
                s1 = float(qz.quantizer_text.scale) if qz.quantizer_text.scale is not None else s0
        else:
            dynamic = dict(bits=int(qz.bits), clip_ratio=float(qz.clip_ratio), sym=bool(qz.sym),
                           per_tensor=bool(qz.act_per_tensor),
                           groupsize=(int(qz.groupsize) if getattr(qz, "groupsize", -1) > 0 and not qz.act_per_tensor else -1))
        if w_groups is not None:
            scale = w_groups[0][-1].contiguous()                                         # (unused by the kernels; the last group's, as upstream keeps)
        return dict(levels=levels, scale=scale, bits=wq.bits,
                    bias=None if bias is None else bias.data.to(device), s0=s0, s1=s1,
                    had=had, w0=w0, dynamic=dynamic, w_shift=w_shift, split_slice=split_slice, w_groups=w_groups, col_perm=col_perm)

    def _build_real(self, device):
        from mquant_amd.engine import W4A8Linear
        p = self._real_parts(device)
        self._real = W4A8Linear(p["levels"], p["scale"], p["bits"], p["bias"], p["s0"], p["s1"],
                                had=p["had"], w0=p["w0"], dynamic=p["dynamic"], w_shift=p["w_shift"],
                                split_slice=p["split_slice"], w_groups=p["w_groups"], col_perm=p["col_perm"])
        return self._real

    def _forward_real(self, x, fast=None):
        if not x.is_cuda:
            from mquant_amd._lib import MQuantHipError
            raise MQuantHipError("ActQuantWrapper: the quantized W4A8 path runs on the GPU only; "
                                 "there is no CPU fallback (got a CPU tensor)")
        if fast is not None:
            # the cached decision: a plain nn.Linear on the integer backend (same launches as the general path below)
            msq, grp = fast[4], fast[5]
            two_d = x.dim() == 2
            rows = x if two_d else x.reshape(-1, x.shape[-1])
            sel = _row_mask(rows.shape[0], x.device) if msq else None
            if grp is not None and grp.enabled:
                y = grp.forward(self, rows, sel)
                if y is not None:
                    return y if two_d else y.reshape(*x.shape[:-1], y.shape[-1])
            real = self.__dict__["_real"]
            if real is None:
                real = self._build_real(x.device)
                self._remember_fast()               # (_build_real wrote self._real: the entry was voided)
            real.in_features = rows.shape[1]
            y = real.forward(rows, sel)
            return y if two_d else y.reshape(*x.shape[:-1], real.N)
        grp = self.__dict__.get("_group")
        if grp is not None and grp.enabled:
            rows = x.reshape(-1, x.shape[-1])
            sel = _row_mask(rows.shape[0], x.device) if getattr(self.quantizer, "msq", False) else None
            y = grp.forward(self, rows, sel)
            if y is not None:
                return y.reshape(*x.shape[:-1], y.shape[-1])
        real = self._real if self._real is not None else self._build_real(x.device)
        if isinstance(self.module, torch.nn.Linear):
            rows = x.reshape(-1, x.shape[-1])
            out_shape = (*x.shape[:-1], real.N)
        else:
            rows = x.reshape(x.shape[0], -1)
            out_shape = (x.shape[0], real.N) + (1,) * (x.dim() - 2)
        sel = _row_mask(rows.shape[0], x.device) if getattr(self.quantizer, "msq", False) else None
        real.in_features = rows.shape[1]           # bookkeeping (bytes read by the quantizer): the un-padded width
        return real.forward(rows, sel).reshape(out_shape)

    # ------------------------------------------------------------------ forward
    def _rotate(self, x, x_dtype):
        if self.pad_to is not None and x.shape[-1] < self.pad_to:
            x = torch.nn.functional.pad(x, (0, self.pad_to - x.shape[-1]))
        if self.online_full_had and not x.is_cuda and getattr(self, "simulate_on_cpu", False):
            # cpu_baseline opt-in: the reference's own CPU operator (hadamard_utils.py:79-100)
            x = hadamard_utils.matmul_hadU(x.float() if self.fp32_had else x).to(x_dtype)
        elif self.online_full_had:
            if self.fp32_had:
                x = hadamard_utils.matmul_hadU_cuda(x.float(), self.had_K, self.K).to(x_dtype)
            else:
                x = hadamard_utils.matmul_hadU_cuda(x, self.had_K, self.K)
        elif self.online_partial_had:
            shape = x.shape
            xf = x.float() if self.fp32_had else x
            heads = shape[-1] // self.had_dim
            v = xf.reshape(-1, heads, self.had_dim)
            if self.K == 1:   # Walsh-Hadamard across heads, for every within-head index
                v = hadamard_utils._fht_blocks(v.transpose(1, 2).contiguous(), heads).transpose(1, 2)
            else:
                v = (self.had_K.to(v.dtype) @ v) / math.sqrt(heads)
            x = v.to(x_dtype).reshape(shape) if self.fp32_had else v.reshape(shape)
        return x

    def _quantize_input(self, x, x_dtype):
        qz = self.quantizer
        if qz.static:
            return qz(x)
        if qz.bits < 16:
            qz.find_params(x)
            x = qz(x).to(x_dtype)
            qz.free()
        return x

    def _remember_fast(self):
        """Cache the decision "a quantized forward of this wrapper runs the integer backend" (plain nn.Linear only: the
        convolution check needs the input's shape).  ``_real_ready`` re-reads ~20 attributes through nn.Module's __getattr__ and
        rebuilds its reason strings on every call -- ~10 us of host time per Linear in an eager run
        (profiles/r5_decode_host_overhead.txt).  The entry dies with any attribute write on the wrapper, any attribute write on
        either activation quantizer (version counters), ``invalidate_real`` (weight passes, model_* toggles) and a dissolved
        sibling group."""
        if isinstance(self.module, torch.nn.Linear):
            qz, oq = self.quantizer, self.out_quantizer
            self.__dict__["_fast"] = (qz, qz.__dict__.get("_ver", 0), oq, oq.__dict__.get("_ver", 0),
                                      bool(getattr(qz, "msq", False)), self.__dict__.get("_group"))

    def fast_path_active(self) -> bool:
        """True while the cached forward decision is still valid (nothing written to the wrapper or its quantizers since)."""
        f = self.__dict__.get("_fast")
        return f is not None and f[0].__dict__.get("_ver", 0) == f[1] and f[2].__dict__.get("_ver", 0) == f[3]

    def forward(self, x):
        f = self.__dict__.get("_fast")
        if (f is not None and f[0].__dict__.get("_ver", 0) == f[1] and f[2].__dict__.get("_ver", 0) == f[3]
                and x.dtype in _REAL_DTYPES):
            return self._forward_real(x, f)
        if self._real_ready(x):
            y = self._forward_real(x)
            self._remember_fast()                   # (after the call: building the engine writes self._real, which voids the entry)
            return y
        qz = self.quantizer
        if qz.static and qz.quant and not x.is_cuda and not getattr(self, "simulate_on_cpu", False):
            # There is no CPU fallback.  ``simulate_on_cpu`` is an explicit opt-in used by bench.py's
            # cpu_baseline only: it times the reference's own simulated evaluation on the host cores.
            from mquant_amd._lib import MQuantHipError
            raise MQuantHipError("ActQuantWrapper: static quantized forward needs a CUDA tensor "
                                 "(no CPU fallback)")
        x_dtype = x.dtype
        x = self._rotate(x, x_dtype)
        if self.split:
            x[..., 1:] = self._quantize_input(x[..., 1:], x_dtype)
            lead = self.L1.float()(x[..., 0:1].float())
            rest = self.L2.float()(x[..., 1:].float())
            x = (lead + rest).to(x_dtype)
        else:
            x = self.module(self._quantize_input(x, x_dtype)).to(x_dtype)
        oq = self.out_quantizer
        if oq.bits < 16:
            oq.find_params(x)
            x = oq(x).to(x_dtype)
            oq.free()
        return x


# =============================================================================== sibling fusion
#: leaf names of Linears that their parent module feeds the SAME tensor: the attention projections and
#: the two input projections of a gated MLP (Qwen2 / Llama / SigLIP naming, InternLM2 ``w1``/``w3``, Qwen-VL
#: v1 ``w1``/``w2``).  The names only nominate candidates; what is fused is decided by the checks below.
SIBLING_LEAVES = {"q_proj": "attn", "k_proj": "attn", "v_proj": "attn",
                  "gate_proj": "mlp", "up_proj": "mlp", "w1": "mlp", "w2": "mlp", "w3": "mlp"}


class _GroupResult:
    __slots__ = ("key", "rows", "sel", "y", "consumed")

    def __init__(self, key, rows, sel, y, first):
        self.key, self.rows, self.sel, self.y, self.consumed = key, rows, sel, y, {first}


def _no_pass_hook(*_):
    return None


class _GroupPassHook:
    """Forward-pre-hook on a group's parent module: a new forward pass empties the group's cache.  Holds the group weakly and
    pickles / deep-copies as a no-op (the copy's wrappers have no group: ``_group`` is dropped like ``_real``)."""

    def __init__(self, group):
        import weakref
        self._ref = weakref.ref(group)

    def __call__(self, *_):
        grp = self._ref()
        if grp is not None:
            grp.new_pass()
        return None

    def __reduce__(self):
        return (_no_pass_hook_factory, ())


def _no_pass_hook_factory():
    return _no_pass_hook


class SiblingGroup:
    """Wrappers under one parent that quantize the same input with the same static scale set run as ONE
    quantize + ONE GEMM over their concatenated output channels; each member's ``forward`` returns its
    column slice (a view) of that product.  The reference wraps -- and evaluates -- every Linear on its
    own (quant_utils.py:626-662); the result here is the same bit for bit because every output channel's
    int32 accumulator, weight scale and bias are its own and the activation levels are shared by
    construction (same input, same scale).

    Correctness does not rest on the name lists: a member takes the shared product only when the tensor
    it is handed IS the one the product was computed from (same storage, shape, strides, dtype and
    version counter -- the group keeps that tensor alive meanwhile, so the address cannot be recycled)
    and each member takes a product at most once.  Anything else recomputes; a group whose products
    keep going unused dissolves itself and its members fall back to their own engines."""

    def __init__(self, parent: str, members):
        self.parent = parent
        self.members = list(members)                       # [(leaf name, wrapper)]
        self.index = {id(w): i for i, (_, w) in enumerate(self.members)}
        self.enabled = True
        self.engine = None
        self.offsets = None
        self._result = None
        self._unused = 0
        self.launches = 0                                  # fused GEMMs issued (tests, bench accounting)
        #: forward passes of the parent module seen so far (a forward-pre-hook on the parent, ``group_siblings``): a cached
        #: product never survives into the next pass -- version counters miss writes through ``.data``, through this
        #: repository's own in-place kernels and into inference tensors, so a partially consumed product could otherwise
        #: be handed out for a refilled buffer
        self._epoch = 0
        self._hook = None

    def new_pass(self, *_):
        self._epoch += 1
        self._result = None

    def reset(self):
        self.engine, self.offsets, self._result = None, None, None

    def dissolve(self):
        self.enabled = False
        self.reset()
        for _, w in self.members:
            if w.__dict__.get("_group") is self:
                w._group = None
        if self._hook is not None:
            self._hook.remove()
            self._hook = None

    def _build(self, rows) -> bool:
        from mquant_amd.engine import W4A8Linear
        parts = []
        for _, w in self.members:
            if not (w._real_ready(rows) and _fusable(w)):
                return False
            parts.append(w._real_parts(rows.device))
        p0 = parts[0]
        if any(p["s0"] != p0["s0"] or p["s1"] != p0["s1"] or p["bits"] != p0["bits"] or p["dynamic"] is not None
               or p["w_shift"] is not None or p["levels"].shape[1] != p0["levels"].shape[1]
               or (p["w_groups"] is None) != (p0["w_groups"] is None) or p["col_perm"] is not None
               or (p["w_groups"] is not None and p["w_groups"][1] != p0["w_groups"][1]) for p in parts):
            return False
        w_groups = None
        if p0["w_groups"] is not None:                      # --w_groupsize: the members' [groups, channels] scale tables side by side
            w_groups = (torch.cat([p["w_groups"][0] for p in parts], dim=1).contiguous(), p0["w_groups"][1])
        bias = None
        if any(p["bias"] is not None for p in parts):
            bias = torch.cat([p["bias"].float() if p["bias"] is not None
                              else torch.zeros(p["levels"].shape[0], device=rows.device) for p in parts])
        ends, n = [], 0
        for p in parts:
            ends.append((n, n + p["levels"].shape[0]))
            n += p["levels"].shape[0]
        self.engine = W4A8Linear(torch.cat([p["levels"] for p in parts], dim=0),
                                 torch.cat([p["scale"] for p in parts], dim=0), p0["bits"], bias, p0["s0"], p0["s1"],
                                 w_groups=w_groups)
        self.offsets = ends
        return True

    def forward(self, wrapper, rows, sel):
        """The column slice of ``wrapper`` in the shared product for ``rows``, or None (caller runs alone)."""
        i = self.index[id(wrapper)]
        if self._hook is None and rows.is_inference():
            return None            # no version counter and no parent pass boundary to tell a refilled buffer by: run alone
        key = (self._epoch, rows.data_ptr(), utils.tensor_version(rows), tuple(rows.shape), rows.stride(), rows.dtype,
               None if sel is None else (sel.data_ptr(), utils.tensor_version(sel)))
        res = self._result
        if res is not None and res.key == key and i not in res.consumed:
            res.consumed.add(i)
            lo, hi = self.offsets[i]
            y = res.y[:, lo:hi]
            if len(res.consumed) == len(self.members):
                self._result = None                        # everybody served: let the buffers go
            return y
        if res is not None:
            # a product that only its producer used: siblings are being fed different tensors
            self._unused = self._unused + 1 if len(res.consumed) == 1 else 0
            if self._unused >= 3:
                self.dissolve()
                return None
        if self.engine is None and not self._build(rows):
            self.dissolve()
            return None
        y = self.engine.forward(rows, sel)
        self.launches += 1
        self._result = _GroupResult(key, rows, sel, y, i)
        lo, hi = self.offsets[i]
        return y[:, lo:hi]


def real_engine(wrapper):
    """The integer engine ``wrapper``'s quantized forward runs on -- its own, or its sibling group's -- or
    None while it still simulates."""
    grp = wrapper.__dict__.get("_group")
    if grp is not None and grp.enabled and grp.engine is not None:
        return grp.engine
    return wrapper._real


def _fusable(w) -> bool:
    """Static per-tensor int8 activations, a plain nn.Linear, nothing that changes the input on the way
    (pad, online Hadamard, split column) and nothing that claims the rank-1 epilogue term."""
    qz = w.quantizer
    return (isinstance(w.module, torch.nn.Linear) and bool(getattr(qz, "static", False)) and not w.split
            and not w.online_full_had and not w.online_partial_had and w.pad_to is None
            and not getattr(w, "_real_frozen", False) and w.real_quant and len(w._forward_pre_hooks) == 0
            and getattr(getattr(qz, "quantizer", None), "scale", None) is not None
            and qz.quantizer.scale.numel() == 1)


def _scale_signature(w):
    qz = w.quantizer
    s0 = float(qz.quantizer.scale)
    s1 = None
    if qz.msq:
        s1 = float(qz.quantizer_text.scale) if qz.quantizer_text.scale is not None else s0
    wq = w.weight_quantizers.get("module")
    return (w.module.in_features, s0, s1, None if wq is None else int(wq.bits), w.module.weight.dtype,
            w.module.weight.device)


def group_siblings(model, args=None) -> list:
    """Form ``SiblingGroup``s under ``model`` (called by ``model_quant``): wrappers with the same parent,
    a leaf name of the same ``SIBLING_LEAVES`` family, the same in_features and IDENTICAL calibrated
    activation scale sets.  Module names, ``module`` / ``L1`` / ``L2`` sub-modules and pickles are
    untouched; ``args.no_sibling_fusion`` switches it off.  Returns the groups formed."""
    wrappers = find_qlayers(model, layers=[ActQuantWrapper])
    for w in wrappers.values():
        g = w.__dict__.get("_group")
        if g is not None:
            g.dissolve()
    if args is not None and getattr(args, "no_sibling_fusion", False):
        return []
    skip = list(getattr(args, "skip_names", [])) if args is not None else []
    cand = {}
    for name, w in wrappers.items():
        parent, _, leaf = name.rpartition(".")
        fam = SIBLING_LEAVES.get(leaf)
        if fam is None or not parent or any(p in name for p in skip) or not _fusable(w) or not w.quantizer.quant:
            continue
        cand.setdefault((parent, fam) + _scale_signature(w), []).append((leaf, w))
    groups = []
    modules = dict(model.named_modules())
    for key, members in cand.items():
        if len(members) < 2:
            continue
        grp = SiblingGroup(key[0], members)
        for _, w in members:
            w._group = grp
        parent_mod = modules.get(key[0])
        if parent_mod is not None:       # every forward pass of the parent starts with an empty cache
            grp._hook = parent_mod.register_forward_pre_hook(_GroupPassHook(grp))
        groups.append(grp)
    return groups


class ActRotateWrapper(torch.nn.Module):
    """Rotates two inputs by a dense Q before calling ``module(x, y)`` (no upstream caller)."""

    def __init__(self, module: torch.nn.Module, QMatrix):
        super().__init__()
        self.module = module
        self.register_buffer("q_matrix", QMatrix)
        self.fp32_had = False

    def forward(self, x, y):
        x_dtype = x.dtype
        if self.fp32_had:
            x = (x.float() @ self.q_matrix).to(x_dtype)
            y.copy_((y.float() @ self.q_matrix).to(y.dtype))
        else:
            x = x @ self.q_matrix
            y.copy_(y @ self.q_matrix.to(y.dtype))
        return self.module(x, y).to(x_dtype)


# =============================================================================== WeightQuantizer
class WeightQuantizer(torch.nn.Module):
    """Round-to-nearest weight quantizer (GPTQ lineage): per output channel (``perchannel``)
    or per tensor, symmetric or affine, optional MSE clip search over shrink factors
    p = 1 - i/grid, i < maxshrink*grid, error = sum |q - x|^norm."""

    def __init__(self, shape=1):
        super().__init__()
        self.register_buffer("maxq", torch.tensor(0))
        self.register_buffer("scale", torch.zeros(shape))
        self.register_buffer("zero", torch.zeros(shape))

    def configure(self, bits, perchannel=False, sym=True, mse=False, norm=2.4, grid=100,
                  maxshrink=0.8):
        self.bits = bits
        self.perchannel = perchannel
        self.sym = sym
        self.mse = mse
        self.norm = norm
        self.grid = grid
        self.maxshrink = maxshrink
        self.maxq = torch.tensor(2 ** (bits - 1) - 1 if sym else 2 ** bits - 1)

    def _grid_params(self, lo, hi):
        if self.sym:
            scale = hi / self.maxq
            return scale, torch.zeros_like(scale)
        scale = (hi - lo) / self.maxq
        return scale, torch.round(-lo / scale)

    def _fake(self, x, scale, zero):
        if self.sym:
            return sym_quant_dequant(x, scale, self.maxq)
        return asym_quant_dequant(x, scale, zero, self.maxq)

    #: CUDA tensors, symmetric + per-channel: one launch of ``mq_wquant_sym`` (same arithmetic,
    #: errors of the clip search summed in ascending k) instead of ~10 torch kernels per candidate
    use_kernel = True

    def find_params(self, x):
        if self.bits == 16:
            return
        dev = x.device
        self.maxq = self.maxq.to(dev)
        shape = x.shape
        if self.use_kernel and x.is_cuda and self.sym and self.perchannel and 2 <= self.bits <= 8:
            from mquant_amd import ops
            scale = ops.wquant_sym(x.flatten(1), self.bits, self.mse, self.norm, self.grid, self.maxshrink,
                                   want_levels=False)[0]
            self.scale = scale.reshape([-1] + [1] * (len(shape) - 1))
            self.zero = torch.zeros_like(self.scale)
            return
        rows = x.flatten(1) if self.perchannel else x.flatten().unsqueeze(0)
        z = torch.zeros(rows.shape[0], device=dev)
        xmin = torch.minimum(rows.min(1)[0], z)
        xmax = torch.maximum(rows.max(1)[0], z)
        if self.sym:
            xmax = torch.maximum(torch.abs(xmin), xmax).clamp(min=1e-5)
            self.scale = xmax / self.maxq
            self.zero = torch.zeros_like(self.scale)
        else:
            dead = (xmin == 0) & (xmax == 0)
            xmin[dead] = -1
            xmax[dead] = +1
            self.scale = (xmax - xmin).clamp(min=1e-5) / self.maxq
            self.zero = torch.round(-xmin / self.scale)
        if self.mse:
            best = torch.full([rows.shape[0]], float("inf"), device=dev)
            for i in range(int(self.maxshrink * self.grid)):
                p = 1 - i / self.grid
                scale1, zero1 = self._grid_params(p * xmin, p * xmax)
                err = self._fake(rows, scale1.unsqueeze(1), zero1.unsqueeze(1))
                err -= rows
                err.abs_()
                err.pow_(self.norm)
                err = torch.sum(err, 1)
                better = err < best
                if torch.any(better):
                    best[better] = err[better]
                    self.scale[better] = scale1[better]
                    self.zero[better] = zero1[better]
        if not self.perchannel:
            self.scale = self.scale.repeat(shape[0])
            self.zero = self.zero.repeat(shape[0])
        bshape = [-1] + [1] * (len(shape) - 1)
        self.scale = self.scale.reshape(bshape)
        self.zero = self.zero.reshape(bshape)

    def quantize(self, x):
        if self.ready() and self.bits < 16:
            return self._fake(x, self.scale, self.zero).to(x.dtype)
        return x

    def enabled(self):
        return self.maxq > 0

    def ready(self):
        return torch.all(self.scale != 0)


def attach_weight_quantizer(wrapper: ActQuantWrapper, submodule: str, quantizer) -> None:
    """Tell a wrapper which ``WeightQuantizer`` produced the (fake-quantized) weight of its
    sub-module ``"module"`` / ``"L2"``; this is what lets ``forward`` recover the integer
    levels and switch to the real W4A8 kernels.  The RTN / GPTQ passes call it."""
    wrapper.weight_quantizers[submodule] = quantizer
    wrapper.invalidate_real()


def attach_weight_quantizers(model, quantizers: dict, prefix: str = "") -> int:
    """Map a ``{dotted_name: WeightQuantizer}`` dict (the return value of the
    ``*_rtn_gptq_fwrd_plus`` passes) onto the wrappers under ``model``; returns the count."""
    wrappers = find_qlayers(model, layers=[ActQuantWrapper], name=prefix)
    n = 0
    for name, qz in quantizers.items():
        for sub in ("module", "L2"):
            tail = "." + sub
            owner = name[: -len(tail)] if name.endswith(tail) else None
            if owner is not None and owner in wrappers:
                attach_weight_quantizer(wrappers[owner], sub, qz)
                n += 1
    return n


# =============================================================================== module surgery
@torch.no_grad()
def fuse_internvl(model):
    """Fold InternViT's LayerScale vectors (ls1, ls2) into attn.proj / mlp.fc2."""
    print("fuse internvl vision model...")
    for layer in model.model.vision_model.encoder.layers:
        for lin, ls in ((layer.attn.proj, layer.ls1), (layer.mlp.fc2, layer.ls2)):
            lin.weight.data *= ls.data.view(-1, 1)
            if getattr(lin, "bias", None) is not None:
                lin.bias.data *= ls.data
            ls[:] = 1


def add_actquant(module, act_per_tensor=False, name="", layers=[torch.nn.Linear]):
    """Recursively wrap every attribute / Sequential entry / ModuleList entry whose EXACT type
    is in ``layers`` (sub-classes are deliberately not matched).  Idempotent."""
    if isinstance(module, ActQuantWrapper):
        return

    def wrap(child):
        return ActQuantWrapper(child, act_per_tensor) if type(child) in layers else child

    for attr in dir(module):
        tmp = getattr(module, attr)
        if type(tmp) in layers:
            setattr(module, attr, ActQuantWrapper(tmp, act_per_tensor))
        elif type(tmp) == torch.nn.Sequential:
            setattr(module, attr, torch.nn.Sequential(
                OrderedDict((n, wrap(c)) for n, c in tmp.named_children())))
        elif type(tmp) == torch.nn.ModuleList:
            setattr(module, attr, torch.nn.ModuleList([wrap(c) for c in tmp.children()]))
    for child_name, child in module.named_children():
        add_actquant(child, act_per_tensor, f"{name}.{child_name}" if name else child_name,
                     [torch.nn.Linear])


def add_actquant_for_mlp1(module, act_per_tensor=False, name="", layers=[torch.nn.Linear]):
    for i in (1, 3):
        module.mlp1[i] = ActQuantWrapper(module.mlp1[i], act_per_tensor)


def internvl_add_act_qaunt(model, args):
    if args.quant_llm:
        add_actquant(model.model.language_model.model, args.act_per_tensor)
    if args.quant_visual_clip:
        emb = model.model.vision_model.embeddings
        emb.patch_embedding = ActQuantWrapper(emb.patch_embedding, args.act_per_tensor)
        add_actquant(model.model.vision_model.encoder, args.act_per_tensor)
    if args.quant_cross_attention:
        add_actquant_for_mlp1(model.model, args.act_per_tensor)


def qwen2vl_add_act_qaunt(model, args):
    if args.quant_llm:
        add_actquant(model.model.model, args.act_per_tensor)
    if args.quant_visual_clip:
        pe = model.model.visual.patch_embed
        pe.proj = ActQuantWrapper(pe.proj, args.act_per_tensor)
        add_actquant(model.model.visual.blocks, args.act_per_tensor)
    if args.quant_cross_attention:
        add_actquant(model.model.visual.merger, args.act_per_tensor)


def qwenvl_add_act_qaunt(model, args):
    if args.quant_llm:
        add_actquant(model.transformer.h, args.act_per_tensor)
    vis = model.transformer.visual
    if args.quant_visual_clip:
        vis.conv1 = ActQuantWrapper(vis.conv1, args.act_per_tensor)
        add_actquant(vis.transformer, args.act_per_tensor)
    if args.quant_cross_attention:
        add_actquant(vis.attn_pool, args.act_per_tensor)
        vis.proj_fc = ActQuantWrapper(vis.proj_fc, args.act_per_tensor)


def minicpmv_add_act_qaunt(model, args):
    if args.quant_llm:
        add_actquant(model.llm.model.layers, args.act_per_tensor)
    if args.quant_visual_clip:
        emb = model.vpm.embeddings
        emb.patch_embedding = ActQuantWrapper(emb.patch_embedding, args.act_per_tensor)
        add_actquant(model.vpm.encoder, args.act_per_tensor)
    if args.quant_cross_attention:
        add_actquant(model.resampler, args.act_per_tensor)


def find_qlayers(module, layers=[torch.nn.Linear, ActQuantWrapper], name=""):
    """{dotted name: module} for every descendant whose exact type is in ``layers``
    (matched modules are not descended into)."""
    if type(module) in layers:
        return {name: module}
    found = {}
    for child_name, child in module.named_children():
        found.update(find_qlayers(child, layers=layers,
                                  name=f"{name}.{child_name}" if name != "" else child_name))
    return found


def _set_flag(model, args, flag, value):
    for name, wrapper in find_qlayers(model, layers=[ActQuantWrapper]).items():
        if any(p in name for p in args.skip_names):
            continue
        setattr(wrapper.quantizer, flag, value)
        wrapper.invalidate_real()
    return model


def model_open_calibrate(model, args):
    return _set_flag(model, args, "calibrate", True)


def model_open_last_calibrate(model, args):
    return _set_flag(model, args, "last_calibrate", True)


def model_close_calibrate(model, args):
    return _set_flag(model, args, "calibrate", False)


def model_quant(model, args):
    _set_flag(model, args, "quant", True)
    group_siblings(model, args)          # q/k/v, gate/up ...: one quantize + one GEMM per shared input
    return model


def model_no_quant(model, args):
    return _set_flag(model, args, "quant", False)


# =============================================================================== calibration
def _calibrate_vlmeval(model, args, dataset, calib_num, kwargs_attr, restore):
    """open -> one ``generate`` per sampled record (20 new tokens; the last one with
    ``last_calibrate`` and a single new token) -> close -> quant."""
    from tqdm import tqdm
    total = len(dataset.data)
    step = math.ceil(total / calib_num)
    print("Calibrating...")
    model_open_calibrate(model.model, args)
    kwargs = getattr(model, kwargs_attr)
    kwargs["max_new_tokens"] = 20
    for i in tqdm(range(0, total, step)):
        if i + step >= total:
            print("last calibrate")
            model_open_last_calibrate(model.model, args)
            kwargs["max_new_tokens"] = 1
        record = dataset.data.iloc[i]
        if hasattr(model, "use_custom_prompt") and model.use_custom_prompt(args.dataset_name):
            struct = model.build_prompt(record, dataset=args.dataset_name)
        else:
            struct = dataset.build_prompt(record)
        model.generate(message=struct, dataset=args.dataset_name)
    restore(model)
    model_close_calibrate(model.model, args)
    print("Calibrate End...")
    model_quant(model.model, args)


def calib_vqa_plus(model, args, dataset, calib_num):
    def restore(m):
        m.kwargs = {}
    _calibrate_vlmeval(model, args, dataset, calib_num, "kwargs", restore)


def calib_qwen2vl_plus(model, args, dataset, calib_num):
    saved = model.generate_kwargs["max_new_tokens"]

    def restore(m):
        m.generate_kwargs["max_new_tokens"] = saved
    _calibrate_vlmeval(model, args, dataset, calib_num, "generate_kwargs", restore)


# ---- Qwen-VL (v1) calibration over the jsonl VQA sets (reference quant_utils.py:722-959) ---------
def _vqa_entry(folder, train, test, metric, max_new_tokens, question=None, annotation=None):
    entry = {"train": f"data/{folder}/{train}", "test": f"data/{folder}/{test}"}
    if question:
        entry["question"] = f"data/{folder}/{question}"
    if annotation:
        entry["annotation"] = f"data/{folder}/{annotation}"
    entry.update(metric=metric, max_new_tokens=max_new_tokens)
    return entry


#: dataset name -> jsonl files (relative to the working directory), metric and generation length
ds_collections = {
    "vqav2_val": _vqa_entry("vqav2", "vqav2_train.jsonl", "vqav2_val.jsonl", "vqa_score", 10,
                            "v2_OpenEnded_mscoco_val2014_questions.json", "v2_mscoco_val2014_annotations.json"),
    "vqav2_testdev": _vqa_entry("vqav2", "vqav2_train.jsonl", "vqav2_testdev.jsonl", None, 10),
    "okvqa_val": _vqa_entry("okvqa", "okvqa_train.jsonl", "okvqa_val.jsonl", "vqa_score", 10,
                            "OpenEnded_mscoco_val2014_questions.json", "mscoco_val2014_annotations.json"),
    "textvqa_val": _vqa_entry("textvqa", "textvqa_train.jsonl", "textvqa_val.jsonl", "vqa_score", 10,
                              "textvqa_val_questions.json", "textvqa_val_annotations.json"),
    "vizwiz_val": _vqa_entry("vizwiz", "vizwiz_train.jsonl", "vizwiz_val.jsonl", "vqa_score", 10,
                             "vizwiz_val_questions.json", "vizwiz_val_annotations.json"),
    "vizwiz_test": _vqa_entry("vizwiz", "vizwiz_train.jsonl", "vizwiz_test.jsonl", None, 10),
    "docvqa_val": _vqa_entry("docvqa", "train.jsonl", "val.jsonl", "anls", 100, annotation="val/val_v1.0.json"),
    "docvqa_test": _vqa_entry("docvqa", "train.jsonl", "test.jsonl", None, 100),
    "chartqa_test_human": _vqa_entry("chartqa", "train_human.jsonl", "test_human.jsonl", "relaxed_accuracy", 100),
    "chartqa_test_augmented": _vqa_entry("chartqa", "train_augmented.jsonl", "test_augmented.jsonl",
                                         "relaxed_accuracy", 100),
    "gqa_testdev": _vqa_entry("gqa", "train.jsonl", "testdev_balanced.jsonl", "accuracy", 10),
    "ocrvqa_val": _vqa_entry("ocrvqa", "ocrvqa_train.jsonl", "ocrvqa_val.jsonl", "accuracy", 100),
    "ocrvqa_test": _vqa_entry("ocrvqa", "ocrvqa_train.jsonl", "ocrvqa_test.jsonl", "accuracy", 100),
    "ai2diagram_test": _vqa_entry("ai2diagram", "train.jsonl", "test.jsonl", "accuracy", 10),
}


class VQADataset(torch.utils.data.Dataset):
    """One json record per line: image, question, question_id[, answer]."""

    def __init__(self, train, test, prompt, few_shot, use_train=False):
        with open(train if use_train else test) as fh:
            self.test = fh.readlines()
        self.prompt, self.few_shot = prompt, few_shot
        if few_shot > 0:
            with open(train) as fh:
                self.train = fh.readlines()

    def __len__(self):
        return len(self.test)

    def __getitem__(self, idx):
        import json
        import random
        rec = json.loads(self.test[idx].strip())
        shots = ""
        if self.few_shot > 0:
            for line in random.sample(self.train, self.few_shot):
                ex = json.loads(line.strip())
                shots += self.prompt.format(ex["image"], ex["question"]) + f" {ex['answer']}"
        return {"question": shots + self.prompt.format(rec["image"], rec["question"]),
                "question_id": rec["question_id"], "annotation": rec.get("answer", None)}


def collate_fn(batches, tokenizer):
    enc = tokenizer([b["question"] for b in batches], return_tensors="pt", padding="longest")
    return ([b["question_id"] for b in batches], enc.input_ids, enc.attention_mask,
            [b["annotation"] for b in batches])


def calib_vqa(model, tokenizers, args, dataset_name, batch_size, num_workers, seed=0, few_shot=0):
    """Qwen-VL driver: ``calib_mode`` "v1" = the first ``calib_num`` batches, "v2" = every
    ``step``-th batch over the whole training split; the last sampled batch runs with
    ``last_calibrate`` and a single new token.  Then close -> quant."""
    from copy import deepcopy
    from tqdm import tqdm
    tokenizer = deepcopy(tokenizers)
    tokenizer.padding_side = "left"
    tokenizer.pad_token_id = tokenizer.eod_id
    info = ds_collections[dataset_name]
    dataset = VQADataset(train=info["train"], test=info["test"], prompt="<img>{}</img>{} Answer:",
                         few_shot=few_shot, use_train=True)
    loader = torch.utils.data.DataLoader(dataset=dataset, batch_size=batch_size, num_workers=num_workers,
                                         pin_memory=torch.cuda.is_available(), drop_last=False,
                                         collate_fn=functools.partial(collate_fn, tokenizer=tokenizer))
    n_batches = math.ceil(len(dataset) / batch_size)
    step = n_batches // args.calib_num

    def run(input_ids, attention_mask, max_new_tokens):
        model.generate(input_ids=input_ids.to(utils.DEV), attention_mask=attention_mask.to(utils.DEV),
                       do_sample=False, num_beams=1, max_new_tokens=max_new_tokens, min_new_tokens=1,
                       length_penalty=1, num_return_sequences=1, output_hidden_states=True, use_cache=True,
                       pad_token_id=tokenizer.eod_id, eos_token_id=tokenizer.eod_id)

    print("Calibrating...")
    model_open_calibrate(model, args)
    idx = 0
    for _, input_ids, attention_mask, _ in tqdm(loader):
        if args.calib_mode == "v1":
            idx += 1
            if idx > args.calib_num:
                break
            last = idx == args.calib_num
        elif args.calib_mode == "v2":
            take, last = idx % step == 0, idx + step > n_batches
            idx += 1
            if not take:
                continue
        else:
            raise ValueError("Invalid calibration mode")
        if last:
            model_open_last_calibrate(model, args)
        run(input_ids, attention_mask, 1 if last else info["max_new_tokens"])
    model_close_calibrate(model, args)
    print("Calibrate End...")
    model_quant(model, args)


def calib_layer(wrapper_or_model, batches, args=None):
    """Protocol helper for stand-alone layers / toy models: open -> forward every batch (the
    last one with ``last_calibrate``) -> close -> quant.  ``batches`` are positional inputs."""
    class _A:
        skip_names = []
    args = args or _A()
    model_open_calibrate(wrapper_or_model, args)
    for i, b in enumerate(batches):
        if i == len(batches) - 1:
            model_open_last_calibrate(wrapper_or_model, args)
        wrapper_or_model(b)
    model_close_calibrate(wrapper_or_model, args)
    model_quant(wrapper_or_model, args)
    return wrapper_or_model
