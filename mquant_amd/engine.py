"""Real-integer execution of one wrapped Linear: the object ``ActQuantWrapper`` freezes into.

``W4A8Linear`` owns the pre-tiled int4/int8 weight image, the per-channel weight scales, the
static activation scale set(s), the optional online-Hadamard descriptor and the optional
rank-1 ``split`` term; ``forward`` is two kernel launches (quantize or Hadamard+quantize,
then the MFMA GEMM with fused dequant).  Everything here is stream-ordered and allocation
free after the first call for a given row count.

Reference semantics: ``ActQuantWrapper.forward``, fake_quant/quant_utils.py:330-391.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch

from . import ops


@dataclass
class HadamardSpec:
    n: int                      # padded size the transform runs on
    K: int                      # special factor (1 for a pure power of two)
    bits: Optional[torch.Tensor]  # K*K packed sign bits on the device (None when K == 1)
    fp32_had: bool = False
    #: NON-DEFAULT: this layer's online rotation may take the fast K x K stage (MQ_HAD_FAST, include/mquant_hip.h): same exact
    #: products, another fp32 accumulation order -- not bit-identical to the reference's CPU run.  A field of the layer's own
    #: descriptor, set by whoever builds the engine (bench.py --had-fast, FullPrefill(had_fast=True)); nothing process-wide.
    fast: bool = False


#: layout of the int8 activations between the quantizer and the GEMM: "tiled" (MQ_LD_TILED: one
#: contiguous KiB per MFMA fragment, what the wave-specialised GEMM kernels stream) or "rows"
#: (row-major, the round-1 layout; kept for A/B measurements).
ACT_LAYOUT = os.environ.get("MQ_ACT_LAYOUT", "tiled")
#: MQ_DEBUG_WORKSPACE=1: every activation image carries a hand-out counter and gemm() refuses a stale handle (Workspace CONTRACT)
DEBUG_WORKSPACE = os.environ.get("MQ_DEBUG_WORKSPACE", "0") == "1"


class Workspace:
    """Scratch for the int8 activations between a quantizer launch and its GEMM: ONE grow-only buffer per
    (device, K_pad, layout), sized for the largest row count seen and sliced per call -- an evaluation over a dataset feeds
    hundreds of distinct prompt lengths (and every decode step) through the same wrappers, and one buffer per distinct
    (M, K_pad) kept forever was ~30 KB x M of dead memory each (VERDICT r4 "What's weak" 9).  A prefix of the tiled image
    is the tiled image of fewer rows (16-row tiles are the slowest index), so a slice needs no copy.

    Everything is stream-ordered: a buffer is written by a quantizer and read by the GEMM behind it on the same stream.
    hipGraph users are pinned: a buffer handed out during stream capture is never released when the workspace grows
    (the graph replays into it), it just stops being handed out.

    CONTRACT: at most ONE outstanding activation image per (device, K_pad, layout).  Two layers of the same input width (or two
    row counts) share the buffer, so a quantizer's output must be consumed by its GEMM before the next quantizer of that width
    is enqueued -- which is what every call site does (quantize -> gemm back to back on one stream).  A handle kept across
    another quantize() of the same width is stale; ``TiledAct.generation`` lets ``gemm`` catch that under MQ_DEBUG_WORKSPACE=1."""

    def __init__(self):
        self._a: Dict[Tuple[int, int, str], torch.Tensor] = {}
        self._x0: Dict[int, torch.Tensor] = {}
        self._captured = set()          # ids of buffers handed out while a stream was capturing
        self._pinned = []               # outgrown buffers a graph may still replay into
        self._gen: Dict[Tuple[int, int, str], int] = {}     # hand-outs per key (debug: stale-handle check)
        self._views: Dict[Tuple[int, int, int], tuple] = {}  # (device, K_pad, M) -> (TiledAct over the live buffer, id(buffer))

    @staticmethod
    def _capturing() -> bool:
        return torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()

    def _grow(self, table, key, numel: int, dtype, device) -> torch.Tensor:
        buf = table.get(key)
        if buf is None or buf.numel() < numel:
            if buf is not None and id(buf) in self._captured:
                self._pinned.append(buf)
            buf = torch.empty((numel,), dtype=dtype, device=device)
            table[key] = buf
        if self._capturing():
            self._captured.add(id(buf))
        return buf

    def act(self, device, M: int, K_pad: int):
        idx = device.index or 0
        if ACT_LAYOUT == "tiled" and not DEBUG_WORKSPACE:
            # the view of (buffer, M) is built once: slicing + view + TiledAct cost ~3 us of host time per call otherwise
            hit = self._views.get((idx, K_pad, M))
            if hit is not None:
                if hit[1] not in self._captured and self._capturing():
                    self._captured.add(hit[1])
                return hit[0]
        rows = ops.ceil_to(max(M, 1), 16) if ACT_LAYOUT == "tiled" else M
        key = (idx, K_pad, ACT_LAYOUT)
        before = self._a.get(key)
        buf = self._grow(self._a, key, rows * K_pad, torch.int8, device)
        if buf is not before:
            self._views = {k: v for k, v in self._views.items() if k[:2] != (idx, K_pad)}
        if ACT_LAYOUT == "tiled":
            t = ops.TiledAct(buf[: rows * K_pad].view(rows // 16, K_pad // 64, 64, 16), M, K_pad)
            if DEBUG_WORKSPACE:
                self._gen[key] = self._gen.get(key, 0) + 1
                t.generation = (key, self._gen[key])
            else:
                if len(self._views) > 4096:
                    self._views.clear()
                self._views[(idx, K_pad, M)] = (t, id(buf))
            return t
        return buf[: M * K_pad].view(M, K_pad)

    def check_fresh(self, a) -> None:
        """MQ_DEBUG_WORKSPACE=1: the image a GEMM is about to read is the LATEST hand-out of its buffer (see CONTRACT)."""
        gen = getattr(a, "generation", None)
        if gen is not None and self._gen.get(gen[0]) != gen[1]:
            raise ops._lib.MQuantHipError(f"stale activation image: buffer {gen[0]} was handed out again (hand-out {self._gen.get(gen[0])}) "
                                     f"after this handle ({gen[1]}) -- one outstanding image per (device, K_pad)")

    def x0(self, device, M: int) -> torch.Tensor:
        return self._grow(self._x0, device.index or 0, M, torch.float32, device)[:M]

    def nbytes(self) -> int:
        """Bytes held (live buffers + outgrown ones pinned by a graph)."""
        bufs = list(self._a.values()) + list(self._x0.values()) + self._pinned
        return sum(b.numel() * b.element_size() for b in bufs)

    def clear(self) -> None:
        """Drop every buffer (only when no captured graph will replay into them)."""
        self._a.clear()
        self._x0.clear()
        self._captured.clear()
        self._pinned.clear()
        self._views.clear()


WORKSPACE = Workspace()


class W4A8Linear:
    """y = dequant( quant(Had(x)) @ W_int^T ) (+ bias) (+ x[:,0] * w0)."""

    def __init__(self, levels: torch.Tensor, s_w: torch.Tensor, w_bits: int,
                 bias: Optional[torch.Tensor], s_x0: float, s_x1: Optional[float] = None,
                 had: Optional[HadamardSpec] = None, w0: Optional[torch.Tensor] = None,
                 in_features: Optional[int] = None, dynamic: Optional[dict] = None,
                 w_shift: Optional[torch.Tensor] = None, split_slice: bool = False,
                 w_groups: Optional[Tuple[torch.Tensor, int]] = None, col_perm: Optional[torch.Tensor] = None):
        assert levels.is_cuda and levels.dtype == torch.int8 and levels.dim() == 2
        self.N, self.K = levels.shape
        self.K_pad = ops.ceil_to(self.K, 128)
        self.w_bits = w_bits
        self.split = w0 is not None
        #: the split column with ASYMMETRIC dynamic activations: ``levels`` holds the quantized columns 1.. only (no zero
        #: column), the quantizer runs on the view x[:, 1:] and column 0 is gathered in floating point
        self.split_slice = bool(split_slice)
        assert not self.split_slice or (self.split and dynamic is not None and not dynamic.get("sym", True))
        self.w_img = ops.prepack(levels, w_bits, zero_col0=self.split and not self.split_slice)
        self.s_w = s_w.reshape(-1).to(torch.float32).contiguous()
        self.bias = None if bias is None else bias.reshape(-1).to(torch.float32).contiguous()
        self.w0 = None if w0 is None else w0.reshape(-1).to(torch.float32).contiguous()
        #: asymmetric weights (--w_asym): levels are stored minus 2^(bits-1) and w_shift[n] = s_w[n] (2^(bits-1) -
        #: z_w[n]); the zero points come back as the rank-1 term (s_x sum_k a[m][k]) * w_shift[n].  The epilogue has TWO
        #: rank-1 slots (mq_gemm_w4a8_rank2_ws): any two of {split column, asymmetric weights, asymmetric activations}
        self.w_shift = None if w_shift is None else w_shift.reshape(-1).to(torch.float32).contiguous()
        #: group-wise WEIGHT scales (--w_groupsize; reference gptq/gptq_utils.py:263-273): (fp32 [K / g, N], g).  The GEMM is
        #: mq_gemm_w4a8_wgroupscale: exact int32 inside a group, the group's scale, fp32 across groups; s_w is unused.
        self.w_groups = None
        if w_groups is not None:
            tbl, g = w_groups
            assert tbl.dtype == torch.float32 and tuple(tbl.shape) == (self.K // g, self.N) and self.K % g == 0
            assert w0 is None and w_shift is None, "weight groups: no split column, symmetric levels"
            self.w_groups = (tbl.contiguous(), int(g))
        #: --act_order with --w_groupsize (reference gptq/gptq_utils.py:228-233, 263-273): the weight image keeps the solver's column
        #: order (column j = input channel col_perm[j]; its groups are runs of those columns), so the activation columns are gathered
        #: the same way -- after the pad and the online Hadamard, in front of the quantizer (a gather of the fp tensor: elementwise
        #: quantizers and per-row scales commute with a column permutation).
        self.col_perm = None
        if col_perm is not None:
            assert col_perm.dtype == torch.long and col_perm.numel() == self.K and w0 is None and not split_slice
            self.col_perm = col_perm.contiguous()
        self.s_x0 = float(s_x0)
        self.s_x1 = None if s_x1 is None else float(s_x1)
        self.had = had
        #: None = static scales; dict(bits=, clip_ratio=, sym=) = dynamic per-token quantizer (symmetric, or
        #: asymmetric: the zero point and the storage offset come back through the rank-1 epilogue term
        #: shift[m] * (s_w[n] * sum_k q_w[n][k]), which is the slot the split column uses -- not both)
        self.dynamic = dynamic
        self.w_colsum = None
        self.wsum_groups = None
        gsz = int(dynamic.get("groupsize", -1) or -1) if dynamic is not None else -1
        if gsz > 0 and not dynamic.get("sym", True):
            # asymmetric group-wise activations: fp32 [K / g, N] sums of the weight levels per group
            assert not self.split and self.w_shift is None and self.K % gsz == 0
            self.wsum_groups = levels.to(torch.int32).reshape(self.N, self.K // gsz, gsz).sum(dim=2).t().to(torch.float32).contiguous()
        elif dynamic is not None and not dynamic.get("sym", True):
            # s_x (2^(b-1) - z_x)[m] multiplies s_w sum_k W~-levels[n][k]; with asymmetric weights the level of column k is
            # p + (2^(bw-1) - z_w): the constant part adds Kq * w_shift[n] (Kq = quantized columns)
            self.w_colsum = (levels.to(torch.int32).sum(dim=1).to(torch.float32) * self.s_w).contiguous()
            if self.w_shift is not None:
                kq = levels.shape[1] - (1 if self.split and not self.split_slice else 0)
                self.w_colsum = (self.w_colsum + float(kq) * self.w_shift).contiguous()
        #: rank-1 terms of the layer: two ride in the GEMM epilogue, a third (all of split column, asymmetric weights and
        #: asymmetric activations at once) is added by mq_rank1_add_cast behind an fp32 GEMM output
        self.n_terms = int(self.split) + int(self.w_shift is not None) + int(self.w_colsum is not None)
        self.in_features = self.K if in_features is None else in_features
        if had is not None:
            assert had.n == self.K + (1 if self.split_slice else 0), "Hadamard size must equal the (padded) reduction dim"

    @classmethod
    def from_float(cls, w: torch.Tensor, w_bits: int, s_x0: float, s_x1: Optional[float] = None,
                   bias: Optional[torch.Tensor] = None, mse: bool = False, had: Optional[HadamardSpec] = None,
                   split: bool = False, in_features: Optional[int] = None) -> "W4A8Linear":
        """Quantize floating-point weights on the device (``mq_wquant_sym``: RTN or MSE clip search,
        reference quant_utils.py:446-518) and freeze.  ``split`` keeps column 0 in fp32 (``w0``)."""
        w2 = w.reshape(w.shape[0], -1)
        w0 = w2[:, 0].float() if split else None
        src = w2[:, 1:] if split else w2          # reference: L2 holds columns 1.., quantized on its own
        scale, levels, _, _ = ops.wquant_sym(src, w_bits, mse)
        if split:
            levels = torch.cat([torch.zeros_like(levels[:, :1]), levels], dim=1)
        return cls(levels, scale, w_bits, bias, s_x0, s_x1, had=had, w0=w0, in_features=in_features)

    # -- the two launches, exposed separately so callers can share one quantization ----
    def quantize(self, x2: torch.Tensor, row_sel: Optional[torch.Tensor] = None):
        M = x2.shape[0]
        a = WORKSPACE.act(x2.device, M, self.K_pad)
        x0 = WORKSPACE.x0(x2.device, M) if self.split else None
        if self.col_perm is not None:
            ops.quantize_act_i8(self._gathered(x2), self.s_x0, self.s_x1, row_sel=row_sel, out=a)
            return a, None
        if self.had is not None:
            ops.hadamard_quant_i8(x2, self.had.n, self.had.K, self.had.bits, self.s_x0, self.s_x1,
                                  fp32_had=self.had.fp32_had, row_sel=row_sel,
                                  skip_col0=self.split, out=a, x0_out=x0, fast=self.had.fast)
        else:
            ops.quantize_act_i8(x2, self.s_x0, self.s_x1, row_sel=row_sel, skip_col0=self.split,
                                out=a, x0_out=x0)
        return a, x0

    def _gathered(self, x2: torch.Tensor) -> torch.Tensor:
        """The Linear's input in the weight image's column order (--act_order): pad, online Hadamard (its own launch), gather."""
        if x2.shape[1] < self.K and self.had is None:
            x2 = torch.nn.functional.pad(x2, (0, self.K - x2.shape[1]))
        if self.had is not None:
            x2 = ops.hadamard(x2, self.had.n, self.had.K, self.had.bits, self.had.fp32_had, fast=self.had.fast)
        return x2.index_select(1, self.col_perm)

    def _no_col_perm(self, what: str) -> None:
        """Producer-side entry points hand out / fill the activation image in the ORIGINAL column order; an --act_order layer's
        weight image is in the solver's order (col_perm), so only quantize() / forward_dynamic(), which gather, may feed it."""
        assert self.col_perm is None, (f"{what}: this layer's weight image is in --act_order column order (col_perm); "
                                       "use quantize() + gemm(), which gather the activation columns")

    def act_buffer(self, M: int):
        """The workspace destination a quantizer of this layer's input writes (for producers that quantize themselves).
        Contract (Workspace): at most ONE outstanding activation image per (device, K_pad) -- the buffer is shared by every
        layer of that width and every row count, so the producer's launch and this layer's GEMM must be back to back on the stream."""
        self._no_col_perm("act_buffer")
        return WORKSPACE.act(self.w_img.device, M, self.K_pad)

    def quantize_rmsn(self, x: torch.Tensor, mean_dim: float, eps: float,
                      row_sel: Optional[torch.Tensor] = None):
        """Weight-less RMS norm + quantize in one launch (layers without an online Hadamard)."""
        assert self.had is None and not self.split
        self._no_col_perm("quantize_rmsn")
        a = WORKSPACE.act(x.device, x.shape[0], self.K_pad)
        ops.rmsn_quantize_i8(x, mean_dim, eps, self.s_x0, self.s_x1, row_sel=row_sel, out=a)
        return a, None

    def quantize_act(self, x: torch.Tensor, x2: Optional[torch.Tensor], act: int,
                     row_sel: Optional[torch.Tensor] = None):
        """Activation (silu(x)*x2 / quick_gelu(x)) + Hadamard + quantize in one launch: the input of a
        rotated Linear straight from the producer's output (needs an online Hadamard on this layer)."""
        assert self.had is not None, "the fused activation prologue lives in the Hadamard kernel"
        self._no_col_perm("quantize_act")
        M = x.shape[0]
        a = WORKSPACE.act(x.device, M, self.K_pad)
        x0 = WORKSPACE.x0(x.device, M) if self.split else None
        ops.act_hadamard_quant_i8(x, x2, act, self.had.n, self.had.K, self.had.bits, self.s_x0, self.s_x1,
                                  fp32_had=self.had.fp32_had, row_sel=row_sel, skip_col0=self.split,
                                  out=a, x0_out=x0, fast=self.had.fast)
        return a, x0

    def gemm(self, a: torch.Tensor, x0: Optional[torch.Tensor], out_dtype: torch.dtype,
             row_sel: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None):
        w0 = self.w0
        if DEBUG_WORKSPACE:
            WORKSPACE.check_fresh(a)
        if self.w_groups is not None:
            return ops.gemm_w4a8_wgroupscale(a, self.w_img, self.w_bits, self.N, self.w_groups[0], self.w_groups[1],
                                             s_x0=self.s_x0, s_x1=self.s_x1, row_sel=row_sel, bias=self.bias,
                                             out_dtype=out_dtype, out=out)
        if self.w_shift is not None:
            xs = ops.act_rowsum_scaled(a, self.s_x0, self.s_x1, row_sel)
            if self.split:                   # --w_asym + --visual_split: both rank-1 slots
                return ops.gemm_w4a8_rank2(a, self.w_img, self.w_bits, self.N, self.s_w, x0, self.w0, xs, self.w_shift,
                                           s_x0=self.s_x0, s_x1=self.s_x1, row_sel=row_sel, bias=self.bias,
                                           out_dtype=out_dtype, out=out)
            x0, w0 = xs, self.w_shift
        return ops.gemm_w4a8(a, self.w_img, self.w_bits, self.N, self.s_x0, self.s_w,
                             s_x1=self.s_x1, row_sel=row_sel, bias=self.bias, x0=x0, w0=w0,
                             out_dtype=out_dtype, out=out)

    def act_in_store_ok(self, act: int) -> bool:
        """Whether ``gemm_act`` can run this layer: a plain static layer (no split column, rank-1 terms, groups or --act_order), tiled
        activations, and for silu(gate) * up a fused gate|up image of 2 x (a multiple of 32) channels."""
        plain = (not self.split and self.w_shift is None and self.w_groups is None and self.dynamic is None and self.col_perm is None
                 and ACT_LAYOUT == "tiled")
        if act == ops.ACT_SILU_MUL:
            return plain and self.N % 64 == 0
        return plain and act == ops.ACT_QUICK_GELU and self.N % 8 == 0

    def gemm_act(self, a, act: int, out_dtype: torch.dtype, row_sel: Optional[torch.Tensor] = None,
                 out: Optional[torch.Tensor] = None):
        """The Linear with the activation of its CONSUMER folded into the GEMM's store (``mq_gemm_w4a8_act_ws``): for a fused
        gate|up engine ``silu(gate) * up`` -- ONE [M, N/2] tensor instead of [M, N] -- and ``quick_gelu`` behind the vision tower's
        fc1; the rotated Linear behind it (down_proj / fc2) then runs its plain Hadamard + quantize launch."""
        assert self.act_in_store_ok(act), "gemm_act: plain static layers only (see act_in_store_ok)"
        if DEBUG_WORKSPACE:
            WORKSPACE.check_fresh(a)
        return ops.gemm_w4a8_act(a, self.w_img, self.w_bits, self.N, self.s_x0, self.s_w, act, s_x1=self.s_x1, row_sel=row_sel,
                                 bias=self.bias, out_dtype=out_dtype, out=out)

    def gemm_rope(self, a, cos: torch.Tensor, sin: torch.Tensor, rope_cols: int, out_dtype: torch.dtype,
                  row_sel: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None):
        """The Linear with the rotary embedding of its first ``rope_cols`` output columns (heads of 128: the q | k part of a
        fused q|k|v projection) folded into the GEMM's store; plain static layers only."""
        assert not self.split and self.w_shift is None and self.w_groups is None and self.dynamic is None
        self._no_col_perm("gemm_rope")
        return ops.gemm_w4a8_rope(a, self.w_img, self.w_bits, self.N, self.s_x0, self.s_w, cos, sin, rope_cols,
                                  s_x1=self.s_x1, row_sel=row_sel, bias=self.bias, out_dtype=out_dtype, out=out)

    def gemm_residual(self, a: torch.Tensor, x0: Optional[torch.Tensor], residual: torch.Tensor,
                      row_sel: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None):
        """residual + Linear in one launch (same rounding as torch's `hidden + linear(x)`)."""
        assert self.w_groups is None, "the residual epilogue is not in the weight-group kernel"
        self._no_col_perm("gemm_residual")
        if DEBUG_WORKSPACE:
            WORKSPACE.check_fresh(a)
        w0 = self.w0
        if self.w_shift is not None:
            assert not self.split, "the residual epilogue carries one rank-1 term"
            x0, w0 = ops.act_rowsum_scaled(a, self.s_x0, self.s_x1, row_sel), self.w_shift
        return ops.gemm_w4a8_residual(a, self.w_img, self.w_bits, self.N, self.s_x0, self.s_w, residual,
                                      s_x1=self.s_x1, row_sel=row_sel, bias=self.bias, x0=x0, w0=w0, out=out)

    def forward_dynamic(self, x2: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """[Hadamard ->] dynamic per-token quantize -> GEMM with per-row scales.  The row maximum
        needs the whole rotated row, so the Hadamard runs as its own launch here."""
        if self.col_perm is not None:
            x2 = self._gathered(x2)
        elif self.had is not None:
            x2 = ops.hadamard(x2, self.had.n, self.had.K, self.had.bits, self.had.fp32_had, fast=self.had.fast)
        a = WORKSPACE.act(x2.device, x2.shape[0], self.K_pad)
        g = int(self.dynamic.get("groupsize", -1) or -1)
        if g > 0 and not self.dynamic.get("sym", True):
            # asymmetric group-wise scales (--a_groupsize + --a_asym): the constant part of a group's levels meets the group's
            # weight sum (wsum_groups, built once)
            a, s_groups, _, sh_groups = ops.quantize_act_group_asym_i8(x2, g, self.dynamic["bits"], self.dynamic["clip_ratio"], out=a)
            return ops.gemm_w4a8_groupscale_asym(a, self.w_img, self.w_bits, self.N, s_groups, sh_groups, self.wsum_groups, g,
                                                 self.s_w, bias=self.bias, out_dtype=x2.dtype, out=out)
        if g > 0:
            # group-wise scales (--a_groupsize): exact int32 sums inside a group, fp32 across groups
            a, s_groups = ops.quantize_act_group_i8(x2, g, self.dynamic["bits"], self.dynamic["clip_ratio"], out=a)
            if self.w_groups is not None:        # ... and --w_groupsize of the same size: both scales per group
                assert self.w_groups[1] == g
                return ops.gemm_w4a8_wgroupscale(a, self.w_img, self.w_bits, self.N, self.w_groups[0], g, s_x_groups=s_groups,
                                                 bias=self.bias, out_dtype=x2.dtype, out=out)
            return ops.gemm_w4a8_groupscale(a, self.w_img, self.w_bits, self.N, s_groups, g, self.s_w, bias=self.bias,
                                            out_dtype=x2.dtype, out=out)
        asym = self.w_colsum is not None
        xq, x0 = x2, None
        if self.split_slice:                 # asymmetric activations + split column: quantize the view x[:, 1:]
            xq, x0 = x2[:, 1:], x2[:, 0].float().contiguous()
        if self.dynamic.get("per_tensor", False):
            a, s_rows, _, shift, x0q = ops.quantize_act_tensor_i8(xq, self.dynamic["bits"], self.dynamic["clip_ratio"],
                                                                  asym=asym, skip_col0=self.split and not self.split_slice, out=a)
        elif asym:
            a, s_rows, _, shift = ops.quantize_act_dyn_asym_i8(xq, self.dynamic["bits"], self.dynamic["clip_ratio"], out=a)
            x0q = None
        else:
            a, s_rows, x0q = ops.quantize_act_dyn_i8(xq, self.dynamic["bits"], self.dynamic["clip_ratio"],
                                                     skip_col0=self.split, out=a)
            shift = None
        if x0 is None:
            x0 = x0q
        if self.w_groups is not None:            # dynamic per-token symmetric activations x weight groups
            return ops.gemm_w4a8_wgroupscale(a, self.w_img, self.w_bits, self.N, self.w_groups[0], self.w_groups[1],
                                             s_x_rows=s_rows, bias=self.bias, out_dtype=x2.dtype, out=out)
        # the rank-1 epilogue terms (row factor, channel factor) of this layer, at most two
        terms = []
        if self.split:
            terms.append((x0, self.w0))
        if asym:
            terms.append((shift, self.w_colsum))
        if self.w_shift is not None:
            terms.append((ops.act_rowsum_scaled(a, s_x_rows=s_rows), self.w_shift))
        if len(terms) == 3:
            # split column + asymmetric activations + asymmetric weights: two terms in the epilogue (fp32 result), the third
            # added behind it in fp32 and rounded once -- what a third epilogue slot would compute
            y32 = ops.gemm_w4a8_rank2(a, self.w_img, self.w_bits, self.N, self.s_w, terms[0][0], terms[0][1], terms[1][0],
                                      terms[1][1], s_x_rows=s_rows, bias=self.bias, out_dtype=torch.float32)
            return ops.rank1_add_cast(y32, terms[2][0], terms[2][1], x2.dtype, out=out)
        if len(terms) == 2:
            return ops.gemm_w4a8_rank2(a, self.w_img, self.w_bits, self.N, self.s_w, terms[0][0], terms[0][1], terms[1][0],
                                       terms[1][1], s_x_rows=s_rows, bias=self.bias, out_dtype=x2.dtype, out=out)
        xt, wt = terms[0] if terms else (None, None)
        return ops.gemm_w4a8_rowscale(a, self.w_img, self.w_bits, self.N, s_rows, self.s_w, bias=self.bias,
                                      x0=xt, w0=wt, out_dtype=x2.dtype, out=out)

    # -- eager fast path ---------------------------------------------------------------------
    # The reference's scripts run eager, one ActQuantWrapper.forward per Linear and token step: at M = 768 a Linear has ~38 us of
    # GPU work and the generic ops (decorator, argument checks, layout helpers, keyword marshalling) cost ~25 us of host time per
    # forward (profiles/r5_decode_host_overhead.txt).  A plain static layer (per-channel symmetric weights, static scale set(s),
    # optional online Hadamard / split column) binds its two entry points and their constant arguments ONCE; per call only the
    # input pointer, the row count, the output and the stream are marshalled.  Same two launches, same arguments.
    def _bind_fast(self):
        from . import _lib
        lib = _lib.load()
        plain = (self.dynamic is None and self.col_perm is None and self.w_groups is None and self.w_shift is None
                 and ACT_LAYOUT == "tiled" and not DEBUG_WORKSPACE)
        self._fast = None
        if plain:
            dev = self.w_img.device
            s0 = float(self.s_x0)
            s1 = float(self.s_x0 if self.s_x1 is None else self.s_x1)
            had = self.had
            self._fast = dict(
                dev=dev, idx=dev.index or 0, s0=s0, s1=s1, scales=(self.s_x0, self.s_x1),
                quant=lib.mq_hadamard_quant_i8 if had is not None else lib.mq_quantize_act_i8,
                gemm=lib.mq_gemm_w4a8_ws, err=lib.mq_last_error,
                had=None if had is None else (had.n, had.K, ops._ptr(had.bits), had.fp32_had, had.fast),
                had_obj=had, w=self.w_img.data_ptr(), s_w=self.s_w.data_ptr(), bias=ops._ptr(self.bias), w0=ops._ptr(self.w0),
                skip=int(self.split),
                # the tensors whose addresses are bound: held here (their storage cannot be recycled) and compared by identity
                # on every call, so re-assigning one of the engine's tensors re-binds instead of launching on a stale pointer
                refs=(self.w_img, self.s_w, self.bias, self.w0, had))
        return self._fast

    def _forward_fast(self, f, x2: torch.Tensor, row_sel, out):
        M, K = x2.shape
        dev = f["dev"]
        a = WORKSPACE.act(dev, M, self.K_pad)
        x0 = WORKSPACE.x0(dev, M) if self.split else None
        stream = torch._C._cuda_getCurrentRawStream(f["idx"])
        sel = None if row_sel is None else row_sel.data_ptr()
        x0p = None if x0 is None else x0.data_ptr()
        dt = ops._DT[x2.dtype]
        h = f["had"]
        if h is None:
            rc = f["quant"](x2.data_ptr(), dt, M, K, x2.stride(0), f["s0"], f["s1"], None, None, sel, f["skip"], x0p,
                            a.data.data_ptr(), self.K_pad, 0, stream)
        else:
            flags = (ops.HAD_FP32 if h[3] else 0) | (ops.HAD_PREPARED if f["had_obj"].bits is not None and f["had_obj"].bits.dtype == torch.int64 else 0) \
                | (ops.HAD_FAST if f["had_obj"].fast else 0)
            rc = f["quant"](x2.data_ptr(), dt, M, K, x2.stride(0), h[0], h[1], h[2], flags, f["s0"], f["s1"], sel, f["skip"], x0p,
                            a.data.data_ptr(), self.K_pad, 0, stream)
        if rc:
            raise ops._lib.MQuantHipError(f"quantize failed (status {rc}): {f['err']().decode('utf-8', 'replace')}")
        if out is None:
            out = torch.empty((M, self.N), dtype=x2.dtype, device=dev)
        ws = ops.splitk_workspace(dev)
        rc = f["gemm"](a.data.data_ptr(), 0, f["w"], self.w_bits, M, self.N, self.K_pad, f["s0"], f["s1"], sel, f["s_w"],
                       f["bias"], x0p, f["w0"], out.data_ptr(), ops._DT[out.dtype], out.stride(0), ws.data_ptr(), ws.numel(), stream)
        if rc:
            raise ops._lib.MQuantHipError(f"mq_gemm_w4a8_ws failed (status {rc}): {f['err']().decode('utf-8', 'replace')}")
        return out

    def forward(self, x: torch.Tensor, row_sel: Optional[torch.Tensor] = None,
                out: Optional[torch.Tensor] = None) -> torch.Tensor:
        two_d = x.dim() == 2
        x2 = x if two_d else x.reshape(-1, x.shape[-1])
        if self.dynamic is not None:
            y = self.forward_dynamic(x2, out)
            return y if two_d else y.reshape(*x.shape[:-1], self.N)
        f = self.__dict__.get("_fast", False)
        if f is False:
            f = self._bind_fast()
        if f is not None:
            r = f["refs"]
            if not (r[0] is self.w_img and r[1] is self.s_w and r[2] is self.bias and r[3] is self.w0 and r[4] is self.had):
                f = self._bind_fast()
        if (f is not None and f["scales"] == (self.s_x0, self.s_x1) and x2.is_cuda and x2.stride(1) == 1 and x2.dtype in ops._DT
                and x2.device == f["dev"] and f["idx"] == torch.cuda.current_device()
                and (row_sel is None or row_sel.is_cuda) and (out is None or out.is_cuda)):
            y = self._forward_fast(f, x2, row_sel, out)
        else:
            if f is not None and f["scales"] != (self.s_x0, self.s_x1):
                self.__dict__.pop("_fast", None)        # the scale set was swapped (calibration): bind again next time
            a, x0 = self.quantize(x2, row_sel)
            y = self.gemm(a, x0, x.dtype, row_sel, out)
        return y if two_d else y.reshape(*x.shape[:-1], self.N)

    __call__ = forward

    # -- bookkeeping used by bench.py / DESIGN.md ------------------------------------------
    def gemm_ops(self, M: int) -> int:
        return 2 * M * self.K_pad * self.N

    def gemm_bytes(self, M: int, out_bytes: int = 2) -> int:
        return M * self.K_pad + self.K_pad * self.N * self.w_bits // 8 + out_bytes * M * self.N + 4 * self.N

    def quant_bytes(self, M: int, in_bytes: int = 2) -> int:
        return in_bytes * M * self.in_features + M * self.K_pad
