"""Min/max observer -- the only one the drivers select (``observer_type="minmax"``).

Semantics of the reference's ``fake_quant/observer/minmax.py:13-52``:
* per-channel max/min of the batch; the FIRST batch is clamped to include 0;
* later batches widen the running range; ``layer_wise`` collapses to scalars every update;
* signed bit types are symmetric: scale = max(|min/qmin|, |max/qmax|) >= eps, zero_point 0;
  unsigned types use the affine form.

On CUDA tensors the two reductions run in the fused gfx950 observer kernel
(``mquant_amd/csrc/minmax.hip``: one read of x); other tensors use torch reductions.
"""
import torch

from .base import BaseObserver


def _batch_minmax(observer, v):
    """(cur_min, cur_max) per channel, fp of v's dtype, for an activation tensor."""
    if (v.is_cuda and observer.module_type == "activation" and v.dim() != 4
            and v.dtype in (torch.float16, torch.bfloat16, torch.float32)):
        from mquant_amd import ops
        if observer.calibration_mode == "layer_wise":
            mm = ops.minmax_tensor(v)
            return mm[0].to(v.dtype), mm[1].to(v.dtype)
        mn, mx = ops.minmax_channels(v)
        return mn.to(v.dtype), mx.to(v.dtype)
    r = observer.reshape_tensor(v)
    return r.min(axis=-1).values, r.max(axis=-1).values


class MinmaxObserver(BaseObserver):
    def __init__(self, module_type, bit_type, calibration_mode):
        super().__init__(module_type, bit_type, calibration_mode)
        self.symmetric = self.bit_type.signed

    def update(self, v):
        cur_min, cur_max = _batch_minmax(self, v)
        if self.max_val is None:
            self.max_val = torch.max(cur_max, torch.zeros_like(cur_max))
        else:
            self.max_val = torch.max(cur_max, self.max_val)
        if self.min_val is None:
            self.min_val = torch.min(cur_min, torch.zeros_like(cur_min))
        else:
            self.min_val = torch.min(cur_min, self.min_val)
        self._collapse()

    def get_quantization_params(self, *args, **kwargs):
        qmin, qmax = self._bounds()
        if not self.symmetric:
            return self._asymmetric_params(self.min_val, self.max_val)
        # tensor / tensor: torch's CUDA kernel turns "tensor / python scalar" into a multiplication by the
        # reciprocal (one ulp off the IEEE quotient the reference's CPU run -- and the goldens -- have)
        lo = torch.tensor(float(qmin), dtype=self.min_val.dtype, device=self.min_val.device)
        hi = torch.tensor(float(qmax), dtype=self.max_val.dtype, device=self.max_val.device)
        scale = torch.max(torch.abs(self.min_val / lo), torch.abs(self.max_val / hi))
        scale.clamp_(self.eps)
        return scale, torch.zeros_like(self.max_val, dtype=torch.int64)
