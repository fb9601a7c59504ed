"""Percentile observer with EMA smoothing, layer_wise only (registry parity; reference
``observer/percentile.py``)."""
import numpy as np
import torch

from .base import BaseObserver


class PercentileObserver(BaseObserver):
    def __init__(self, module_type, bit_type, calibration_mode, percentile_sigma=0.01,
                 percentile_alpha=0.99999):
        super().__init__(module_type, bit_type, calibration_mode)
        self.percentile_sigma = percentile_sigma
        self.percentile_alpha = percentile_alpha
        self.symmetric = self.bit_type.signed

    def _quantile(self, flat, q):
        try:
            return torch.quantile(flat.float(), q)
        except Exception:  # torch.quantile refuses very large inputs
            return torch.tensor(np.percentile(flat.cpu(), q * 100), device=flat.device,
                                dtype=torch.float32)

    def update(self, v):
        assert self.calibration_mode == "layer_wise"  # channel-wise is far too slow
        flat = self.reshape_tensor(v).reshape(-1)
        cur_max = self._quantile(flat, self.percentile_alpha)
        cur_min = self._quantile(flat, 1.0 - self.percentile_alpha)
        if self.max_val is None:
            self.max_val = torch.max(cur_max, torch.zeros_like(cur_max))
        else:
            self.max_val = self.max_val + self.percentile_sigma * (cur_max - self.max_val)
        if self.min_val is None:
            self.min_val = torch.min(cur_min, torch.zeros_like(cur_min))
        else:
            self.min_val = self.min_val + self.percentile_sigma * (cur_min - self.min_val)

    def get_quantization_params(self, *args, **kwargs):
        if not self.symmetric:
            return self._asymmetric_params(self.min_val, self.max_val)
        qmin, qmax = self._bounds()
        lo = torch.tensor(float(qmin), dtype=self.min_val.dtype, device=self.min_val.device)   # tensor / tensor: see minmax.py
        hi = torch.tensor(float(qmax), dtype=self.max_val.dtype, device=self.max_val.device)
        scale = torch.max(torch.abs(self.min_val / lo), torch.abs(self.max_val / hi))
        scale.clamp_(self.eps)
        return scale, torch.zeros_like(self.max_val, dtype=torch.int64)
