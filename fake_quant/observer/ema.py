"""Exponential-moving-average observer (registry parity; reference ``observer/ema.py``)."""
import torch

from .base import BaseObserver


class EmaObserver(BaseObserver):
    def __init__(self, module_type, bit_type, calibration_mode, ema_sigma=0.01):
        super().__init__(module_type, bit_type, calibration_mode)
        self.ema_sigma = ema_sigma
        self.symmetric = self.bit_type.signed

    def _blend(self, old, cur):
        return cur if old is None else old + self.ema_sigma * (cur - old)

    def update(self, v):
        r = self.reshape_tensor(v)
        self.max_val = self._blend(self.max_val, r.max(axis=1).values)
        self.min_val = self._blend(self.min_val, r.min(axis=1).values)
        self._collapse()

    def get_quantization_params(self, *args, **kwargs):
        if not self.symmetric:
            return self._asymmetric_params(self.min_val, self.max_val)
        qmin, qmax = self._bounds()
        span = torch.max(-self.min_val, self.max_val)
        half = torch.tensor(float(qmax - qmin) / 2, dtype=span.dtype, device=span.device)   # tensor / tensor: see minmax.py
        scale = span / half
        scale.clamp_(self.eps)
        return scale, torch.zeros_like(span, dtype=torch.int64)
