"""OMSE observer: shrink search for the range minimising the L2 quantisation error
(registry parity; reference ``observer/omse.py``; LAPQ, arXiv:1911.07190)."""
import torch

from .base import BaseObserver
from .utils import lp_loss


class OmseObserver(BaseObserver):
    def update(self, v):
        r = self.reshape_tensor(v)
        cur_max, cur_min = r.max(axis=1).values, r.min(axis=1).values
        self.max_val = cur_max if self.max_val is None else torch.max(cur_max, self.max_val)
        self.min_val = cur_min if self.min_val is None else torch.min(cur_min, self.min_val)
        self._collapse()

    def get_quantization_params(self, inputs):
        qmin, qmax = self._bounds()
        hi, lo = self.max_val, self.min_val
        best = 1e10
        scale = zero_point = None
        for step in range(90):
            shrink = 1.0 - step * 0.01
            new_max, new_min = hi * shrink, lo * shrink
            s, z = self._asymmetric_params(new_min, new_max)
            deq = ((inputs / s + z).round().clamp(qmin, qmax) - z) * s
            score = lp_loss(inputs, deq, p=2.0, reduction="all")
            if score < best:
                best = score
                self.max_val, self.min_val = new_max, new_min
                scale, zero_point = s, z
        return scale, zero_point
