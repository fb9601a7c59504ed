"""Power-of-two-factor observer (registry parity; reference ``observer/ptf.py``): one shared
8-bit scale and a per-channel factor in {1,2,4,8} chosen by L2 error."""
import torch

from .base import BaseObserver
from .utils import lp_loss


class PtfObserver(BaseObserver):
    def update(self, v):
        r = self.reshape_tensor(v)
        cur_max, cur_min = r.max(axis=1).values, r.min(axis=1).values
        self.max_val = cur_max if self.max_val is None else torch.max(cur_max, self.max_val)
        self.min_val = cur_min if self.min_val is None else torch.min(cur_min, self.min_val)
        self._collapse()

    def get_quantization_params(self, inputs, *args, **kwargs):
        qmin, qmax = self._bounds()
        top, bottom = self.max_val.max(), self.min_val.min()
        scale8, zero_point = self._asymmetric_params(bottom, top)
        ladder = [scale8 / 8, scale8 / 4, scale8 / 2, scale8]   # factors 1, 2, 4, 8
        factor = torch.ones_like(self.max_val)
        for ch in range(inputs.shape[2]):
            data = inputs[..., ch].unsqueeze(-1)
            scores = []
            for s in ladder:
                deq = ((data / s + zero_point).round().clamp(qmin, qmax) - zero_point) * s
                scores.append(lp_loss(data, deq, p=2.0, reduction="all"))
            factor[ch] *= 2 ** scores.index(min(scores))
        return ladder[0] * factor, zero_point
