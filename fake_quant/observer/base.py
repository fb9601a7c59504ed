"""Observer base class: running statistics of a tensor seen during calibration.

Surface of the reference's ``fake_quant/observer/base.py``: ``module_type``, ``bit_type``,
``calibration_mode``, ``max_val``, ``min_val``, ``eps``, ``reshape_tensor``, ``update``,
``get_quantization_params``.
"""
import torch


class BaseObserver:
    def __init__(self, module_type, bit_type, calibration_mode):
        self.module_type = module_type
        self.bit_type = bit_type
        self.calibration_mode = calibration_mode
        self.max_val = None
        self.min_val = None
        self.eps = torch.finfo(torch.float32).eps

    def reshape_tensor(self, v):
        """Bring the channel axis to the front: the result is (channels, everything else).

        weights: channel = dim 0; softmax: untouched; activations: channel = last dim
        (4-D NCHW inputs are first permuted to NHWC).
        """
        if not isinstance(v, torch.Tensor):
            v = torch.tensor(v)
        v = v.detach()
        if self.module_type in ("conv_weight", "linear_weight"):
            return v.reshape(v.shape[0], -1)
        if self.module_type == "softmax":
            return v
        if v.dim() == 4:
            v = v.permute(0, 2, 3, 1)
        return v.reshape(-1, v.shape[-1]).transpose(0, 1)

    # helpers shared by the concrete observers -------------------------------------------
    def _collapse(self):
        if self.calibration_mode == "layer_wise":
            self.max_val = self.max_val.max()
            self.min_val = self.min_val.min()

    def _bounds(self):
        return self.bit_type.lower_bound, self.bit_type.upper_bound

    def _asymmetric_params(self, min_val, max_val):
        qmin, qmax = self._bounds()
        span = torch.tensor(float(qmax - qmin), dtype=max_val.dtype, device=max_val.device)   # tensor / tensor: see minmax.py
        scale = (max_val - min_val) / span
        scale.clamp_(self.eps)
        zero_point = qmin - torch.round(min_val / scale)
        zero_point.clamp_(qmin, qmax)
        return scale, zero_point

    def update(self, v):
        raise NotImplementedError

    def get_quantization_params(self, *args, **kwargs):
        raise NotImplementedError
