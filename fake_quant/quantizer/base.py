"""Static quantizer base (surface of the reference's ``fake_quant/quantizer/base.py``)."""
import torch.nn as nn

_ACT_RANGE_SHAPES = {2: (1, -1), 3: (1, 1, -1), 4: (1, -1, 1, 1), 5: (1, -1, 1, 1, 1)}


class BaseQuantizer(nn.Module):
    def __init__(self, bit_type, observer, module_type):
        super().__init__()
        self.bit_type = bit_type
        self.observer = observer
        self.module_type = module_type

    def get_reshape_range(self, inputs):
        """Broadcast shape of scale / zero_point: the channel axis is the LAST dim for 2-D and
        3-D activations and dim 1 for 4-D / 5-D ones; dim 0 for weights."""
        if self.module_type == "conv_weight":
            return (-1, 1, 1, 1)
        if self.module_type == "linear_weight":
            return (-1, 1)
        if self.module_type == "activation":
            try:
                return _ACT_RANGE_SHAPES[inputs.dim()]
            except KeyError:
                raise NotImplementedError from None
        raise NotImplementedError

    def update_quantization_params(self, *args, **kwargs):
        pass

    def quant(self, inputs, scale=None, zero_point=None):
        raise NotImplementedError

    def dequantize(self, inputs, scale=None, zero_point=None):
        raise NotImplementedError

    def forward(self, inputs):
        dtype = inputs.dtype
        return self.dequantize(self.quant(inputs.float())).to(dtype)
