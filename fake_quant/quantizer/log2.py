"""Log2 quantizer for softmax outputs (registry parity; no driver builds it)."""
import torch

from .base import BaseQuantizer


class Log2Quantizer(BaseQuantizer):
    def __init__(self, bit_type, observer, module_type):
        super().__init__(bit_type, observer, module_type)
        self.softmax_mask = None

    def quant(self, inputs):
        levels = torch.round(-1 * inputs.log2())
        self.softmax_mask = levels >= 2 ** self.bit_type.bits
        return torch.clamp(levels, 0, 2 ** self.bit_type.bits - 1)

    def dequantize(self, inputs):
        out = 2 ** (-1 * inputs)
        out[self.softmax_mask] = 0
        return out
