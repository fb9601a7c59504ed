"""Integer grids used by the static quantizers.

Mirrors the surface of the reference's ``fake_quant/bit_type.py:7-50``: ``BitType`` objects
with ``bits / signed / name / lower_bound / upper_bound / range`` and the registry
``BIT_TYPE_DICT`` holding exactly uint4, int8, uint8, int16, int20, int18 (there is no
``int4`` entry, so ``ActQuantizer.configure(bits=4, static=True)`` raises ``KeyError`` as
it does upstream).
"""
from dataclasses import dataclass, field


@dataclass
class BitType:
    bits: int
    signed: bool
    name: str = field(default=None)

    def __post_init__(self):
        if self.name is None:
            self.update_name()

    def update_name(self):
        self.name = f"{'int' if self.signed else 'uint'}{self.bits}"

    @property
    def lower_bound(self) -> int:
        return -(1 << (self.bits - 1)) if self.signed else 0

    @property
    def upper_bound(self) -> int:
        return (1 << (self.bits - 1)) - 1 if self.signed else (1 << self.bits) - 1

    @property
    def range(self) -> int:
        return 1 << self.bits


BIT_TYPE_LIST = [BitType(b, s) for b, s in ((4, False), (8, True), (8, False), (16, True),
                                            (20, True), (18, True))]
BIT_TYPE_DICT = {bt.name: bt for bt in BIT_TYPE_LIST}
