"""MI355X-native W4A8 static-quant kernels behind MQuant's ``fake_quant`` operator API.

``mquant_amd.ops``   torch-tensor front-end of the C ABI (``include/mquant_hip.h``)
``mquant_amd._lib``  ctypes loader of ``libmquant_hip.so`` (no CPU fallback)
``mquant_amd.csrc``  the hand-written gfx950 kernels
"""
__version__ = "0.1.0"
