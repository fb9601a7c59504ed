"""Offline weight quantization passes (reference: ``fake_quant/gptq``).

Each ``*_rtn_gptq_fwrd_plus(model, dataset, dev, dataset_name, args)`` (Qwen-VL v1: without
``dataset_name``, as upstream ``gptq/qwenvl_gptq_plus.py:620``) fake-quantizes the
wrapped Linears in place and returns ``{dotted_name: WeightQuantizer}`` like upstream.  In
addition every quantizer is attached to its ``ActQuantWrapper`` so the wrapper can recover the
integer levels and run the real W4A8 kernels (``quant_utils.attach_weight_quantizer``).
"""
from .minicpmv_gptq_plus import minicpmv_rtn_gptq_fwrd_plus  # noqa: F401
from .qwenvl_gptq_plus import qwenvl_rtn_gptq_fwrd_plus  # noqa: F401
from .internvl_gptq_plus import internvl_rtn_gptq_fwrd_plus  # noqa: F401
from .qwen2vl_gptq_plus import qwen2vl_rtn_gptq_fwrd_plus  # noqa: F401
