"""ctypes binding of ``libmquant_hip.so`` (the C ABI declared in ``include/mquant_hip.h``).

There is no CPU fallback: if the shared library is missing or a call fails, this module
raises.  Build the library with ``python __graft_entry__.py`` (or ``make -C mquant_amd/csrc``).
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (must load first: the library then binds to torch's HIP runtime)

_HERE = os.path.dirname(os.path.abspath(__file__))
# MQUANT_HIP_LIB selects another build of the same ABI (kernel experiments); default = in-tree .so
LIB_PATH = os.environ.get("MQUANT_HIP_LIB") or os.path.join(_HERE, "libmquant_hip.so")

MQ_F16, MQ_BF16, MQ_F32 = 0, 1, 2

_vp = C.c_void_p
_l = C.c_long
_i = C.c_int
_f = C.c_float

# name -> (restype, argtypes); mirrors include/mquant_hip.h one to one
SIGNATURES = {
    "mq_version": (_i, []),
    "mq_last_error": (C.c_char_p, []),
    "mq_device_info": (_i, [C.c_char_p, C.c_size_t, C.POINTER(C.c_int)]),
    "mq_quantize_act_i8": (_i, [_vp, _i, _l, _l, _l, _f, _f, _vp, _vp, _vp, _i, _vp, _vp, _l, _l, _vp]),
    "mq_fakequant_act": (_i, [_vp, _i, _l, _l, _l, _f, _f, _vp, _vp, _vp, _i, _vp, _l, _vp]),
    "mq_hadamard": (_i, [_vp, _i, _l, _l, _l, _l, _i, _vp, _i, _vp, _l, _vp]),
    "mq_hadamard_quant_i8": (_i, [_vp, _i, _l, _l, _l, _l, _i, _vp, _i, _f, _f, _vp, _i, _vp, _vp, _l, _l, _vp]),
    "mq_hadamard_debug_threads": (_i, [_i]),
    "mq_hadamard_prepared_bytes": (C.c_size_t, [_i]),
    "mq_hadamard_prepare": (_i, [_vp, _i, _vp, _vp]),
    "mq_pack_i4": (_i, [_vp, _l, _l, _vp, _vp]),
    "mq_unpack_i4": (_i, [_vp, _l, _l, _vp, _vp]),
    "mq_weight_levels": (_i, [_vp, _i, _l, _l, _l, _vp, _i, _i, _vp, _vp]),
    "mq_act_hadamard_quant_i8": (_i, [_vp, _vp, _i, _i, _l, _l, _l, _l, _i, _vp, _i, _f, _f, _vp, _i, _vp, _vp, _l, _l, _vp]),
    "mq_quantize_act_dyn_i8": (_i, [_vp, _i, _l, _l, _l, _i, _f, _i, _vp, _vp, _vp, _l, _l, _vp]),
    "mq_quantize_act_dyn_asym_i8": (_i, [_vp, _i, _l, _l, _l, _i, _f, _vp, _vp, _vp, _vp, _l, _l, _vp]),
    "mq_quantize_act_range_i8": (_i, [_vp, _i, _l, _l, _l, _i, _f, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _l, _l, _vp]),
    "mq_quantize_act_group_i8": (_i, [_vp, _i, _l, _l, _l, _i, _i, _f, _vp, _vp, _l, _l, _vp]),
    "mq_gemm_w4a8_groupscale": (_i, [_vp, _l, _vp, _i, _l, _l, _l, _vp, _l, _i, _vp, _vp, _vp, _i, _l, _vp]),
    "mq_quantize_act_group_asym_i8": (_i, [_vp, _i, _l, _l, _l, _i, _i, _f, _vp, _vp, _vp, _vp, _l, _l, _vp]),
    "mq_gemm_w4a8_groupscale_asym": (_i, [_vp, _l, _vp, _i, _l, _l, _l, _vp, _vp, _vp, _l, _i, _vp, _vp, _vp, _i, _l, _vp]),
    "mq_rank1_add_cast": (_i, [_vp, _l, _l, _l, _vp, _vp, _vp, _i, _l, _vp]),
    "mq_gemm_w4a8_rope_ws": (_i, [_vp, _l, _vp, _i, _l, _l, _l, _f, _f, _vp, _vp, _vp, _vp, _vp, _l, _i, _vp, _i, _l, _vp]),
    "mq_gemm_w4a8_act_ws": (_i, [_vp, _l, _vp, _i, _l, _l, _l, _f, _f, _vp, _vp, _vp, _vp, _i, _vp, _i, _l, _vp]),
    "mq_gemm_w4a8_wgroupscale": (_i, [_vp, _l, _vp, _i, _l, _l, _l, _vp, _l, _i, _f, _f, _vp, _vp, _vp, _vp, _vp, _i, _l, _vp]),
    "mq_act_rowsum_scaled": (_i, [_vp, _l, _l, _l, _f, _f, _vp, _vp, _vp, _vp]),
    "mq_gemm_w4a8_residual_ws": (_i, [_vp, _l, _vp, _i, _l, _l, _l, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _l, _vp, _i, _l, _vp, C.c_size_t, _vp]),
    "mq_gemm_w4a8_rank2_ws": (_i, [_vp, _l, _vp, _i, _l, _l, _l, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _l, _vp, C.c_size_t, _vp]),
    "mq_gemm_w4a8_rowscale_ws": (_i, [_vp, _l, _vp, _i, _l, _l, _l, _vp, _vp, _vp, _vp, _vp, _vp, _i, _l, _vp, C.c_size_t, _vp]),
    "mq_rope_inplace": (_i, [_vp, _i, _l, _i, _i, _l, _vp, _vp, _vp]),
    "mq_rmsn_quantize_i8": (_i, [_vp, _i, _l, _l, _l, _f, _f, _f, _f, _vp, _vp, _l, _vp, _l, _l, _vp]),
    "mq_wquant_sym": (_i, [_vp, _i, _l, _l, _l, _i, _i, _f, _i, _f, _vp, _vp, _vp, _vp, _l, _vp]),
    "mq_rotate_f64": (_i, [_vp, _i, _l, _l, _l, _vp, _i, _vp, _vp]),
    "mq_kv_quant_fp8": (_i, [_vp, _i, _l, _i, _i, _l, _vp, _vp, _l, _vp]),
    "mq_kv_quant_fp8_readback": (_i, [_vp, _i, _l, _i, _i, _l, _vp, _vp, _l, _vp, _l, _vp]),
    "mq_kv_dequant_fp8": (_i, [_vp, _l, _i, _i, _l, _vp, _vp, _i, _l, _vp]),
    "mq_attn_prefill_fp8kv": (_i, [_vp, _i, _l, _i, _i, _i, _l, _vp, _l, _vp, _f, _i, _vp, _l, _vp]),
    "mq_attn_debug_waves": (_i, [_i]),
    "mq_gemv_f16": (_i, [_vp, _i, _i, _l, _l, _vp, _l, _l, _vp, _l, _vp]),
    "mq_attn_prefill": (_i, [_vp, _i, _l, _i, _i, _i, _l, _vp, _vp, _l, _f, _i, _vp, _l, _vp]),
    "mq_attn_prefill_quant_i8": (_i, [_vp, _i, _l, _i, _i, _i, _l, _vp, _vp, _l, _vp, _l, _vp, _f, _i, _f, _f, _vp, _vp, _l, _l, _vp]),
    "mq_gptq_block": (_i, [_vp, _l, _i, _l, _vp, _l, _vp, _i, _vp, _l, _vp, _l, _vp]),
    "mq_prepack_w4": (_i, [_vp, _l, _l, _i, _vp, _vp]),
    "mq_prepack_w8": (_i, [_vp, _l, _l, _i, _vp, _vp]),
    "mq_prepacked_bytes": (C.c_size_t, [_l, _l, _i]),
    "mq_gemm_w4a8": (_i, [_vp, _l, _vp, _i, _l, _l, _l, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _i, _l, _vp]),
    "mq_gemm_w4a8_i32": (_i, [_vp, _l, _vp, _i, _l, _l, _l, _vp, _l, _vp]),
    "mq_gemm_w4a8_ws": (_i, [_vp, _l, _vp, _i, _l, _l, _l, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _i, _l, _vp, C.c_size_t, _vp]),
    "mq_gemm_w4a8_i32_ws": (_i, [_vp, _l, _vp, _i, _l, _l, _l, _vp, _l, _vp, C.c_size_t, _vp]),
    "mq_gemm_debug_force": (_i, [_i, _i]),
    "mq_debug_act_table": (_i, [_i, _i, _vp, _vp, _l, _vp, _vp, _vp]),
    "mq_gemm_debug_plan": (_i, [_l, _l, _l, _i, _i, _i, _vp, _vp]),
    "mq_minmax_channels": (_i, [_vp, _i, _l, _l, _l, _l, _vp, _vp, _vp]),
    "mq_minmax_tensor": (_i, [_vp, _i, _l, _l, _l, _l, _vp, _vp]),
}

#: bench-only library (include/mquant_bench.h, csrc/bench_probe.hip): loaded by bench.py alone
BENCH_LIB_PATH = os.path.join(_HERE, "libmquant_bench.so")
BENCH_SIGNATURES = {
    "mq_bench_mfma_burn": (_i, [_i, _vp, _i, _i, _vp, _vp, _vp]),
    "mq_bench_last_error": (C.c_char_p, []),
}

_lib = None
_bench = None


class MQuantHipError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load the HIP library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MQuantHipError(
                f"{LIB_PATH} not found: the W4A8 path has no CPU fallback. "
                "Build it with `python -c 'import __graft_entry__ as g; g.build()'`.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the ABI and the header drift apart
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def call(name: str, *args) -> None:
    """Invoke an ``int``-returning entry point and raise on a non-zero status."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        msg = lib.mq_last_error().decode("utf-8", "replace")
        raise MQuantHipError(f"{name} failed (status {rc}): {msg}")


def load_bench() -> C.CDLL:
    """The bench-only probe library (register-only MFMA burn); raises if it has not been built."""
    global _bench
    if _bench is None:
        if not os.path.exists(BENCH_LIB_PATH):
            raise MQuantHipError(f"{BENCH_LIB_PATH} not found: build it with `make -C mquant_amd/csrc`")
        lib = C.CDLL(BENCH_LIB_PATH)
        for name, (res, args) in BENCH_SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _bench = lib
    return _bench


def call_bench(name: str, *args) -> None:
    lib = load_bench()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise MQuantHipError(f"{name} failed (status {rc}): {lib.mq_bench_last_error().decode('utf-8', 'replace')}")
