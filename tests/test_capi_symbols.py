"""The C-ABI shared library loads on a CPU-only box and exports every symbol declared in
include/mquant_hip.h (no compute calls are made here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header="mquant_hip.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mq_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_hot_path_entry_points():
    syms = declared_symbols()
    for name in ["mq_quantize_act_i8", "mq_fakequant_act", "mq_hadamard", "mq_hadamard_quant_i8",
                 "mq_gemm_w4a8", "mq_gemm_w4a8_ws", "mq_gemm_w4a8_i32", "mq_minmax_channels",
                 "mq_minmax_tensor", "mq_pack_i4", "mq_unpack_i4", "mq_prepack_w4", "mq_prepack_w8",
                 "mq_weight_levels", "mq_version", "mq_last_error"]:
        assert name in syms, name


def test_library_exports_every_declared_symbol():
    from mquant_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing
    assert lib.mq_version() >= 100


def test_python_binding_covers_the_header():
    from mquant_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_symbols()
    _lib.load()


def test_bench_probe_is_its_own_library():
    """The MFMA burn probe is bench-only code: declared in include/mquant_bench.h, built into libmquant_bench.so, absent
    from the product library and from its binding table."""
    from mquant_amd import _lib
    assert sorted(_lib.BENCH_SIGNATURES) == declared_symbols("mquant_bench.h")
    assert not set(_lib.BENCH_SIGNATURES) & set(_lib.SIGNATURES)
    if not os.path.exists(_lib.BENCH_LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    _lib.load_bench()
    assert not hasattr(ctypes.CDLL(_lib.LIB_PATH), "mq_bench_mfma_burn")


def test_missing_library_is_a_loud_error(monkeypatch, tmp_path):
    from mquant_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.MQuantHipError):
        _lib.load()


def test_cpu_tensors_are_rejected_by_every_op():
    import torch
    from mquant_amd import ops
    from mquant_amd._lib import MQuantHipError
    x = torch.zeros(4, 128)
    for fn in (lambda: ops.quantize_act_i8(x, 0.1), lambda: ops.fakequant_act(x, 0.1),
               lambda: ops.hadamard(x, 128, 1, None), lambda: ops.minmax_tensor(x),
               lambda: ops.pack_i4(torch.zeros(2, 8, dtype=torch.int8)),
               lambda: ops.gemm_w4a8_i32(torch.zeros(4, 128, dtype=torch.int8), torch.zeros(1024, dtype=torch.uint8), 4, 16)):
        with pytest.raises(MQuantHipError):
            fn()


def test_checkpoint_record_helpers_need_no_gpu():
    import torch
    from mquant_amd import checkpoint
    meta = checkpoint._meta(version=1, w_bits=4, a_bits=8, N=16, K=64, in_features=64, had_K=12, split=1)
    assert checkpoint.read_meta(meta)["had_K"] == 12 and checkpoint.read_meta(meta)["msq"] == 0
    tensors = {"model.layers.0.mlp.down_proj.qweight": torch.zeros((16, 32), dtype=torch.uint8),
               "model.layers.0.mlp.down_proj.meta": meta, "model.norm.weight": torch.ones(4)}
    recs = checkpoint.split_records(tensors)
    assert list(recs) == ["model.layers.0.mlp.down_proj"] and set(recs["model.layers.0.mlp.down_proj"]) == {"qweight", "meta"}
    assert checkpoint.split_records(tensors, prefix="model.") .keys() == {"layers.0.mlp.down_proj"}
