"""The GEMM tile plan (host arithmetic, csrc/gemm_w4a8.hip make_plan) for tiled int8 activations and int4 weights, pinned
for the shapes of the benchmark model (the choices every profile under profiles/ was taken with), the 72B shapes, and the
shapes the round-3 spill rule was introduced for (a handful of 256 x 256 tiles in one more round -> 192 x 128 tiles)."""
import ctypes as C

import pytest

PIPE, WS64, WS96, WS128, WS192 = 20, 53, 54, 51, 52   # PIPE: the 256 x 256 tile (ping-pong kernel since round 4); WS*: the
#                                                       wave-specialised tiles with V_MFMA_I32_16X16X64_I8 math waves (round 5) and the
#                                                       slab-free epilogue (round 6; ids 47 / 48 / 45 / 46 are their slab twins)


def _plan(M, N, K):
    from mquant_amd import _lib
    lib = _lib.load()
    tile, splits = C.c_int(-1), C.c_int(-1)
    rc = lib.mq_gemm_debug_plan(M, N, K, 4, 1, 1, C.byref(tile), C.byref(splits))
    assert rc == 0
    return tile.value, splits.value


@pytest.mark.parametrize("name,M,N,K,want", [
    # Qwen2-VL-7B, 1 x 448^2 + 512 tokens (workload.qwen2vl_7b_specs; q|k|v and gate|up fused)
    ("vis.qkv", 1024, 3840, 1280, WS128), ("vis.proj", 1024, 1280, 1280, WS64), ("vis.fc1", 1024, 5120, 1280, WS192),
    ("vis.fc2", 1024, 1280, 5120, WS64), ("llm.qkv", 768, 4608, 3584, WS128), ("llm.o", 768, 3584, 3584, WS96),
    ("llm.gate_up", 768, 37888, 3584, PIPE), ("llm.down", 768, 3584, 19968, WS96),
    # Qwen2-VL-72B
    ("72b.qkv", 768, 10240, 8192, WS128), ("72b.o", 768, 8192, 8192, WS192), ("72b.gate_up", 768, 59136, 8192, PIPE),
    ("72b.down", 768, 8192, 30720, WS192),
    # where a few 256 x 256 tiles used to spill into a second round
    ("qwenvl.w1w2", 768, 22016, 4096, WS192), ("internvl2.wqkv b4", 3072, 6144, 4096, WS192),
    ("internvl2.w1w3 b4", 3072, 28672, 4096, PIPE),
])
def test_plan_of_the_model_shapes(name, M, N, K, want):
    tile, splits = _plan(M, N, K)
    assert (tile, splits) == (want, 1), name


def test_few_tiles_and_a_long_reduction_still_split_k():
    tile, splits = _plan(256, 2048, 19968)
    assert splits > 1


@pytest.mark.parametrize("M", [1, 8, 16, 32, 33, 64])
def test_a_few_rows_take_the_weight_streaming_kernels(M):
    """Generation steps (W4, tiled activations), gemm_skinny.hip.  Up to two row tiles: id 61, the K slices are the eight waves of a
    workgroup (short reductions with enough channel pairs: one launch; long reductions: a few workgroup slices through the workspace).
    Three or four row tiles on long reductions over few channel tiles: id 60, one wave per pair and slice."""
    for N, K in ((4608, 3584), (3584, 3584), (37888, 3584)):
        assert (_plan(M, N, K) == (61, 1)) == (M <= 32), (M, N, K, _plan(M, N, K))
    assert _plan(M, 3584, 19968) == ((61, 5) if M <= 32 else (60, 28))
    assert _plan(65, 3584, 19968)[0] not in (60, 61)
    if M <= 16:
        assert _plan(M, 200, 1280) == (60, 5)          # few pairs, short reduction: slices across workgroups
