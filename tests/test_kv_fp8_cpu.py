"""fp8 KV cache oracle.  The reference has no KV-cache quantization, so there are no
reference goldens: PARITY UNPINNED.  What can be pinned is the number format -- the oracle's OCP
e4m3fn codec against torch.float8_e4m3fn -- and the oracle's write / read formulas against the same
arithmetic spelled in torch."""
import numpy as np
import torch

import oracle


def _torch_encode(x):
    return torch.from_numpy(x).to(torch.float8_e4m3fn).view(torch.uint8).numpy()


def test_decode_matches_torch_for_all_256_codes():
    b = np.arange(256, dtype=np.uint8)
    ours = oracle.fp8_e4m3fn_decode(b)
    ref = torch.from_numpy(b).view(torch.float8_e4m3fn).float().numpy()
    assert np.array_equal(np.isnan(ours), np.isnan(ref)) and np.isnan(ours).sum() == 2      # 0x7F, 0xFF
    np.testing.assert_array_equal(ours[~np.isnan(ours)], ref[~np.isnan(ref)])
    assert ours[0x7E] == 448.0 and ours[0x01] == 2.0 ** -9 and ours[0x08] == 2.0 ** -6


def test_encode_matches_torch_on_every_rounding_boundary():
    """Every representable value, every midpoint between neighbours and one fp32 ulp either side of
    it (ties go to the even code), both signs."""
    vals = oracle.fp8_e4m3fn_decode(np.arange(0, 0x7F, dtype=np.uint8))       # 0 .. 448 ascending
    mids = (vals[:-1].astype(np.float64) + vals[1:].astype(np.float64)) / 2
    pts = [vals, mids.astype(np.float32),
           np.nextafter(mids.astype(np.float32), np.float32(1e9)), np.nextafter(mids.astype(np.float32), np.float32(-1e9))]
    x = np.concatenate(pts).astype(np.float32)
    x = np.concatenate([x, -x])
    np.testing.assert_array_equal(oracle.fp8_e4m3fn_encode(x), _torch_encode(x))


def test_encode_matches_torch_on_random_values_and_saturates():
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.normal(size=5000).astype(np.float32) * s for s in (1e-3, 0.02, 1.0, 30.0, 200.0)])
    x = np.clip(x, -448, 448)
    np.testing.assert_array_equal(oracle.fp8_e4m3fn_encode(x), _torch_encode(x))
    sat = oracle.fp8_e4m3fn_encode(np.array([449.0, 1e6, -1e6, np.inf, -np.inf], dtype=np.float32))
    assert sat.tolist() == [0x7E, 0x7E, 0xFE, 0x7E, 0xFE]          # torch would give NaN beyond 464: callers clamp first
    assert int(oracle.fp8_e4m3fn_encode(np.array([np.nan], dtype=np.float32))[0]) == 0x7F


def test_write_and_read_formulas():
    rng = np.random.default_rng(2)
    T, H, D = 33, 4, 16
    x = (rng.normal(size=(T, H, D)) * np.array([0.1, 1.0, 7.0, 40.0])[None, :, None]).astype(np.float16)
    x[0, 2, :4] = [1000.0, -1000.0, 0.0, 6e-5]                      # beyond the scaled range: saturates
    xf = x.astype(np.float32)
    scale = (np.abs(xf).max(axis=(0, 2)) / 448.0).astype(np.float32)
    scale[2] = 0.05
    q = oracle.kv_quant_fp8(xf, scale)
    want = _torch_encode(np.clip(xf / scale[None, :, None], -448, 448).astype(np.float32))
    np.testing.assert_array_equal(q, want)
    assert q[0, 2, 0] == 0x7E and q[0, 2, 1] == 0xFE
    y = oracle.kv_dequant_fp8(q, scale, mode=1)
    ref = (torch.from_numpy(q).view(torch.float8_e4m3fn).float() * torch.from_numpy(scale)[None, :, None]).half().float().numpy()
    np.testing.assert_array_equal(y, ref)
    # round trip error: 3 mantissa bits -> at most 2^-4 relative for normal values
    ok = np.abs(xf) > scale[None, :, None] * 2.0 ** -6
    ok &= np.abs(xf) <= scale[None, :, None] * 448
    assert float(np.max(np.abs(y[ok] - xf[ok]) / np.abs(xf[ok]))) <= 2.0 ** -4 + 1e-3


def test_kv_scale_check_is_per_tensor_object_not_per_address(monkeypatch):
    """advisor finding r4: the positive / finite check of kv_scale was cached on (address, version); a NEW tensor that the
    caching allocator places at a freed, already validated address skipped it.  The cache is keyed on the tensor object now."""
    from mquant_amd import ops
    monkeypatch.setattr(torch.cuda, "is_current_stream_capturing", lambda: False)
    good = torch.full((8,), 0.5)
    ops._check_kv_scale(good, 4)
    ops._check_kv_scale(good, 4)                               # cached: same object, same version
    ptr = good.data_ptr()
    bad = torch.full((8,), 0.5)
    bad[3] = -1.0
    # same address and version as a validated tensor, different object: must be checked (simulated through the cache itself)
    ops._KV_SCALE_OK[id(bad)] = ops._KV_SCALE_OK[id(good)]
    try:
        ops._check_kv_scale(bad, 4)
        raise RuntimeError("a foreign cache entry validated a bad tensor")
    except AssertionError:
        pass
    good[0] = 0.0                                              # in-place write: version bump, checked again
    try:
        ops._check_kv_scale(good, 4)
        raise RuntimeError("a version bump was not re-checked")
    except AssertionError:
        pass
    assert ptr == good.data_ptr()
