"""Prefill attention that reads the e4m3 KV cache directly (``mq_attn_prefill_fp8kv``).  The reference
has no attention kernel and no cache quantizer: PARITY UNPINNED.  The checker restated here is softmax attention in
float64 over the DEQUANTISED cache (value * per-head scale), which is what dequantise-on-read followed by SDPA
computes; the kernel multiplies the e4m3 values themselves and folds the scales into the score / output scale, so
the two differ by floating-point rounding only (P and the output are rounded to q's dtype, accumulation is fp32)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
torch.set_grad_enabled(False)


def _case(seed, T, H, HKV, dtype, q_gain=1.0):
    from mquant_amd import ops
    g = torch.Generator(device=DEV).manual_seed(seed)
    D = 128
    qkv = (torch.randn(T, (H + 2 * HKV) * D, generator=g, device=DEV) * 0.8).to(dtype)
    qkv[:, :H * D] *= q_gain
    # per-head magnitudes differ by > 100x: a scale that is wrong for one head shows immediately
    gain = torch.tensor([0.05, 1.0, 6.0, 20.0] * HKV, device=DEV)[:2 * HKV].repeat_interleave(D)
    qkv[:, H * D:] = (qkv[:, H * D:].float() * gain).to(dtype)
    q = qkv[:, :H * D].view(T, H, D)                                   # column slices of the fused output, read in place
    kv = qkv[:, H * D:].view(T, 2 * HKV, D)
    scale = ops.kv_scale_from_absmax(kv)
    cache = ops.kv_quant_fp8(kv, scale)
    return q, cache, scale


def _ref(q, cache, scale, causal):
    """float64 softmax attention over the dequantised cache -> [T, H * D]"""
    T, H, D = q.shape
    HKV = cache.shape[1] // 2
    kvd = cache.float().double() * scale.double()[None, :, None]
    k, v = kvd[:, :HKV], kvd[:, HKV:]
    rep = H // HKV
    k = k.repeat_interleave(rep, dim=1).permute(1, 0, 2)               # [H, T, D]
    v = v.repeat_interleave(rep, dim=1).permute(1, 0, 2)
    s = torch.einsum("thd,hkd->htk", q.double(), k) * D ** -0.5
    if causal:
        s = s.masked_fill(torch.ones(T, T, device=q.device, dtype=torch.bool).triu(1), float("-inf"))
    o = torch.softmax(s, dim=-1) @ v                                   # [H, T, D]
    return o.permute(1, 0, 2).reshape(T, H * D)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("T,H,HKV,causal", [(768, 28, 4, True), (768, 64, 8, True), (1, 4, 2, True), (33, 8, 8, True),
                                            (129, 4, 1, True), (500, 8, 2, False), (2048, 8, 2, True)])
def test_attention_over_the_fp8_cache_equals_attention_over_the_dequantised_cache(dtype, T, H, HKV, causal):
    from mquant_amd import ops
    q, cache, scale = _case(T + H, T, H, HKV, dtype)
    got = ops.attn_prefill_fp8kv(q, cache, scale, causal=causal)
    assert got.shape == (T, H * 128) and got.dtype == dtype
    want = _ref(q, cache, scale, causal)
    err = (got.double() - want).abs()
    # per head: the rounding of P and of the output to 10 (fp16) / 7 (bf16) mantissa bits, relative to the head's range
    tol = 2.5e-3 if dtype == torch.float16 else 1.6e-2
    per_head = want.view(T, H, 128).abs().amax(dim=(0, 2)).clamp_min(1e-9)
    rel = err.view(T, H, 128).amax(dim=(0, 2)) / per_head
    assert float(rel.max()) < tol, rel.cpu().numpy()
    cos = F.cosine_similarity(got.double().flatten(), want.flatten(), dim=0)
    assert float(cos) > (0.999999 if dtype == torch.float16 else 0.99995)


def test_peaked_scores_and_the_first_rows():
    """Large logits (one key dominates), and the causal rows 0 and 1 that attend to one and two keys."""
    from mquant_amd import ops
    T, H, HKV = 300, 4, 2
    q, cache, scale = _case(5, T, H, HKV, torch.float16, q_gain=6.0)
    got = ops.attn_prefill_fp8kv(q, cache, scale, causal=True).double()
    want = _ref(q, cache, scale, True)
    assert torch.isfinite(got).all()
    v0 = cache.float().double()[0, HKV:] * scale.double()[HKV:, None]   # row 0 attends to key 0 only: O = V[0]
    np.testing.assert_allclose(got[0].view(H, 128).cpu().numpy(), v0.repeat_interleave(H // HKV, dim=0).cpu().numpy(),
                               rtol=2 ** -10, atol=1e-6)
    per_head = want.view(T, H, 128).abs().amax(dim=(0, 2))
    rel = (got - want).abs().view(T, H, 128).amax(dim=(0, 2)) / per_head
    assert float(rel.max()) < 4e-3, rel


def test_it_agrees_with_dequantise_then_sdpa_in_half_precision():
    """The path it replaces: mq_kv_dequant_fp8 -> fp16 K / V in HBM -> torch SDPA."""
    from mquant_amd import ops
    T, H, HKV = 768, 28, 4
    q, cache, scale = _case(11, T, H, HKV, torch.float16)
    got = ops.attn_prefill_fp8kv(q, cache, scale, causal=True).float()
    kvd = ops.kv_dequant_fp8(cache, scale, torch.float16)
    k, v = kvd[:, :HKV], kvd[:, HKV:]
    rep = H // HKV
    o = F.scaled_dot_product_attention(q.permute(1, 0, 2)[None], k.repeat_interleave(rep, 1).permute(1, 0, 2)[None],
                                       v.repeat_interleave(rep, 1).permute(1, 0, 2)[None], is_causal=True)
    o = o[0].permute(1, 0, 2).reshape(T, H * 128).float()
    assert float((got - o).abs().max() / o.abs().max()) < 5e-3       # two fp16 pipelines, each ~1e-3 from the exact result
    assert float(F.cosine_similarity(got.flatten(), o.flatten(), dim=0)) > 0.999999


def test_out_buffer_and_strides():
    from mquant_amd import ops
    T, H, HKV = 200, 8, 2
    q, cache, scale = _case(2, T, H, HKV, torch.float16)
    wide = torch.full((T, H * 128 + 64), 7.0, device=DEV, dtype=torch.float16)
    ops.attn_prefill_fp8kv(q, cache, scale, out=wide[:, :H * 128])
    want = ops.attn_prefill_fp8kv(q.contiguous(), cache, scale)
    assert torch.equal(wide[:, :H * 128], want) and bool((wide[:, H * 128:] == 7.0).all())


def test_bad_arguments_are_refused():
    from mquant_amd import ops
    from mquant_amd._lib import MQuantHipError as MQuantError
    q, cache, scale = _case(1, 16, 4, 2, torch.float16)
    with pytest.raises((MQuantError, AssertionError)):
        ops.attn_prefill_fp8kv(q.float(), cache, scale)
    with pytest.raises((MQuantError, AssertionError)):
        ops.attn_prefill_fp8kv(q[:, :, :64].contiguous(), cache[:, :, :64].contiguous(), scale)
    with pytest.raises((MQuantError, AssertionError)):
        ops.attn_prefill_fp8kv(q[:, :3].contiguous(), cache, scale)     # 3 heads over 2 kv heads


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("T,H,HKV,causal,D", [(768, 28, 4, True, 128), (768, 64, 8, True, 128), (1, 4, 2, True, 128),
                                              (97, 8, 8, True, 128), (500, 8, 2, False, 128), (1500, 4, 1, True, 128),
                                              (1024, 16, 16, False, 80), (333, 4, 4, False, 80), (200, 6, 2, True, 80),
                                              (1, 2, 2, False, 80)])
def test_attention_over_unquantised_k_v(dtype, T, H, HKV, causal, D):
    """mq_attn_prefill: the same kernel with 16-bit K / V read in place from the fused q|k|v output (head_dim 128: the
    decoder; 80: Qwen2-VL's vision tower, non-causal); checker = float64 softmax attention."""
    from mquant_amd import ops
    g = torch.Generator(device=DEV).manual_seed(T + 3 * H)
    qkv = (torch.randn(T, (H + 2 * HKV) * D, generator=g, device=DEV) * 0.9).to(dtype)
    q = qkv[:, :H * D].view(T, H, D)
    k = qkv[:, H * D:(H + HKV) * D].view(T, HKV, D)
    v = qkv[:, (H + HKV) * D:].view(T, HKV, D)
    got = ops.attn_prefill(q, k, v, causal=causal)
    rep = H // HKV
    kk = k.double().repeat_interleave(rep, dim=1).permute(1, 0, 2)
    vv = v.double().repeat_interleave(rep, dim=1).permute(1, 0, 2)
    s = torch.einsum("thd,hkd->htk", q.double(), kk) * D ** -0.5
    if causal:
        s = s.masked_fill(torch.ones(T, T, device=DEV, dtype=torch.bool).triu(1), float("-inf"))
    want = (torch.softmax(s, dim=-1) @ vv).permute(1, 0, 2).reshape(T, H * D)
    tol = 2.5e-3 if dtype == torch.float16 else 1.6e-2
    assert float((got.double() - want).abs().max() / want.abs().max()) < tol
    # and against torch SDPA in the same dtype (the op it replaces in the whole-prefill glue)
    o = F.scaled_dot_product_attention(q.permute(1, 0, 2)[None], k.permute(1, 0, 2)[None], v.permute(1, 0, 2)[None],
                                       is_causal=causal, enable_gqa=True)[0].permute(1, 0, 2).reshape(T, H * D)
    assert float((got.float() - o.float()).abs().max() / o.float().abs().max()) < 2 * tol


@pytest.mark.parametrize("variant,T,H,HKV,D,causal", [("fp16", 768, 28, 4, 128, True), ("fp8", 768, 28, 4, 128, True),
                                                      ("fp16", 1024, 16, 16, 80, False), ("fp16", 77, 4, 2, 128, True),
                                                      ("fp8", 130, 8, 8, 128, False), ("bf16", 300, 8, 4, 80, True)])
@pytest.mark.parametrize("tiled", [True, False])
def test_fused_output_quantizer_equals_attention_then_quantize(variant, T, H, HKV, D, causal, tiled):
    """mq_attn_prefill_quant_i8: the int8 levels of the next Linear's static quantizer straight from the attention store
    -- bit for bit what mq_quantize_act_i8 makes of the 16-bit attention output, two scales selected by the token-type
    mask (MSQ), tiled and row-major destinations."""
    from mquant_amd import ops
    dtype = torch.bfloat16 if variant == "bf16" else torch.float16
    g = torch.Generator(device=DEV).manual_seed(T + H + D)
    qkv = (torch.randn(T, (H + 2 * HKV) * D, generator=g, device=DEV) * 0.9).to(dtype)
    q = qkv[:, :H * D].view(T, H, D)
    k = qkv[:, H * D:(H + HKV) * D].view(T, HKV, D)
    v = qkv[:, (H + HKV) * D:].view(T, HKV, D)
    sel = (torch.arange(T, device=DEV) % 3 == 1).to(torch.uint8)
    if variant == "fp8":
        kv = qkv[:, H * D:].view(T, 2 * HKV, D)
        scale = ops.kv_scale_from_absmax(kv)
        cache = ops.kv_quant_fp8(kv, scale)
        o = ops.attn_prefill_fp8kv(q, cache, scale, causal=causal)
        kw = dict(kv_cache=cache, kv_scale=scale)
    else:
        o = ops.attn_prefill(q, k, v, causal=causal)
        kw = dict(k=k, v=v)
    s0 = float(o.float().abs().max()) / 127.0 * 0.8          # some rows saturate
    s1 = 0.37 * s0
    want, _ = ops.quantize_act_i8(o, s0, s1, row_sel=sel, tiled=tiled)
    got = ops.attn_prefill_quant_i8(q, s0, s1, causal=causal, row_sel=sel, tiled=tiled, **kw)
    a = got.to_rows() if tiled else got
    b = want.to_rows() if tiled else want
    assert a.shape == b.shape == (T, H * D)
    assert torch.equal(a, b)
    assert int(a.abs().max()) >= 127


@pytest.mark.parametrize("waves", [2, 4])
@pytest.mark.parametrize("variant,T,H,HKV,D,causal", [("fp8", 768, 28, 4, 128, True), ("fp16", 333, 8, 2, 128, True),
                                                      ("fp16", 1024, 16, 16, 80, False), ("fp8", 65, 4, 4, 128, False)])
def test_both_workgroup_widths_give_the_same_attention(waves, variant, T, H, HKV, D, causal):
    """The keys are split over 4 or 2 waves of a workgroup (chosen by shape; forced here through the test hook): each split
    against the float64 checker, and the two against each other within half-precision rounding (the partial softmax states
    merge in a different grouping)."""
    from mquant_amd import ops
    from mquant_amd._lib import call
    g = torch.Generator(device=DEV).manual_seed(T + D)
    qkv = (torch.randn(T, (H + 2 * HKV) * D, generator=g, device=DEV) * 0.9).half()
    q = qkv[:, :H * D].view(T, H, D)
    k = qkv[:, H * D:(H + HKV) * D].view(T, HKV, D)
    v = qkv[:, (H + HKV) * D:].view(T, HKV, D)
    try:
        call("mq_attn_debug_waves", waves)
        if variant == "fp8":
            kv = qkv[:, H * D:].view(T, 2 * HKV, D)
            scale = ops.kv_scale_from_absmax(kv)
            cache = ops.kv_quant_fp8(kv, scale)
            got = ops.attn_prefill_fp8kv(q, cache, scale, causal=causal)
            want = _ref(q, cache, scale, causal)
        else:
            got = ops.attn_prefill(q, k, v, causal=causal)
            rep = H // HKV
            s = torch.einsum("thd,hkd->htk", q.double(), k.double().repeat_interleave(rep, 1).permute(1, 0, 2)) * D ** -0.5
            if causal:
                s = s.masked_fill(torch.ones(T, T, device=DEV, dtype=torch.bool).triu(1), float("-inf"))
            want = (torch.softmax(s, dim=-1) @ v.double().repeat_interleave(rep, 1).permute(1, 0, 2)).permute(1, 0, 2).reshape(T, H * D)
    finally:
        call("mq_attn_debug_waves", 0)
    assert float((got.double() - want).abs().max() / want.abs().max()) < 2.5e-3


@pytest.mark.parametrize("variant", ["fp16", "bf16", "fp8"])
@pytest.mark.parametrize("T", [33, 64, 65, 97, 160, 161, 352, 1000])
def test_paired_shallow_tiles_cover_every_tile_count(variant, T):
    """Causal launches may give a workgroup TWO query tiles of the shallow half (round 6; taken by shape at the 7B prefill, forced
    here through the test hook: 4 = paired, 5 = one tile per workgroup).  Tile counts 2, 3, 5, 6, 11 and 32 walk every branch of
    the row map -- an odd shallow half leaves its middle tile alone --; ragged last tiles; each form against the float64 checker,
    and the fused int8 store against attention-then-quantize under the pairing."""
    from mquant_amd import ops
    from mquant_amd._lib import call
    H, HKV, D = 6, 2, 128
    dtype = torch.bfloat16 if variant == "bf16" else torch.float16
    g = torch.Generator(device=DEV).manual_seed(T)
    qkv = (torch.randn(T, (H + 2 * HKV) * D, generator=g, device=DEV) * 0.9).to(dtype)
    q = qkv[:, :H * D].view(T, H, D)
    k = qkv[:, H * D:(H + HKV) * D].view(T, HKV, D)
    v = qkv[:, (H + HKV) * D:].view(T, HKV, D)
    out = {}
    try:
        for hook in (4, 5):
            call("mq_attn_debug_waves", hook)
            if variant == "fp8":
                kv = qkv[:, H * D:].view(T, 2 * HKV, D)
                scale = ops.kv_scale_from_absmax(kv)
                cache = ops.kv_quant_fp8(kv, scale)
                out[hook] = ops.attn_prefill_fp8kv(q, cache, scale, causal=True)
                want = _ref(q, cache, scale, True)
                kw = dict(kv_cache=cache, kv_scale=scale)
            else:
                out[hook] = ops.attn_prefill(q, k, v, causal=True)
                s = torch.einsum("thd,hkd->htk", q.double(), k.double().repeat_interleave(H // HKV, 1).permute(1, 0, 2)) * D ** -0.5
                s = s.masked_fill(torch.ones(T, T, device=DEV, dtype=torch.bool).triu(1), float("-inf"))
                want = (torch.softmax(s, dim=-1) @ v.double().repeat_interleave(H // HKV, 1).permute(1, 0, 2)).permute(1, 0, 2).reshape(T, H * D)
                kw = dict(k=k, v=v)
            tol = 1.6e-2 if variant == "bf16" else 2.5e-3
            assert float((out[hook].double() - want).abs().max() / want.abs().max()) < tol, hook
            if hook == 4:
                s0 = float(out[4].float().abs().max()) / 127.0 * 0.8
                sel = (torch.arange(T, device=DEV) % 2).to(torch.uint8)
                lv, _ = ops.quantize_act_i8(out[4], s0, 0.5 * s0, row_sel=sel, tiled=True)
                fused = ops.attn_prefill_quant_i8(q, s0, 0.5 * s0, causal=True, row_sel=sel, tiled=True, **kw)
                assert torch.equal(fused.to_rows(), lv.to_rows())
    finally:
        call("mq_attn_debug_waves", 0)
    # the deepest half of the tiles takes the same path in both forms: identical bits there
    n = (T + 31) // 32
    deep_from = (n - n // 2) * 32
    assert torch.equal(out[4][deep_from:], out[5][deep_from:])


def test_random_shapes_through_every_launch_form():
    """Seeded sweep over token counts, head geometries, causal / not, the three K / V forms and the four launch forms (by shape, 2 waves,
    4 waves with paired shallow tiles, 4 waves unpaired): the row map of the grid -- heads in x, second round reversed, paired rows -- must
    visit every (head, query tile) exactly once whatever the counts are; each result against the float64 checker."""
    from mquant_amd import ops
    from mquant_amd._lib import call
    rng = np.random.default_rng(20260)
    try:
        for case in range(36):
            D = 80 if case % 6 == 5 else 128
            HKV = int(rng.choice([1, 2, 4, 8]))
            H = HKV * int(rng.choice([1, 2, 7]))
            T = int(rng.choice([1, 31, 32, 63, 64, 96, 127, 192, 250, 383, 640, 900, 1111]))
            causal = bool(rng.integers(0, 2))
            variant = "fp16" if D == 80 else str(rng.choice(["fp16", "bf16", "fp8"]))
            hook = int(rng.choice([0, 2, 4, 5]))
            dtype = torch.bfloat16 if variant == "bf16" else torch.float16
            g = torch.Generator(device=DEV).manual_seed(1000 + case)
            qkv = (torch.randn(T, (H + 2 * HKV) * D, generator=g, device=DEV) * 0.9).to(dtype)
            q = qkv[:, :H * D].view(T, H, D)
            k = qkv[:, H * D:(H + HKV) * D].view(T, HKV, D)
            v = qkv[:, (H + HKV) * D:].view(T, HKV, D)
            call("mq_attn_debug_waves", hook)
            if variant == "fp8":
                kv = qkv[:, H * D:].view(T, 2 * HKV, D)
                scale = ops.kv_scale_from_absmax(kv)
                cache = ops.kv_quant_fp8(kv, scale)
                got = ops.attn_prefill_fp8kv(q, cache, scale, causal=causal)
                want = _ref(q, cache, scale, causal)
            else:
                got = ops.attn_prefill(q, k, v, causal=causal)
                s = torch.einsum("thd,hkd->htk", q.double(), k.double().repeat_interleave(H // HKV, 1).permute(1, 0, 2)) * D ** -0.5
                if causal:
                    s = s.masked_fill(torch.ones(T, T, device=DEV, dtype=torch.bool).triu(1), float("-inf"))
                want = (torch.softmax(s, dim=-1) @ v.double().repeat_interleave(H // HKV, 1).permute(1, 0, 2)).permute(1, 0, 2).reshape(T, H * D)
            tol = 1.6e-2 if variant == "bf16" else 2.5e-3
            err = float((got.double() - want).abs().max() / want.abs().max())
            assert err < tol, (case, T, H, HKV, D, causal, variant, hook, err)
    finally:
        call("mq_attn_debug_waves", 0)


def test_size_independent_properties_at_full_size():
    """Properties that need no checker: (i) scaling V by a power of two scales the output exactly (the V scale rides in the
    output scale / the values themselves, the softmax does not see it); (ii) a query whose keys all carry the same V row
    returns that row (softmax weights sum to one) within half-precision rounding; (iii) T = 0 is accepted."""
    from mquant_amd import ops
    T, H, HKV, D = 768, 28, 4, 128
    g = torch.Generator(device=DEV).manual_seed(99)
    qkv = (torch.randn(T, (H + 2 * HKV) * D, generator=g, device=DEV) * 0.7).half()
    q = qkv[:, :H * D].view(T, H, D)
    k = qkv[:, H * D:(H + HKV) * D].view(T, HKV, D)
    v = qkv[:, (H + HKV) * D:].view(T, HKV, D)
    a = ops.attn_prefill(q, k, v, causal=True)
    kc = k.contiguous()                                   # k and v must share their token stride
    b = ops.attn_prefill(q, kc, (v.float() * 4.0).half(), causal=True)
    # exact up to fp16 subnormals: an output below 2^-14 is rounded on a coarser grid than four times itself
    d = (b.float() - a.float() * 4.0).abs()
    assert float(d.max()) <= 3e-7 and bool((d[a.float().abs() >= 2.0 ** -14] == 0).all())
    kv = qkv[:, H * D:].view(T, 2 * HKV, D)
    scale = ops.kv_scale_from_absmax(kv)
    cache = ops.kv_quant_fp8(kv, scale)
    scale4 = scale.clone()
    scale4[HKV:] *= 4.0                                   # the V heads' scales
    a8 = ops.attn_prefill_fp8kv(q, cache, scale, causal=True)
    b8 = ops.attn_prefill_fp8kv(q, cache, scale4, causal=True)
    assert float((b8.float() - a8.float() * 4.0).abs().max()) <= float(a8.float().abs().max()) * 4.0 * 2 ** -10
    row = torch.randn(1, HKV, D, generator=g, device=DEV).half()
    const_v = row.expand(T, HKV, D).contiguous()
    c = ops.attn_prefill(q, kc, const_v, causal=True).view(T, H, D).float()
    want = row[0].float().repeat_interleave(H // HKV, dim=0)[None].expand(T, H, D)
    assert float((c - want).abs().max()) <= float(want.abs().max()) * 3e-3
    e = torch.empty((0, H, D), device=DEV, dtype=torch.float16)
    assert ops.attn_prefill(e, e[:, :HKV], e[:, :HKV]).shape == (0, H * D)


def test_scales_that_would_silently_break_the_softmax_are_refused():
    """The running maximum is taken over raw scores and the scale applied afterwards: zero, negative and non-finite softmax /
    K-V scales must be errors, not wrong probabilities (advisor finding, round 3)."""
    from mquant_amd import ops
    from mquant_amd._lib import MQuantHipError
    T, H, HKV, D = 33, 4, 2, 128
    q = torch.randn(T, H, D, device=DEV, dtype=torch.float16)
    k = torch.randn(T, HKV, D, device=DEV, dtype=torch.float16)
    v = torch.randn(T, HKV, D, device=DEV, dtype=torch.float16)
    for bad in (0.0, -0.1, float("nan"), float("inf")):
        with pytest.raises(MQuantHipError):
            ops.attn_prefill(q, k, v, softmax_scale=bad)
    cache = torch.zeros(T, 2 * HKV, D, device=DEV, dtype=torch.float8_e4m3fn)
    for scales in (torch.tensor([1.0, 1.0, 0.0, 1.0]), torch.tensor([1.0, -1.0, 1.0, 1.0]), torch.tensor([1.0, 1.0, 1.0])):
        with pytest.raises(AssertionError):
            ops.attn_prefill_fp8kv(q, cache, scales.to(DEV))
