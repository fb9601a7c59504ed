"""The objects bench.py times -- ``workload.WrapperPrefill`` (the prefill built through the drop-in API, the
default), ``workload.Prefill`` (the engines assembled directly) and ``full_prefill.FullPrefill`` -- checked
against the oracle composition of every layer and against each other: fused q|k|v / gate|up GEMMs equal the
per-Linear ones slice by slice, the wrapper-built prefill equals the directly assembled one bit for bit."""
import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
torch.set_grad_enabled(False)


def _np(t):
    return t.detach().float().cpu().numpy()


def oracle_layer(spec, li, x, had_table, rows=None, seed=1234, w_bits=4):
    """Reference composition of one Linear instance of the workload on the CPU oracle: RTN levels of the
    synthetic weight, [pad + Hadamard], min/max scales (per token type with MSQ), static int8 levels,
    exact int32 GEMM, dequantising epilogue (+ bias, + the fp32 split column), rounded to fp16.
    ``rows`` restricts the GEMM (not the calibration) to a subset of the rows."""
    from mquant_amd import workload
    w, bias = workload.synth_weight(spec, li, seed, DEV, torch.float16)
    w = _np(w)
    w0 = None
    if spec.split:
        s_w, lv = oracle.wquant_sym(w[:, 1:], bits=w_bits)
        lv = np.concatenate([np.zeros((spec.n, 1), np.int8), lv], axis=1)
        w0 = w[:, 0]
    else:
        s_w, lv = oracle.wquant_sym(w, bits=w_bits)
    xr = _np(x)
    if spec.had_K:
        hk = had_table["mats"][spec.had_K] if spec.had_K > 1 else None
        xr = oracle.hadamard(xr, spec.k, spec.had_K, hk, mid_round=1, out_round=1)
    cb = 1 if spec.split else 0

    def scale(part):
        # observer/minmax.py:30-46 on fp16 activations: every intermediate is an fp16 tensor
        r16 = lambda v: oracle.round_to(np.float32(v).reshape(1), 1)[0]
        mn, mx = min(float(part[:, cb:].min()), 0.0), max(float(part[:, cb:].max()), 0.0)
        s = max(abs(r16(np.float32(mn) / np.float32(-128.0))), abs(r16(np.float32(mx) / np.float32(127.0))))
        return np.float32(max(s, r16(np.finfo(np.float32).eps)))
    sel, s1 = None, None
    if spec.msq:
        sel = _np(workload.vision_text_mask(spec.M, DEV)).astype(np.uint8)
        nv = int((sel == 0).sum())
        s0, s1 = scale(xr[:nv]), scale(xr[nv:])
    else:
        s0 = scale(xr)
    if rows is not None:
        xr = xr[rows]
        sel = None if sel is None else sel[rows]
    q = oracle.quant_static(xr, s0, scale1=s1, row_sel=sel)
    x0 = None
    if spec.split:
        x0 = xr[:, 0].copy()
        q[:, 0] = 0
    acc = oracle.gemm_i32(q, lv)
    y = oracle.epilogue(acc, s0, s_w, bias=None if bias is None else _np(bias), sx1=s1, row_sel=sel, x0=x0, w0=w0)
    return oracle.round_to(y, 1)


def expected_outputs(specs, had_table, rows=None):
    """{(spec name, instance): oracle output} in the workload's instance numbering."""
    from mquant_amd import workload
    inputs = workload.synth_inputs(specs, DEV, torch.float16)
    out, li = {}, 0
    for spec in specs:
        for c in range(spec.count):
            out[(spec.name, c)] = oracle_layer(spec, li, inputs[(spec.M, spec.k_in)], had_table, rows)
            li += 1
    return out


def prefill_outputs(pf, specs):
    """{(spec name, instance): GPU output} of a ``workload.Prefill`` (fused groups sliced back)."""
    got = {}
    by_group = {}
    for sp in specs:
        if sp.group:
            by_group.setdefault(sp.group, []).append(sp)
    for L in pf.layers:
        a, x0 = L.lin.quantize(L.x, L.row_sel)
        y = L.lin.gemm(a, x0, pf.dtype, L.row_sel, None)
        if hasattr(L, "order_name"):              # fused group: members in spec order
            lo = 0
            for sp in by_group[L.spec.group]:
                got[(sp.name, L.idx)] = _np(y[:, lo:lo + sp.n])
                lo += sp.n
            assert lo == y.shape[1]
        else:
            got[(L.spec.name, L.idx)] = _np(y)
    return got


def wrapper_outputs(wp):
    from mquant_amd import workload
    ys = wp.outputs()
    keys = workload.execution_order(wp.specs)
    assert len(keys) == len(ys)
    return {k: _np(y) for k, y in zip(keys, ys)}


@pytest.mark.parametrize("share", [True, False])
def test_tiny_prefill_every_layer_equals_the_oracle_composition(share, had_table):
    from mquant_amd import workload
    specs = workload.tiny_specs()
    want = expected_outputs(specs, had_table)
    pf = workload.Prefill(specs, device=DEV, share_groups=share)
    n_lin = sum(sp.count for sp in specs)
    assert pf.gemm_launches() == (n_lin - 2 * 2 - 2 * 1 if share else n_lin)      # q|k|v and gate|up fused per block
    got = prefill_outputs(pf, specs)
    assert set(got) == set(want)
    for key in want:
        np.testing.assert_array_equal(got[key], want[key], err_msg=str(key))
    # the timed method itself: one pass, the last Linear's output comes back
    y = pf.step()
    last = workload.execution_order(specs)[-1]
    np.testing.assert_array_equal(_np(y), want[last])


@pytest.mark.parametrize("fuse", [True, False])
def test_tiny_wrapper_prefill_equals_the_oracle_and_the_direct_engines(fuse, had_table):
    """The default bench object: module tree -> add_actquant -> RTN -> calibration protocol -> model_quant ->
    ActQuantWrapper.forward.  Same bits as the oracle, same launch count as the directly assembled engines."""
    from fake_quant import quant_utils as qu
    from mquant_amd import workload
    specs = workload.tiny_specs()
    want = expected_outputs(specs, had_table)
    wp = workload.WrapperPrefill(specs, device=DEV, fuse_siblings=fuse)
    got = wrapper_outputs(wp)
    for key in want:
        np.testing.assert_array_equal(got[key], want[key], err_msg=str(key))
    pf = workload.Prefill(specs, device=DEV, share_groups=fuse)
    assert wp.gemm_launches() == pf.gemm_launches()
    assert wp.gemm_ops() == pf.gemm_ops() and wp.gemm_bytes() == pf.gemm_bytes() and wp.quant_bytes() == pf.quant_bytes()
    groups = {id(w._group) for w, _, _ in wp.calls if w.__dict__.get("_group") is not None}
    assert len(groups) == (4 if fuse else 0)                    # 2 blocks x (q|k|v, gate|up)
    np.testing.assert_array_equal(_np(wp.step()), want[workload.execution_order(specs)[-1]])
    # pickles must not drag the group or the engines along (reference checkpoints pickle whole modules)
    import copy
    clone = copy.deepcopy(wp.calls[-1][0])
    assert clone.__dict__["_group"] is None and clone._real is None
    assert isinstance(clone, qu.ActQuantWrapper)


def one_llm_layer_specs():
    from mquant_amd import workload
    return [sp for sp in workload._qwen2vl_7b_specs(True, 1, 1) if sp.name.startswith("llm.")]


def test_real_size_llm_layer_fused_unfused_and_wrappers_agree_bit_for_bit(had_table):
    """One Qwen2-VL-7B decoder layer at the benchmark's sizes (M = 768): the fused q|k|v and gate|up GEMMs equal
    the per-Linear ones slice by slice, the wrapper-built layer equals both, and sampled rows equal the oracle."""
    from mquant_amd import workload
    specs = one_llm_layer_specs()
    fused = prefill_outputs(workload.Prefill(specs, device=DEV, share_groups=True), specs)
    plain = prefill_outputs(workload.Prefill(specs, device=DEV, share_groups=False), specs)
    wrapped = wrapper_outputs(workload.WrapperPrefill(specs, device=DEV, fuse_siblings=True))
    assert set(fused) == set(plain) == set(wrapped) and len(fused) == 7
    for key in fused:
        np.testing.assert_array_equal(fused[key], plain[key], err_msg=f"fused vs per-Linear {key}")
        np.testing.assert_array_equal(wrapped[key], fused[key], err_msg=f"wrappers vs engines {key}")
    rows = np.array([0, 1, 255, 256, 257, 511, 766, 767])       # both token types, both ends
    want = expected_outputs(specs, had_table, rows=rows)
    for key in want:
        np.testing.assert_array_equal(fused[key][rows], want[key], err_msg=f"oracle {key}")


def test_full_prefill_fused_glue_equals_unfused_glue():
    """The chained prefill of the TTFT report (2 ViT blocks + 2 decoder layers of the real widths): with the
    norm -> quantize, activation -> Hadamard -> quantize, residual-epilogue and RoPE fusions the logits are the
    ones the same dataflow gives with those steps as separate torch kernels.  The fused RMS norm sums in a
    different order than torch's (<= 1 ulp before quantization), so a few int8 levels may differ:
    the tolerance is 2 % of the logit range, and finite, non-trivial logits are required."""
    from mquant_amd import workload
    from mquant_amd.full_prefill import FullPrefill
    specs = workload._qwen2vl_7b_specs(True, 2, 2)
    pf = workload.Prefill(specs, device=DEV, share_groups=True)
    outs = []
    for fused in (False, True):
        fp = FullPrefill(pf, fused_glue=fused, attn_kernel=False)       # same SDPA on both sides: this test is about the glue kernels
        fp.calibrate()
        outs.append(fp.step().float().clone())
        fp.restore_hot_path_scales()
    a, b = outs
    assert torch.isfinite(a).all() and torch.isfinite(b).all() and float(a.abs().max()) > 0
    span = float(a.max() - a.min())
    assert float((a - b).abs().max()) <= 0.02 * span, (float((a - b).abs().max()), span)
    cos = torch.nn.functional.cosine_similarity(a.flatten(), b.flatten(), dim=0)
    assert float(cos) > 0.999


def test_full_prefill_with_this_repositorys_attention_kernel():
    """The attention of the chained prefill through mq_attn_prefill -- vision tower (head_dim 80, non-causal) and decoder
    (128, causal), q / k / v read in place from the fused GEMM outputs -- against torch SDPA in the same prefill.  The first
    vision block sees identical inputs in both runs, so its attention output is held to half-precision rounding; the first
    decoder layer already sees activations that went through a vision block, the merger and static int8 quantizers with
    that rounding difference in them (a few levels flip), so it is held to 3 %; the logits of the random stack stay
    correlated."""
    from mquant_amd import workload
    from mquant_amd.full_prefill import FullPrefill
    specs = workload._qwen2vl_7b_specs(True, 1, 2)
    pf = workload.Prefill(specs, device=DEV, share_groups=True)
    outs = []
    for own in (False, True):
        fp = FullPrefill(pf, fused_glue=True, attn_kernel=own)
        assert fp.attn_kernel == own and fp.vis_attn_kernel == own
        fp.calibrate()
        outs.append((fp.step().float().clone(), fp.attn_first.float().clone(), fp.vis_attn_first.float().clone()))
        fp.restore_hot_path_scales()
    (la, aa, va), (lb, ab, vb) = outs
    assert torch.isfinite(lb).all()
    assert float((va - vb).abs().max() / va.abs().max()) < 5e-3
    assert float(torch.nn.functional.cosine_similarity(va.flatten(), vb.flatten(), dim=0)) > 0.99999
    assert float((aa - ab).norm() / aa.norm()) < 3e-2
    assert float(torch.nn.functional.cosine_similarity(aa.flatten(), ab.flatten(), dim=0)) > 0.9995
    assert float(torch.nn.functional.cosine_similarity(la.flatten(), lb.flatten(), dim=0)) > 0.99


def test_attention_with_the_fused_output_quantizer_changes_no_bit_of_the_prefill():
    """FullPrefill.attn_quant: from the second block / layer on the attention kernels hand o_proj / proj their int8
    activations directly (mq_attn_prefill_quant_i8).  Same 16-bit rounding, same quantizer arithmetic: the logits must be
    IDENTICAL to the run that stores 16-bit attention outputs and quantizes them in a separate launch."""
    from mquant_amd import workload
    from mquant_amd.full_prefill import FullPrefill
    specs = workload._qwen2vl_7b_specs(True, 3, 3)
    pf = workload.Prefill(specs, device=DEV, share_groups=True)
    outs = []
    for fused_q in (False, True):
        for kv8 in (False, True):
            fp = FullPrefill(pf, fused_glue=True, kv_fp8=kv8, attn_fp8=kv8)
            fp.attn_quant = fused_q
            fp.calibrate()
            outs.append(fp.step().float().clone())
            fp.restore_hot_path_scales()
    assert torch.isfinite(outs[0]).all() and float(outs[0].abs().max()) > 0
    assert torch.equal(outs[0], outs[2])      # 16-bit K / V
    assert torch.equal(outs[1], outs[3])      # fp8 cache, attention over the e4m3 bytes
