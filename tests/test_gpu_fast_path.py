"""The eager fast paths (round 6): ``ActQuantWrapper.forward`` keeps its "integer backend" decision while nothing was written
to the wrapper or its quantizers, and ``W4A8Linear.forward`` binds its two entry points once.  Both must give the bits of the
general path, and every way the reference's scripts change a wrapper's state must void the cached decision
(reference flag toggles: fake_quant/quant_utils.py:672-720; forward: :330-391)."""
import types

import pytest
import torch

pytestmark = pytest.mark.gpu


def _wrapped(K=512, N=256, had=False, msq=False, bias=True, seed=0):
    from fake_quant import hadamard_utils as hu, quant_utils as qu
    from fake_quant.gptq.rtn import rtn_module
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(seed)
    root = torch.nn.Module()
    root.lin = torch.nn.Linear(K, N, bias=bias, device=dev, dtype=torch.float16)
    root.lin.weight.data = (torch.randn((N, K), generator=g, device=dev) * 0.05).half()
    qu.add_actquant(root)
    wrap = root.lin
    if had:
        wrap.had_K, wrap.K = hu.get_hadK(K)
        wrap.online_full_had = True
    wrap.quantizer.configure(bits=8, sym=True, static=True, observer_type="minmax", msq=msq)
    rtn_module(root, "m", 4, True, False, [], {})
    args = types.SimpleNamespace(skip_names=[], no_sibling_fusion=True)
    x = torch.randn((48, K), generator=g, device=dev).half()
    mask = None
    if msq:                                            # rows 0..19 vision, 20.. text: both scale sets get calibrated
        mask = torch.zeros(48, dtype=torch.uint8, device=dev)
        mask[20:] = 1
    with qu.token_type_mask(mask):
        qu.model_open_calibrate(root, args)
        qu.model_open_last_calibrate(root, args)
        wrap(x)
        qu.model_close_calibrate(root, args)
        qu.model_quant(root, args)
    return qu, root, wrap, args, x


@pytest.mark.parametrize("had", [False, True])
def test_fast_forward_equals_general_forward(had):
    qu, root, wrap, args, x = _wrapped(had=had)
    y1 = wrap(x)                      # builds the engine, remembers the decision
    assert wrap.fast_path_active() and "integer" in wrap.backend()
    y2 = wrap(x)                      # cached decision + bound entry points
    eng = wrap._real
    a, x0 = eng.quantize(x)           # the general ops
    y3 = eng.gemm(a, x0, x.dtype)
    assert torch.equal(y1, y2) and torch.equal(y2, y3)
    x3 = x.reshape(4, 12, -1)
    assert torch.equal(wrap(x3), y3.reshape(4, 12, -1))
    # a strided input takes the general path of the engine and still agrees
    wide = torch.zeros((48, 2 * x.shape[1]), dtype=x.dtype, device=x.device)
    wide[:, ::2] = x
    assert torch.equal(wrap(wide[:, ::2]), y3)


def test_state_changes_void_the_cached_decision():
    qu, root, wrap, args, x = _wrapped()
    y_int = wrap(x)
    assert wrap.fast_path_active()
    qu.model_no_quant(root, args)                     # the reference's toggle: forward returns the unquantized Linear
    assert not wrap.fast_path_active()
    y_float = wrap(x)
    assert not wrap.fast_path_active()
    assert torch.equal(y_float, torch.nn.functional.linear(x, wrap.module.weight, wrap.module.bias))
    qu.model_quant(root, args)
    assert torch.equal(wrap(x), y_int) and wrap.fast_path_active()
    wrap.quantizer.quant = False                      # a direct flag write on the quantizer (version counter)
    assert torch.equal(wrap(x), y_float)
    wrap.quantizer.quant = True
    assert torch.equal(wrap(x), y_int)
    wrap.real_quant = False                           # attribute write on the wrapper
    y_sim = wrap(x)
    assert "simulated" in wrap.backend() and (y_sim.float() - y_int.float()).abs().max() <= 2e-2 * y_int.float().abs().max()
    wrap.real_quant = True
    assert torch.equal(wrap(x), y_int)
    wrap.out_quantizer.configure(bits=8, sym=True)    # an output quantizer: simulated path
    assert "simulated" in wrap.backend()
    wrap(x)
    assert not wrap.fast_path_active()


def test_engine_rebinds_when_the_scale_set_is_swapped():
    qu, root, wrap, args, x = _wrapped(msq=False)
    y = wrap(x)
    eng = wrap._real
    s = eng.s_x0
    eng.s_x0 = s * 2.0                                # what FullPrefill.calibrate does to its layers
    y2 = eng.forward(x)
    a, x0 = eng.quantize(x)
    assert torch.equal(y2, eng.gemm(a, x0, x.dtype)) and not torch.equal(y2, y)
    assert torch.equal(eng.forward(x), y2)            # bound again on the new scale
    eng.s_x0 = s
    assert torch.equal(eng.forward(x), y) and torch.equal(eng.forward(x), y)


def test_msq_mask_and_split_take_the_fast_path_with_the_same_bits():
    from fake_quant import quant_utils as qu
    qu2, root, wrap, args, x = _wrapped(msq=True, seed=3)
    mask = torch.zeros(48, dtype=torch.uint8, device=x.device)
    mask[20:] = 1
    qu.set_token_type_mask(mask)
    try:
        y1, y2 = wrap(x), wrap(x)
        eng = wrap._real
        a, x0 = eng.quantize(x, mask)
        assert torch.equal(y1, y2) and torch.equal(y2, eng.gemm(a, x0, x.dtype, mask))
    finally:
        qu.set_token_type_mask(None)


def test_debug_workspace_catches_a_stale_activation_handle(monkeypatch):
    from mquant_amd import engine
    from mquant_amd._lib import MQuantHipError
    qu, root, wrap, args, x = _wrapped()
    wrap(x)
    eng = wrap._real
    monkeypatch.setattr(engine, "DEBUG_WORKSPACE", True)
    a1, _ = eng.quantize(x)
    eng.gemm(a1, None, x.dtype)                       # fresh: fine
    a2, _ = eng.quantize(x[:16])                      # same K_pad: the buffer is handed out again
    with pytest.raises(MQuantHipError):
        eng.gemm(a1, None, x.dtype)
    eng.gemm(a2, None, x.dtype)


def test_act_order_engines_refuse_producer_side_entry_points():
    from mquant_amd.engine import W4A8Linear
    dev = torch.device("cuda:0")
    levels = torch.randint(-8, 8, (64, 256), dtype=torch.int8, device=dev)
    tbl = torch.full((2, 64), 0.01, dtype=torch.float32, device=dev)
    eng = W4A8Linear(levels, torch.ones(64, device=dev), 4, None, 0.05, w_groups=(tbl, 128),
                     col_perm=torch.randperm(256, device=dev))
    for fn in (lambda: eng.act_buffer(16), lambda: eng.quantize_rmsn(torch.zeros(16, 256, device=dev, dtype=torch.float16), 256, 1e-6)):
        with pytest.raises(AssertionError):
            fn()


def test_engine_rebinds_when_one_of_its_tensors_is_replaced():
    qu, root, wrap, args, x = _wrapped(bias=True)
    y = wrap(x)
    eng = wrap._real
    eng.forward(x)                                     # bound
    new_bias = (eng.bias + 1.0).contiguous()
    eng.bias = new_bias                                # a different tensor object: the bound address is stale
    y2 = eng.forward(x)
    a, x0 = eng.quantize(x)
    assert torch.equal(y2, eng.gemm(a, x0, x.dtype)) and not torch.equal(y2, y)
