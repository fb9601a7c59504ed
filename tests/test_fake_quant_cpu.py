"""Host-side mirror of the reference interface (the ``fake_quant`` package) on CPU tensors,
against goldens produced by the reference itself (tools/gen_golden.py).

Covers the host logic only: registries, observers, quantizer math in torch, module surgery and
the calibration protocol.  The quantized forward itself is GPU-only and must refuse CPU tensors.
"""
import os
import pickle

import numpy as np
import pytest
import torch

from golden_inputs import make_w, make_x

from fake_quant import hadamard_utils as hu
from fake_quant import module_util, quant_utils as qu, utils
from fake_quant.bit_type import BIT_TYPE_DICT, BitType
from fake_quant.observer import build_observer
from fake_quant.quantizer import build_quantizer

torch.set_grad_enabled(False)


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"), allow_pickle=False)


class Args:
    skip_names = []


# ------------------------------------------------------------------------------- bit types
def test_bit_type_registry():
    assert sorted(BIT_TYPE_DICT) == sorted(["uint4", "int8", "uint8", "int16", "int20", "int18"])
    assert (BIT_TYPE_DICT["int8"].lower_bound, BIT_TYPE_DICT["int8"].upper_bound) == (-128, 127)
    assert (BIT_TYPE_DICT["uint4"].lower_bound, BIT_TYPE_DICT["uint4"].upper_bound) == (0, 15)
    assert BitType(4, True).name == "int4" and BitType(4, True).range == 16
    with pytest.raises(KeyError):           # static 4-bit activations are impossible upstream too
        qu.ActQuantizer().configure(bits=4, sym=True, static=True)


# ------------------------------------------------------------------------------- Hadamard utils
def test_hadk_dispatch_and_padding(golden_dir):
    g = load(golden_dir, "hadk_table")
    for n, K in zip(g["n"].tolist(), g["K"].tolist()):
        h, k = hu.get_hadK(n)
        assert k == K and (h is None) == (K == 1)
    for a, b in zip(g["pad_in"].tolist(), g["pad_out"].tolist()):
        assert hu.auto_pad_size(a) == b
    for K in (12, 20, 28, 36, 40, 52, 60, 108, 140, 156, 172):
        ref = (np.unpackbits(g[f"had{K}"])[: K * K].reshape(K, K).astype(np.float32) * 2 - 1)
        np.testing.assert_array_equal(getattr(hu, f"get_had{K}")().numpy(), ref)
    ht, _ = hu.get_hadK(5120, transpose=True)
    np.testing.assert_array_equal(ht.numpy(), hu.get_had40().numpy().T)
    assert hu.is_pow2(64) and not hu.is_pow2(0) and not hu.is_pow2(96)


def test_sign_words_layout():
    h = hu._had_np(156)
    w = hu.sign_words(h)
    assert w.shape == (156, 5) and w.dtype == np.uint32
    for j, k in [(0, 0), (17, 77), (155, 155), (3, 128), (99, 31)]:
        assert ((int(w[j, k // 32]) >> (k % 32)) & 1) == int(h[j, k] > 0)
    assert (w[:, 4] >> (156 - 128)).max() == 0    # unused high bits are zero


@pytest.mark.parametrize("n", [64, 1280, 5120, 11008, 19968])
def test_matmul_hadU_matches_reference(golden_dir, n):
    g = load(golden_dir, "hadamard_fwd")
    x = torch.from_numpy(make_x(100 + n, (2 if n <= 5120 else 1, n)))
    np.testing.assert_array_equal(hu.matmul_hadU(x).numpy(), g[f"hadU_{n}"])


def test_random_hadamard_matrix_uses_global_cpu_rng(golden_dir):
    g = load(golden_dir, "random_hadamard")
    utils.seed_everything(42)
    Q = hu.random_hadamard_matrix(64, torch.device("cpu"))
    assert Q.dtype == torch.float64
    np.testing.assert_array_equal(Q.numpy(), g["Q64"])
    utils.seed_everything(42)
    Q = hu.random_hadamard_matrix(3584, torch.device("cpu"))
    np.testing.assert_array_equal(Q[:4].numpy(), g["Q3584_rows"])
    assert torch.allclose(Q @ Q.T, torch.eye(3584, dtype=torch.float64), atol=1e-9)


def test_offline_hadamard_on_linear(golden_dir):
    g = load(golden_dir, "offline_hadamard")
    for n in (5120, 1280):
        lin = torch.nn.Linear(n, 24, bias=True)
        lin.weight.data = torch.from_numpy(make_w(700 + n, (24, n)))
        hu.apply_exact_had_to_linear(lin, had_dim=-1, output=False)
        np.testing.assert_allclose(lin.weight.data.numpy(), g[f"W_in_{n}"], rtol=0, atol=2e-7)
        lin2 = torch.nn.Linear(24, n, bias=True)
        lin2.weight.data = torch.from_numpy(make_w(710 + n, (n, 24)))
        lin2.bias.data = torch.from_numpy(make_w(711 + n, (n,), std=0.1))
        hu.apply_exact_had_to_linear(lin2, had_dim=-1, output=True)
        np.testing.assert_allclose(lin2.weight.data.numpy(), g[f"W_out_{n}"], rtol=0, atol=2e-7)
        np.testing.assert_allclose(lin2.bias.data.numpy(), g[f"b_out_{n}"], rtol=0, atol=2e-7)
    lin = torch.nn.Linear(512, 24, bias=False)
    lin.weight.data = torch.from_numpy(make_w(720, (24, 512)))
    hu.apply_exact_had_to_linear(lin, had_dim=128, output=False)
    np.testing.assert_allclose(lin.weight.data.numpy(), g["W_headin_128"], rtol=0, atol=2e-7)
    lin = torch.nn.Linear(24, 512, bias=True)
    lin.weight.data = torch.from_numpy(make_w(721, (512, 24)))
    lin.bias.data = torch.from_numpy(make_w(722, (512,), std=0.1))
    hu.apply_exact_had_to_linear(lin, had_dim=128, output=True)
    np.testing.assert_allclose(lin.weight.data.numpy(), g["W_headout_128"], rtol=0, atol=2e-7)
    np.testing.assert_allclose(lin.bias.data.numpy(), g["b_headout_128"], rtol=0, atol=2e-7)


def test_online_offline_hadamard_invariance():
    """online-Had(x) . Had(W)^T == x . W^T (the invariant behind rotate + online_full_had)."""
    n = 5120
    x = torch.from_numpy(make_x(1, (8, n)))
    lin = torch.nn.Linear(n, 16, bias=False)
    lin.weight.data = torch.from_numpy(make_w(2, (16, n)))
    y0 = lin(x)
    hu.apply_exact_had_to_linear(lin, had_dim=-1, output=False)
    hadK, K = hu.get_hadK(n)
    y1 = lin(hu._hadamard_torch(x, hadK, K))
    assert (y0 - y1).abs().max() < 5e-6


def test_online_hadamard_refuses_cpu_tensors():
    from mquant_amd._lib import MQuantHipError
    with pytest.raises(MQuantHipError):
        hu.matmul_hadU_cuda(torch.zeros(2, 64), None, 1)


# ------------------------------------------------------------------------------- observers
def test_minmax_observer_sequence(golden_dir):
    g = load(golden_dir, "observers")
    bt = BIT_TYPE_DICT["int8"]
    for mode in ("layer_wise", "channel_wise"):
        ob = build_observer("minmax", "activation", bt, mode)
        for i in range(3):
            ob.update(torch.from_numpy(g[f"batch{i}"]))
            np.testing.assert_array_equal(np.asarray(ob.max_val.numpy(), np.float32), g[f"minmax_{mode}_max{i}"])
            np.testing.assert_array_equal(np.asarray(ob.min_val.numpy(), np.float32), g[f"minmax_{mode}_min{i}"])
        s, z = ob.get_quantization_params()
        np.testing.assert_array_equal(s.numpy(), g[f"minmax_{mode}_scale"])
        np.testing.assert_array_equal(z.numpy(), g[f"minmax_{mode}_zp"])
        assert z.dtype == torch.int64
    for tag, i in (("pos", 1), ("neg", 2)):   # zero-inclusion rule on the first batch
        ob = build_observer("minmax", "activation", bt, "layer_wise")
        ob.update(torch.from_numpy(g[f"batch{i}"]))
        s, _ = ob.get_quantization_params()
        np.testing.assert_array_equal(
            np.array([ob.min_val.item(), ob.max_val.item(), s.item()], np.float32), g[f"minmax_first_{tag}"])
    ob = build_observer("minmax", "activation", bt, "channel_wise")
    ob.update(torch.from_numpy(g["x4"]))      # 4-D: NCHW -> channel = dim 1
    np.testing.assert_array_equal(ob.max_val.numpy(), g["minmax_4d_max"])
    np.testing.assert_array_equal(ob.min_val.numpy(), g["minmax_4d_min"])
    ob = build_observer("minmax", "activation", BIT_TYPE_DICT["uint8"], "channel_wise")
    ob.update(torch.from_numpy(g["batch0"]))
    s, z = ob.get_quantization_params()
    np.testing.assert_array_equal(s.numpy(), g["minmax_uint8_scale"])
    np.testing.assert_array_equal(z.numpy(), g["minmax_uint8_zp"])


@pytest.mark.parametrize("name", ["ema", "omse", "percentile", "ptf"])
def test_other_observers_registry_parity(golden_dir, name):
    g = load(golden_dir, "observers")
    mode = "layer_wise" if name == "percentile" else "channel_wise"
    bt = BIT_TYPE_DICT["uint8"] if name in ("omse", "ptf") else BIT_TYPE_DICT["int8"]
    ob = build_observer(name, "activation", bt, mode)
    for i in range(3):
        ob.update(torch.from_numpy(g[f"batch{i}"]))
    if name in ("omse", "ptf"):
        s, z = ob.get_quantization_params(torch.from_numpy(g["batch0"]))
    else:
        s, z = ob.get_quantization_params()
    np.testing.assert_allclose(np.asarray(ob.max_val.numpy(), np.float32), g[f"{name}_max"], rtol=1e-6, atol=0)
    np.testing.assert_allclose(np.asarray(ob.min_val.numpy(), np.float32), g[f"{name}_min"], rtol=1e-6, atol=0)
    np.testing.assert_allclose(np.asarray(s.numpy(), np.float32), g[f"{name}_scale"], rtol=1e-6, atol=0)
    np.testing.assert_array_equal(np.asarray(z.numpy()), g[f"{name}_zp"])


# ------------------------------------------------------------------------------- quantizers
@pytest.mark.parametrize("tag", ["2d", "3d", "4d", "5d"])
def test_uniform_quantizer_all_ranks(golden_dir, tag):
    g = load(golden_dir, "uniform_quantizer")
    bt = BIT_TYPE_DICT["int8"]
    x = torch.from_numpy(g[f"x_{tag}"])
    for mode in ("layer_wise", "channel_wise"):
        qz = build_quantizer("uniform", bt, build_observer("minmax", "activation", bt, mode), "activation")
        if mode == "layer_wise":
            qz.scale, qz.zero_point = torch.tensor(float(g["scale_lw"])), torch.tensor(0)
        else:
            qz.scale, qz.zero_point = torch.from_numpy(g["scale_cw"]), torch.zeros(40, dtype=torch.int64)
        np.testing.assert_array_equal(qz.quant(x.float()).numpy().astype(np.int8), g[f"q_{tag}_{mode}"])
        np.testing.assert_array_equal(qz(x).numpy(), g[f"dq_{tag}_{mode}"])
        np.testing.assert_array_equal(qz(x.half()).float().numpy(), g[f"dq16_{tag}_{mode}"])


def test_act_quantizer_dynamic_modes(golden_dir):
    g = load(golden_dir, "act_dynamic")
    cases = {"tok_sym": dict(bits=8, sym=True), "tok_asym": dict(bits=8, sym=False),
             "tok_sym4": dict(bits=4, sym=True),
             "tensor_sym": dict(bits=8, sym=True, act_per_tensor=True),
             "tensor_asym": dict(bits=8, sym=False, act_per_tensor=True),
             "group_sym": dict(bits=8, sym=True, groupsize=128),
             "group_asym": dict(bits=8, sym=False, groupsize=128),
             "clip_sym": dict(bits=8, sym=True, clip_ratio=0.9)}
    for tag, kw in cases.items():
        aq = qu.ActQuantizer()
        aq.configure(**kw)
        x = torch.from_numpy(g["x"].copy())
        aq.find_params(x)
        np.testing.assert_array_equal(aq(x).numpy(), g[f"y_{tag}"], err_msg=tag)
        sc = aq.scale
        np.testing.assert_array_equal(np.asarray(sc.numpy() if torch.is_tensor(sc) else sc, np.float32),
                                      g[f"scale_{tag}"], err_msg=tag)
    aq = qu.ActQuantizer()
    x = torch.randn(3, 5)
    assert aq(x) is x                               # bits == 16: identity
    with pytest.raises(AssertionError):
        aq.configure(bits=8, clip_ratio=1.5)


def test_weight_quantizer_all_modes(golden_dir):
    g = load(golden_dir, "weight_quantizer")
    cases = {"w4_sym": dict(bits=4, perchannel=True, sym=True, mse=False),
             "w4_sym_mse": dict(bits=4, perchannel=True, sym=True, mse=True),
             "w8_sym": dict(bits=8, perchannel=True, sym=True, mse=False),
             "w8_sym_mse": dict(bits=8, perchannel=True, sym=True, mse=True),
             "w4_asym": dict(bits=4, perchannel=True, sym=False, mse=False),
             "w4_asym_mse": dict(bits=4, perchannel=True, sym=False, mse=True),
             "w4_tensor": dict(bits=4, perchannel=False, sym=True, mse=False)}
    for tag, kw in cases.items():
        wq = qu.WeightQuantizer()
        wq.configure(**kw)
        W = torch.from_numpy(g["W"].copy())
        wq.find_params(W)
        assert wq.ready() and wq.enabled()
        np.testing.assert_array_equal(wq.scale.numpy(), g[f"scale_{tag}"], err_msg=tag)
        np.testing.assert_array_equal(wq.zero.numpy(), g[f"zero_{tag}"], err_msg=tag)
        np.testing.assert_array_equal(wq.quantize(W).numpy(), g[f"wq_{tag}"], err_msg=tag)


def test_pack_unpack_i4(golden_dir):
    g = load(golden_dir, "pack_i4")
    q = torch.from_numpy(g["q"])
    packed = qu.pack_i4(q)
    np.testing.assert_array_equal(packed.numpy(), g["packed"])
    un = qu.unpack_i4(packed)
    assert un.dtype == torch.int32
    np.testing.assert_array_equal(un.numpy(), g["unpacked"])
    with pytest.raises(AssertionError):
        qu.pack_i4(torch.full((2, 4), 9, dtype=torch.int8))
    mn, mx = qu.get_minq_maxq(4, True)
    assert int(mn) == -8 and int(mx) == 7 and qu.get_minq_maxq(8, False)[0] == 0


# ------------------------------------------------------------------------------- module surgery
class Block(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.q_proj = torch.nn.Linear(8, 8)
        self.seq = torch.nn.Sequential(torch.nn.Linear(8, 8), torch.nn.GELU(), torch.nn.Linear(8, 8))
        self.lst = torch.nn.ModuleList([torch.nn.Linear(8, 8), torch.nn.LayerNorm(8)])

    def forward(self, x):
        return self.lst[0](self.seq(self.q_proj(x)))


class SubLinear(torch.nn.Linear):
    pass


def test_add_actquant_and_find_qlayers():
    m = torch.nn.Sequential(Block(), Block())
    m[1].odd = SubLinear(8, 8)                      # subclasses are NOT wrapped (exact type match)
    qu.add_actquant(m)
    wrapped = qu.find_qlayers(m, layers=[qu.ActQuantWrapper])
    assert sorted(wrapped) == sorted([f"{i}.{n}" for i in (0, 1) for n in ("q_proj", "seq.0", "seq.2", "lst.0")])
    assert type(m[1].odd) is SubLinear
    qu.add_actquant(m)                              # idempotent
    assert len(qu.find_qlayers(m, layers=[qu.ActQuantWrapper])) == 8
    lin = qu.find_qlayers(m, layers=[torch.nn.Linear])
    assert "0.q_proj.module" in lin and "1.odd" not in lin
    w = m[0].q_proj
    assert w.weight is w.module.weight and w.bias is w.module.bias and w.had_K is None and w.K == 1
    x = torch.randn(3, 8)
    torch.testing.assert_close(m(x), m(x))          # unconfigured wrappers are transparent


def test_split_weights_views_and_names():
    w = qu.ActQuantWrapper(torch.nn.Linear(16, 4, bias=True))
    w.split = True
    w.split_weights()
    assert w.L1.weight.shape == (4, 1) and w.L2.weight.shape == (4, 15)
    assert w.L2.weight.data.data_ptr() == w.module.weight.data[:, 1:].data_ptr()
    assert sorted(n for n, _ in w.named_children()) == ["L1", "L2", "module", "out_quantizer", "quantizer"]
    x = torch.randn(5, 16)
    torch.testing.assert_close(w(x.clone()), w.module(x), rtol=1e-5, atol=1e-5)


def test_pad_hook_is_folded_into_the_wrapper():
    import functools
    w = qu.ActQuantWrapper(torch.nn.Linear(24, 4, bias=False))
    h = w.register_forward_pre_hook(functools.partial(utils.revise_down_input, new_size=24))
    assert w.pad_to == 24 and len(w._forward_pre_hooks) == 0
    h.remove()
    x = torch.randn(3, 20)
    torch.testing.assert_close(w(x), w.module(torch.nn.functional.pad(x, (0, 4))))
    other = w.register_forward_pre_hook(lambda m, i: None)   # unrelated hooks still register
    assert len(w._forward_pre_hooks) == 1
    other.remove()
    assert utils.revise_down_input(None, (x,), 24)[0].shape == (3, 24)


def test_wrapper_pickles_under_the_reference_import_path():
    w = qu.ActQuantWrapper(torch.nn.Linear(8, 4))
    assert type(w).__module__ == "fake_quant.quant_utils"
    w2 = pickle.loads(pickle.dumps(w))
    assert isinstance(w2, qu.ActQuantWrapper) and w2._real is None


def test_rmsn_and_replace_modules():
    n = module_util.RMSN(16)
    x = torch.randn(4, 16)
    ref = x * torch.rsqrt(x.pow(2).sum(-1, keepdim=True) / 16 + 1e-5)
    torch.testing.assert_close(n(x), ref)
    assert n(x.half()).dtype == torch.float16
    m = torch.nn.Sequential(torch.nn.LayerNorm(16), torch.nn.Sequential(torch.nn.LayerNorm(16)))
    module_util.replace_modules(m, torch.nn.LayerNorm, lambda old: module_util.RMSN(16), replace_layers=False)
    assert isinstance(m[0], module_util.RMSN) and isinstance(m[1][0], module_util.RMSN)


# ------------------------------------------------------------------------------- calibration
@pytest.mark.parametrize("case", ["plain_3584", "plain_w8"])
def test_calibration_protocol_trace_on_cpu(golden_dir, case):
    """open -> 3 forwards (last one with last_calibrate) -> close -> quant, flags and scale
    identical to the reference; activations flow unquantized while calibrating."""
    g = load(golden_dir, "wrapper_" + case)
    K_in, K_pad, N, M, seed, had, split, w_bits, w_mse, bias = [int(v) for v in g["meta"]]
    lin = torch.nn.Linear(K_pad, N, bias=bool(bias))
    lin.weight.data = torch.from_numpy(make_w(seed, (N, K_pad)))
    if bias:
        lin.bias.data = torch.from_numpy(make_w(seed + 1, (N,), std=0.1))
    wrap = qu.ActQuantWrapper(lin)
    wq = qu.WeightQuantizer()
    wq.configure(w_bits, perchannel=True, sym=True, mse=bool(w_mse))
    wq.find_params(lin.weight.data)
    lin.weight.data = wq.quantize(lin.weight.data)
    np.testing.assert_array_equal(wq.scale.numpy().reshape(-1), g["s_w"])
    qu.attach_weight_quantizer(wrap, "module", wq)
    wrap.quantizer.configure(bits=8, sym=True, static=True, observer_type="minmax")
    args = Args()
    trace = []
    qu.model_open_calibrate(wrap, args)
    for i in range(3):
        if i == 2:
            qu.model_open_last_calibrate(wrap, args)
        y = wrap(torch.from_numpy(make_x(seed + 10 + i, (M, K_in))))
        trace.append([wrap.quantizer.calibrate, wrap.quantizer.last_calibrate, wrap.quantizer.quant])
        if i == 0:
            np.testing.assert_allclose(y.numpy(), g["y_calib0"], rtol=0, atol=1e-5)
    qu.model_close_calibrate(wrap, args)
    qu.model_quant(wrap, args)
    trace.append([wrap.quantizer.calibrate, wrap.quantizer.last_calibrate, wrap.quantizer.quant])
    np.testing.assert_array_equal(np.array(trace, np.uint8), g["flag_trace"])
    np.testing.assert_array_equal(np.asarray(wrap.quantizer.quantizer.scale.numpy(), np.float32), g["s_x"])
    np.testing.assert_array_equal(np.asarray(wrap.quantizer.observer.min_val.numpy(), np.float32), g["obs_min"])
    np.testing.assert_array_equal(np.asarray(wrap.quantizer.observer.max_val.numpy(), np.float32), g["obs_max"])
    # the quantized forward is the HIP path: a CPU tensor must fail loudly, never fall back
    from mquant_amd._lib import MQuantHipError
    with pytest.raises(MQuantHipError):
        wrap(torch.from_numpy(make_x(seed + 20, (M, K_in))))
    qu.model_no_quant(wrap, args)
    assert wrap.quantizer.quant is False


def test_skip_names_are_left_untouched():
    m = torch.nn.Sequential(Block())
    qu.add_actquant(m)
    for w in qu.find_qlayers(m, layers=[qu.ActQuantWrapper]).values():
        w.quantizer.configure(bits=8, sym=True, static=True)
    a = Args()
    a.skip_names = ["seq"]
    qu.model_open_calibrate(m, a)
    flags = {n: w.quantizer.calibrate for n, w in qu.find_qlayers(m, layers=[qu.ActQuantWrapper]).items()}
    assert flags == {"0.q_proj": True, "0.seq.0": False, "0.seq.2": False, "0.lst.0": True}


def test_msq_calibration_keeps_two_scale_sets():
    w = qu.ActQuantWrapper(torch.nn.Linear(32, 8, bias=False))
    w.quantizer.configure(bits=8, sym=True, static=True, msq=True)
    x = torch.from_numpy(make_x(5, (12, 32)))
    x[:4] *= 10.0                                            # "vision" rows are much larger
    mask = torch.tensor([0] * 4 + [1] * 8)
    with qu.token_type_mask(mask):
        qu.calib_layer(w, [x, x])
    s_vis, s_txt = float(w.quantizer.quantizer.scale), float(w.quantizer.quantizer_text.scale)
    assert s_vis > 5 * s_txt
    exp_vis = max(abs(float(x[:4].min())) / 128, float(x[:4].max()) / 127)
    assert abs(s_vis - exp_vis) < 1e-6 * exp_vis
