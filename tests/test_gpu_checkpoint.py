"""Flat quantized checkpoint (mquant_amd/checkpoint.py): export -> safetensors ->
load into a freshly initialised model reproduces the quantized outputs bit for bit; the weight
bytes are the reference's pack_i4 wire format."""
import functools

import numpy as np
import pytest
import torch

import oracle
from golden_inputs import make_w, make_x

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
torch.set_grad_enabled(False)


class Args:
    skip_names = []


class Net(torch.nn.Module):
    def __init__(self, seed):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.plain = torch.nn.Linear(256, 96, bias=True)
        self.down = torch.nn.Linear(768, 64, bias=False)       # fed 700 features, padded, K = 12 * 64
        self.fc2 = torch.nn.Linear(1280, 80, bias=True)        # Hadamard 40 x 32 + split
        self.txt = torch.nn.Linear(256, 48, bias=False)        # MSQ
        self.w8 = torch.nn.Linear(128, 32, bias=True)
        for p in self.parameters():
            p.data = torch.randn(p.shape, generator=g) * (0.05 if p.dim() > 1 else 0.2)


def prepare(seed):
    from fake_quant import hadamard_utils as hu, quant_utils as qu, utils
    from fake_quant.gptq.rtn import rtn_module
    net = Net(seed).to(DEV).half()
    qu.add_actquant(net)
    net.down.register_forward_pre_hook(functools.partial(utils.revise_down_input, new_size=768))
    for wrap, n in ((net.down, 768), (net.fc2, 1280)):
        hadK, K = hu.get_hadK(n)
        wrap.online_full_had, wrap.had_K, wrap.K = True, hadK, K
    net.fc2.split = True
    net.fc2.split_weights()
    quantizers = {}
    for name, wrap in qu.find_qlayers(net, [qu.ActQuantWrapper]).items():
        rtn_module(wrap, name, 8 if name == "w8" else 4, True, name == "plain", [], quantizers)
        wrap.quantizer.configure(bits=8, sym=True, static=True, msq=(name == "txt"))
    return net


def inputs(i):
    return {"plain": torch.from_numpy(make_x(10 + i, (24, 256))).to(DEV).half(),
            "down": torch.from_numpy(make_x(20 + i, (24, 700))).to(DEV).half(),
            "fc2": torch.from_numpy(make_x(30 + i, (24, 1280))).to(DEV).half(),
            "txt": torch.from_numpy(make_x(40 + i, (24, 256))).to(DEV).half(),
            "w8": torch.from_numpy(make_x(50 + i, (24, 128))).to(DEV).half()}


def run(net, xs):
    from fake_quant import quant_utils as qu
    mask = torch.tensor([0] * 8 + [1] * 16, device=DEV)
    with qu.token_type_mask(mask):
        return {k: getattr(net, k)(v) for k, v in xs.items()}


def test_export_load_round_trip(tmp_path):
    from fake_quant import quant_utils as qu
    from mquant_amd import checkpoint
    net = prepare(1)
    qu.model_open_calibrate(net, Args())
    run(net, inputs(0))
    qu.model_open_last_calibrate(net, Args())
    run(net, inputs(1))
    qu.model_close_calibrate(net, Args())
    qu.model_quant(net, Args())
    xs = inputs(2)
    want = run(net, xs)
    path = str(tmp_path / "q.safetensors")
    tensors = checkpoint.save_quantized(net, path)
    assert tensors["down.qweight"].dtype == torch.uint8 and tuple(tensors["down.qweight"].shape) == (64, 384)
    assert tensors["w8.qweight"].dtype == torch.int8 and "fc2.w0" in tensors and "txt.bias" not in tensors
    assert float(tensors["txt.act_scale"][0]) != float(tensors["txt.act_scale"][1])
    meta = checkpoint.read_meta(tensors["fc2.meta"])
    assert (meta["had_K"], meta["split"], meta["K"], meta["N"], meta["w_bits"]) == (40, 1, 1280, 80, 4)
    # wire format == reference pack_i4 bytes of the levels the wrapper runs with
    lv = oracle.unpack_i4(tensors["plain.qweight"].numpy())
    W = net.plain.module.weight.data.float().cpu().numpy()
    s = tensors["plain.w_scale"].numpy()
    np.testing.assert_array_equal(lv, np.rint(W / s[:, None]).astype(np.int8))
    np.testing.assert_array_equal(lv[:, 0::2] & 0xf, tensors["plain.qweight"].numpy() & 0xf)

    fresh = Net(999).to(DEV).half()                              # different weights, never calibrated
    qu.add_actquant(fresh)
    assert checkpoint.load_quantized(fresh, path, DEV) == 5
    got = run(fresh, xs)
    for k in want:
        torch.testing.assert_close(got[k], want[k], rtol=0, atol=0, msg=k)
    # the frozen engines survive the calibration toggles
    qu.model_no_quant(fresh, Args())
    torch.testing.assert_close(run(fresh, xs)["down"], want["down"], rtol=0, atol=0)
    # stand-alone engines (no model object)
    engines = checkpoint.load_linears(path, DEV)
    torch.testing.assert_close(engines["plain"](xs["plain"]), want["plain"], rtol=0, atol=0)


def test_mismatched_checkpoint_is_refused(tmp_path):
    from fake_quant import quant_utils as qu
    from mquant_amd import checkpoint
    lin = torch.nn.Linear(64, 16).to(DEV).half()
    holder = torch.nn.Module()
    holder.a = lin
    qu.add_actquant(holder)
    with pytest.raises(ValueError):
        checkpoint.export_quantized(holder)                      # not calibrated / no weight quantizer
    bad = {"b.qweight": torch.zeros((16, 32), dtype=torch.uint8), "b.meta": torch.zeros(12, dtype=torch.int64)}
    with pytest.raises(ValueError):
        checkpoint.load_linears(bad, DEV)                        # version 0
    ok_meta = checkpoint._meta(version=1, w_bits=4, a_bits=8, N=16, K=64, in_features=64)
    rec = {"b.qweight": torch.zeros((16, 32), dtype=torch.uint8), "b.meta": ok_meta,
           "b.w_scale": torch.ones(16), "b.act_scale": torch.ones(2)}
    with pytest.raises(KeyError):
        checkpoint.load_quantized(holder, rec, DEV)              # names do not match the model


# ---- format version 2 (round 5): every configuration the integer backend runs round-trips bit for bit --------------------------
def _one(K, N, mode, seed):
    """A lone wrapper in one of the integer-path configurations, ready to run (weights quantized, activations configured)."""
    from fake_quant import hadamard_utils as hu, quant_utils as qu
    from fake_quant.gptq.rtn import rtn_module
    lin = torch.nn.Linear(K, N, bias=mode.get("bias", True))
    lin.weight.data = torch.from_numpy(make_w(seed, (N, K)))
    if mode.get("bias", True):
        lin.bias.data = torch.from_numpy(make_w(seed + 1, (N,), std=0.1))
    wrap = qu.ActQuantWrapper(lin.to(DEV).to(mode.get("dtype", torch.float16)))
    if mode.get("had"):
        wrap.had_K, wrap.K = hu.get_hadK(K)
        wrap.online_full_had = True
    if mode.get("split"):
        wrap.split = True
        wrap.split_weights()
    if mode.get("w_groups"):
        g = mode["w_groups"]
        W = wrap.module.weight.data.float()
        gs = (W.reshape(N, K // g, g).abs().amax(dim=2).clamp(min=1e-5) / 7)
        wrap.module.weight.data = (torch.round(W.reshape(N, K // g, g) / gs[:, :, None]).clamp(-8, 7) * gs[:, :, None]).reshape(N, K).to(wrap.module.weight.dtype)
        wq = qu.WeightQuantizer()
        wq.configure(4, perchannel=True, sym=True)
        wq.scale, wq.zero = gs[:, -1:].clone(), torch.zeros(N, 1, device=DEV)
        wq.groupsize, wq.group_permuted, wq.group_scales = g, False, gs
        qu.attach_weight_quantizer(wrap, "module", wq)
    else:
        rtn_module(wrap, "layer", mode.get("w_bits", 4), not mode.get("w_asym", False), False, [], {})
    act = mode["act"]
    if act == "static":
        wrap.quantizer.configure(bits=8, sym=True, static=True, observer_type="minmax")
        qu.calib_layer(wrap, [torch.from_numpy(make_x(seed + 10 + i, (32, K))).to(DEV).to(wrap.module.weight.dtype) for i in range(2)], Args())
    else:
        wrap.quantizer.act_per_tensor = bool(act.get("per_tensor", False))
        wrap.quantizer.configure(bits=act.get("bits", 8), sym=act.get("sym", True), groupsize=act.get("groupsize", -1),
                                 clip_ratio=act.get("clip", 1.0), act_per_tensor=bool(act.get("per_tensor", False)))
    return wrap


V2_MODES = {
    "static_w_asym": dict(act="static", w_asym=True),
    "static_w_asym_split_had": dict(act="static", w_asym=True, split=True, had=True),
    "dyn_sym": dict(act=dict(sym=True, clip=0.9)),
    "dyn_asym": dict(act=dict(sym=False)),
    "dyn_asym_w_asym": dict(act=dict(sym=False), w_asym=True),
    "dyn_asym_split": dict(act=dict(sym=False), split=True, bias=False),
    "dyn_asym_w_asym_split_had": dict(act=dict(sym=False), w_asym=True, split=True, had=True),     # all three rank-1 terms
    "dyn_per_tensor_sym": dict(act=dict(sym=True, per_tensor=True)),
    "dyn_per_tensor_asym": dict(act=dict(sym=False, per_tensor=True)),
    "dyn_a6": dict(act=dict(sym=True, bits=6)),
    "agrp_sym": dict(act=dict(sym=True, groupsize=128)),
    "agrp_asym": dict(act=dict(sym=False, groupsize=128)),
    "wgrp_static": dict(act="static", w_groups=128),
    "wgrp_dyn_had": dict(act=dict(sym=True), w_groups=128, had=True),
    "wgrp_agrp": dict(act=dict(sym=True, groupsize=128), w_groups=128),
    "w8_dyn": dict(act=dict(sym=True), w_bits=8),
}


@pytest.mark.parametrize("tag", sorted(V2_MODES))
def test_every_integer_path_mode_round_trips_bit_for_bit(tag, tmp_path):
    from fake_quant import quant_utils as qu
    from mquant_amd import checkpoint
    mode = V2_MODES[tag]
    K, N = 1280, 72
    holder = torch.nn.Module()
    holder.layer = _one(K, N, mode, seed=sum(map(ord, tag)))
    x = torch.from_numpy(make_x(7, (40, K))).to(DEV).to(holder.layer.module.weight.dtype)
    assert holder.layer._real_ready(x), (tag, holder.layer.backend())
    want = holder.layer(x.clone())
    assert holder.layer._real is not None
    path = str(tmp_path / f"{tag}.safetensors")
    tensors = checkpoint.save_quantized(holder, path)
    m = checkpoint.read_meta(tensors["layer.meta"])
    assert m["version"] == 2 and m["act_mode"] == (0 if mode["act"] == "static" else 1)
    assert ("layer.w_shift" in tensors) == bool(mode.get("w_asym")) and ("layer.w_group_scales" in tensors) == bool(mode.get("w_groups"))
    fresh = torch.nn.Module()
    fresh.layer = torch.nn.Linear(K, N, bias=mode.get("bias", True)).to(DEV).to(x.dtype)     # other weights, nothing configured
    qu.add_actquant(fresh)
    assert checkpoint.load_quantized(fresh, path, DEV) == 1
    assert "flat checkpoint" in fresh.layer.backend()
    got = fresh.layer(x.clone())
    torch.testing.assert_close(got, want, rtol=0, atol=0, msg=tag)


def test_version_1_records_still_load():
    from mquant_amd import checkpoint
    rec = {"qweight": torch.zeros((16, 64), dtype=torch.uint8), "w_scale": torch.ones(16), "act_scale": torch.tensor([0.5, 0.25]),
           "meta": checkpoint._meta(version=1, w_bits=4, a_bits=8, N=16, K=128, in_features=128, msq=1)}
    assert rec["meta"].numel() == 12
    eng = checkpoint.build_linear(rec, DEV)
    assert eng.dynamic is None and eng.s_x1 == 0.25 and eng.w_groups is None and eng.w_shift is None
    y = eng(torch.ones((4, 128), dtype=torch.float16, device=DEV))
    assert tuple(y.shape) == (4, 16) and not y.any()


def test_a_wrapper_that_simulates_is_refused_with_the_reason():
    from fake_quant import quant_utils as qu
    from mquant_amd import checkpoint
    wrap = _one(256, 16, dict(act=dict(sym=True)), 3)
    wrap.out_quantizer.configure(bits=8, sym=True)
    holder = torch.nn.Module()
    holder.layer = wrap
    with pytest.raises(ValueError, match="output quantizer"):
        checkpoint.export_quantized(holder)
