#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE (read-only at /root/reference).

Runs only in the build container; the GPU box never sees /root/reference.
Outputs are data (inputs / expected outputs), written to tests/golden/*.npz and
fake_quant/had_signs.npz (the 11 special Hadamard matrices as packed sign bits).

The reference imports the third-party CUDA extension ``fast_hadamard_transform``
(Dao-AILab, un-pinned; reference docs/install.md:13-19) which is not installed
here.  A stand-in implementing its published algorithm (fp32 butterflies in
ascending stride, scale on store, output in the input dtype) is injected for the
import; everything else that runs is the reference's own code on CPU.
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden_inputs import make_ties, make_w, make_x  # noqa: E402

REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")


def _install_shims():
    fht = types.ModuleType("fast_hadamard_transform")

    def hadamard_transform(x, scale=1.0):
        n = x.shape[-1]
        y = x.reshape(-1, n).float().clone()
        h = 1
        while h < n:
            v = y.view(-1, n // (2 * h), 2, h)
            a, b = v[:, :, 0, :], v[:, :, 1, :]
            y = torch.stack((a + b, a - b), dim=2).reshape(-1, n)
            h *= 2
        y = y * torch.tensor(float(scale), dtype=torch.float32)
        return y.to(x.dtype).reshape(x.shape)

    fht.hadamard_transform = hadamard_transform
    sys.modules["fast_hadamard_transform"] = fht
    # `fake_quant` must resolve to the reference here, not to this repo's package
    sys.path.insert(0, REF)
    for k in [k for k in sys.modules if k == "fake_quant" or k.startswith("fake_quant.")]:
        del sys.modules[k]


def save(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"  {name}.npz  {os.path.getsize(path) / 1024:.1f} KiB")


class Args:
    skip_names = []


def main():
    _install_shims()
    torch.set_grad_enabled(False)
    from fake_quant import hadamard_utils as hu
    from fake_quant import quant_utils as qu
    from fake_quant import utils as ru
    from fake_quant.bit_type import BIT_TYPE_DICT
    from fake_quant.observer import build_observer
    from fake_quant.quantizer import build_quantizer

    assert hu.__file__.startswith(REF), hu.__file__

    # ------------------------------------------------------------------ 1
    print("[1] get_hadK / auto_pad_size / matrices")
    sizes = [12, 20, 28, 36, 40, 52, 60, 108, 140, 156, 172]
    mats = {}
    for k in sizes:
        h = getattr(hu, f"get_had{k}")()
        assert h.shape == (k, k) and torch.all(h.abs() == 1)
        mats[f"had{k}"] = np.packbits((h.numpy() > 0).astype(np.uint8).reshape(-1))
    np.savez_compressed(os.path.join(ROOT, "fake_quant", "had_signs.npz"), **mats)
    ns = [64, 1280, 3584, 4096, 5120, 8192, 11008, 14336, 19968, 30720, 768, 1664, 27648]
    ks = [hu.get_hadK(n)[1] for n in ns]
    pads_in = [18944, 29568, 3420, 4304, 11008, 4096, 14336, 1000, 13]
    pads_out = [int(hu.auto_pad_size(n)) for n in pads_in]
    save("hadk_table", n=np.array(ns), K=np.array(ks), pad_in=np.array(pads_in),
         pad_out=np.array(pads_out), **mats)

    # ------------------------------------------------------------------ 2
    print("[2] matmul_hadU / matmul_hadU_cuda")
    out = {}
    for n in [64, 1280, 3584, 4096, 5120, 11008, 14336, 19968, 30720]:
        x = make_x(100 + n, (2 if n <= 5120 else 1, n))
        xt = torch.from_numpy(x)
        hadK, K = hu.get_hadK(n)
        out[f"hadU_{n}"] = hu.matmul_hadU(xt).numpy()
        out[f"cuda_{n}"] = hu.matmul_hadU_cuda(xt, hadK, K).numpy()
        xh = xt.half()
        hk = None if hadK is None else hadK
        y16 = hu.matmul_hadU_cuda(xh, hk, K)
        out[f"cuda16_{n}"] = y16.numpy()  # float16 values, exact
    save("hadamard_fwd", **out)

    # ------------------------------------------------------------------ 3
    print("[3] random_hadamard_matrix")
    ru.seed_everything(42)
    state = torch.get_rng_state()
    Q = hu.random_hadamard_matrix(64, torch.device("cpu"))
    torch.set_rng_state(state)
    diag = (torch.randint(low=0, high=2, size=(64,)).to(torch.float64) * 2 - 1)
    ru.seed_everything(42)
    Q3584 = hu.random_hadamard_matrix(3584, torch.device("cpu"))
    ru.seed_everything(42)
    d3584 = (torch.randint(low=0, high=2, size=(3584,)).to(torch.float64) * 2 - 1)
    save("random_hadamard", Q64=Q.numpy(), diag64=diag.numpy(),
         Q3584_rows=Q3584[:4].numpy(), diag3584=d3584.numpy())

    # ------------------------------------------------------------------ 4
    print("[4] observers")
    out = {}
    bt = BIT_TYPE_DICT["int8"]
    batches = [make_x(7, (2, 5, 24)), np.abs(make_x(8, (2, 3, 24))) + 0.5,
               -np.abs(make_x(9, (1, 4, 24))) - 0.25]
    for mode in ["layer_wise", "channel_wise"]:
        ob = build_observer("minmax", "activation", bt, mode)
        for i, b in enumerate(batches):
            ob.update(torch.from_numpy(b))
            out[f"minmax_{mode}_max{i}"] = np.asarray(ob.max_val.numpy(), dtype=np.float32)
            out[f"minmax_{mode}_min{i}"] = np.asarray(ob.min_val.numpy(), dtype=np.float32)
        s, z = ob.get_quantization_params()
        out[f"minmax_{mode}_scale"] = s.numpy()
        out[f"minmax_{mode}_zp"] = z.numpy()
    # first batch all-positive / all-negative (zero-inclusion rule, minmax.py:17,22)
    for tag, b in [("pos", batches[1]), ("neg", batches[2])]:
        ob = build_observer("minmax", "activation", bt, "layer_wise")
        ob.update(torch.from_numpy(b))
        s, z = ob.get_quantization_params()
        out[f"minmax_first_{tag}"] = np.array([ob.min_val.item(), ob.max_val.item(), s.item()],
                                              dtype=np.float32)
    # 4-D (NCHW) reshape rule, base.py:24-27
    x4 = make_x(10, (2, 6, 3, 4))
    ob = build_observer("minmax", "activation", bt, "channel_wise")
    ob.update(torch.from_numpy(x4))
    out["minmax_4d_max"] = ob.max_val.numpy()
    out["minmax_4d_min"] = ob.min_val.numpy()
    # unsigned type -> asymmetric branch (minmax.py:47-51)
    ob = build_observer("minmax", "activation", BIT_TYPE_DICT["uint8"], "channel_wise")
    ob.update(torch.from_numpy(batches[0]))
    s, z = ob.get_quantization_params()
    out["minmax_uint8_scale"] = s.numpy()
    out["minmax_uint8_zp"] = z.numpy()
    for i, b in enumerate(batches):
        out[f"batch{i}"] = b
    out["x4"] = x4
    # the other registry entries, tiny inputs
    for name in ["ema", "omse", "percentile", "ptf"]:
        mode = "layer_wise" if name == "percentile" else "channel_wise"
        btype = BIT_TYPE_DICT["uint8"] if name in ("omse", "ptf") else bt
        ob = build_observer(name, "activation", btype, mode)
        for b in batches:
            ob.update(torch.from_numpy(b))
        out[f"{name}_max"] = np.asarray(ob.max_val.numpy(), dtype=np.float32)
        out[f"{name}_min"] = np.asarray(ob.min_val.numpy(), dtype=np.float32)
        if name in ("omse", "ptf"):
            s, z = ob.get_quantization_params(torch.from_numpy(batches[0]))
        else:
            s, z = ob.get_quantization_params()
        out[f"{name}_scale"] = np.asarray(s.numpy(), dtype=np.float32)
        out[f"{name}_zp"] = np.asarray(z.numpy())
    save("observers", **out)

    # ------------------------------------------------------------------ 5
    print("[5] UniformQuantizer")
    out = {}
    scale = np.float32(0.0371)
    for tag, shape in [("2d", (6, 40)), ("3d", (2, 5, 40)), ("4d", (2, 40, 3, 2)),
                       ("5d", (3, 40, 2, 2, 2))]:
        x = make_x(20 + len(shape), shape, outlier_gain=40.0)
        flat = x.reshape(-1)
        ties = make_ties(scale, np.arange(-130, 130, 7))
        flat[:ties.size] = ties
        flat[ties.size:ties.size + 4] = [1e9, -1e9, 127.5 * scale, -128.5 * scale]
        x = flat.reshape(shape)
        for mode in ["layer_wise", "channel_wise"]:
            ob = build_observer("minmax", "activation", bt, mode)
            qz = build_quantizer("uniform", bt, ob, "activation")
            if mode == "layer_wise":
                qz.scale = torch.tensor(scale)
                qz.zero_point = torch.tensor(0, dtype=torch.int64)
            else:
                c = 40
                sc = (np.float32(0.01) + np.arange(c, dtype=np.float32) * np.float32(0.003))
                qz.scale = torch.from_numpy(sc)
                qz.zero_point = torch.zeros(c, dtype=torch.int64)
                out[f"scale_cw"] = sc
            xt = torch.from_numpy(x)
            out[f"q_{tag}_{mode}"] = qz.quant(xt.float()).numpy().astype(np.int8)
            out[f"dq_{tag}_{mode}"] = qz(xt).numpy()
            out[f"dq16_{tag}_{mode}"] = qz(xt.half()).float().numpy()
            out[f"q16_{tag}_{mode}"] = qz.quant(xt.half().float()).numpy().astype(np.int8)
        out[f"x_{tag}"] = x
    out["scale_lw"] = scale
    save("uniform_quantizer", **out)

    # ------------------------------------------------------------------ 6
    print("[6] ActQuantizer dynamic")
    out = {}
    x = make_x(31, (2, 6, 256))
    x[0, 2, :] = 0.0
    out["x"] = x
    for tag, kw in [("tok_sym", dict(bits=8, sym=True)),
                    ("tok_asym", dict(bits=8, sym=False)),
                    ("tok_sym4", dict(bits=4, sym=True)),
                    ("tensor_sym", dict(bits=8, sym=True, act_per_tensor=True)),
                    ("tensor_asym", dict(bits=8, sym=False, act_per_tensor=True)),
                    ("group_sym", dict(bits=8, sym=True, groupsize=128)),
                    ("group_asym", dict(bits=8, sym=False, groupsize=128)),
                    ("clip_sym", dict(bits=8, sym=True, clip_ratio=0.9))]:
        aq = qu.ActQuantizer()
        aq.configure(**kw)
        xt = torch.from_numpy(x.copy())
        aq.find_params(xt)
        out[f"y_{tag}"] = aq(xt).numpy()
        sc = aq.scale
        out[f"scale_{tag}"] = np.asarray(sc.numpy() if torch.is_tensor(sc) else sc,
                                         dtype=np.float32)
    save("act_dynamic", **out)

    # ------------------------------------------------------------------ 7
    print("[7] WeightQuantizer")
    out = {}
    W = make_w(41, (24, 512))
    W[3, :] = 0.0
    W[5, 17] = 0.4
    out["W"] = W
    for tag, kw in [("w4_sym", dict(bits=4, perchannel=True, sym=True, mse=False)),
                    ("w4_sym_mse", dict(bits=4, perchannel=True, sym=True, mse=True)),
                    ("w8_sym", dict(bits=8, perchannel=True, sym=True, mse=False)),
                    ("w8_sym_mse", dict(bits=8, perchannel=True, sym=True, mse=True)),
                    ("w4_asym", dict(bits=4, perchannel=True, sym=False, mse=False)),
                    ("w4_asym_mse", dict(bits=4, perchannel=True, sym=False, mse=True)),
                    ("w4_tensor", dict(bits=4, perchannel=False, sym=True, mse=False))]:
        wq = qu.WeightQuantizer()
        wq.configure(**kw)
        Wt = torch.from_numpy(W.copy())
        wq.find_params(Wt)
        out[f"scale_{tag}"] = wq.scale.numpy()
        out[f"zero_{tag}"] = wq.zero.numpy()
        out[f"wq_{tag}"] = wq.quantize(Wt).numpy()
    save("weight_quantizer", **out)

    # ------------------------------------------------------------------ 8
    print("[8] pack_i4 / unpack_i4")
    rs = np.random.RandomState(51)
    q = rs.randint(-8, 8, size=(16, 64)).astype(np.int8)
    q[0, :16] = np.arange(-8, 8)
    packed = qu.pack_i4(torch.from_numpy(q))
    unpacked = qu.unpack_i4(packed)
    save("pack_i4", q=q, packed=packed.numpy(), unpacked=unpacked.numpy())

    # ------------------------------------------------------------------ 9 / 11
    print("[9] ActQuantWrapper.forward single layers + calibration trace")

    def run_layer(tag, K_in, K_pad, N, M, seed, had, split, w_bits=4, w_mse=False, bias=False,
                  dtype=torch.float32, fp32_had=False):
        out = {}
        lin = torch.nn.Linear(K_pad, N, bias=bias)
        W = make_w(seed, (N, K_pad))
        lin.weight.data = torch.from_numpy(W.copy())
        if bias:
            lin.bias.data = torch.from_numpy(make_w(seed + 1, (N,), std=0.1))
        lin = lin.to(dtype)
        wrap = qu.ActQuantWrapper(lin)
        if had:
            hadK, Kh = hu.get_hadK(K_pad)
            wrap.online_full_had = True
            wrap.had_K = hadK
            wrap.K = Kh
            wrap.fp32_had = fp32_had
            out["had_K"] = np.array(Kh)
        if split:
            wrap.split = True
            wrap.split_weights()
        if K_pad != K_in:
            import functools
            wrap.register_forward_pre_hook(
                functools.partial(ru.revise_down_input, new_size=K_pad))
        # offline weight fake-quant (RTN), gptq/qwen2vl_gptq_plus.py:381-407
        subset = qu.find_qlayers(wrap, layers=[torch.nn.Linear])
        wscale = None
        for name in subset:
            if "L1" in name:
                continue
            wq = qu.WeightQuantizer()
            wq.configure(w_bits, perchannel=True, sym=True, mse=w_mse)
            Wd = subset[name].weight.data
            wq.find_params(Wd)
            subset[name].weight.data = wq.quantize(Wd).to(Wd.dtype)
            if name in ("module", "L2"):
                wscale = wq.scale.float().numpy().reshape(-1)
                out["w_name"] = np.array(name)
        out["s_w"] = wscale
        # with split, module.weight[:,1:] is a view shared with L2 (quant_utils.py:325-326)
        # until RTN rebinds L2.weight.data; capture what forward will really use.
        wrap.quantizer.configure(bits=8, sym=True, static=True, observer_type="minmax")
        args = Args()
        qu.model_open_calibrate(wrap, args)
        calib = [make_x(seed + 10 + i, (M, K_in)) for i in range(3)]
        trace = []
        for i, c in enumerate(calib):
            if i == len(calib) - 1:
                qu.model_open_last_calibrate(wrap, args)
            y = wrap(torch.from_numpy(c.copy()).to(dtype))
            trace.append([wrap.quantizer.calibrate, wrap.quantizer.last_calibrate,
                          wrap.quantizer.quant])
            if i == 0:
                out["y_calib0"] = y.float().numpy()
        qu.model_close_calibrate(wrap, args)
        qu.model_quant(wrap, args)
        trace.append([wrap.quantizer.calibrate, wrap.quantizer.last_calibrate,
                      wrap.quantizer.quant])
        out["flag_trace"] = np.array(trace, dtype=np.uint8)
        s_x = wrap.quantizer.quantizer.scale
        out["s_x"] = np.asarray(s_x.numpy(), dtype=np.float32)
        out["obs_min"] = np.asarray(wrap.quantizer.observer.min_val.numpy(), dtype=np.float32)
        out["obs_max"] = np.asarray(wrap.quantizer.observer.max_val.numpy(), dtype=np.float32)
        x = make_x(seed + 20, (M, K_in))
        y = wrap(torch.from_numpy(x.copy()).to(dtype))
        out["y"] = y.float().numpy()
        # integer restatement from the reference's own quantizers
        xt = torch.from_numpy(x.copy()).to(dtype)
        if K_pad != K_in:
            xt = torch.nn.functional.pad(xt, (0, K_pad - K_in))
        if had:
            if fp32_had:
                xt = hu.matmul_hadU_cuda(xt.float(), wrap.had_K, wrap.K).to(dtype)
            else:
                xt = hu.matmul_hadU_cuda(xt, wrap.had_K, wrap.K)
        if had:
            out["x_rot"] = xt.float().numpy()[:, :512]
        xq_in = xt[..., 1:] if split else xt
        qx = wrap.quantizer.quantizer.quant(xq_in.float()).to(torch.int64)
        Wq = (wrap.L2.weight.data if split else wrap.module.weight.data).float()
        qw = torch.round(Wq / torch.from_numpy(wscale).reshape(-1, 1)).to(torch.int64)
        assert qw.abs().max() <= 2 ** (w_bits - 1)
        acc = qx @ qw.T
        assert acc.abs().max() < 2 ** 31
        out["acc"] = acc.numpy().astype(np.int32)
        out["qx_head"] = qx[:, :64].numpy().astype(np.int8)
        out["qx_sum"] = qx.sum(dim=1).numpy()
        out["qw_sum"] = qw.sum(dim=1).numpy()
        if split:
            out["w0"] = wrap.L1.weight.data.float().numpy().reshape(-1)
            out["x0"] = xt[..., 0].float().numpy()
        out["meta"] = np.array([K_in, K_pad, N, M, seed, int(had), int(split), w_bits,
                                int(w_mse), int(bias)])
        save(f"wrapper_{tag}", **out)

    # (i) plain static W4A8, Qwen2-VL q_proj K and Qwen-VL 4096 (config 1)
    run_layer("plain_3584", 3584, 3584, 48, 16, 600, had=False, split=False, bias=True)
    run_layer("plain_4096_mse", 4096, 4096, 32, 16, 610, had=False, split=False, w_mse=True)
    run_layer("plain_w8", 1280, 1280, 32, 16, 615, had=False, split=False, w_bits=8, bias=True)
    # (ii) vision fc2: online Hadamard K=40, n=5120, split (canonical --visual_split)
    run_layer("had_5120_split", 5120, 5120, 32, 16, 620, had=True, split=True, bias=True)
    run_layer("had_5120", 5120, 5120, 32, 16, 625, had=True, split=False, bias=True)
    run_layer("had_5120_fp32had", 5120, 5120, 32, 16, 627, had=True, split=False, fp32_had=True)
    # Qwen-VL c_proj: K=11008 (172 x 64); power-of-two visual 8192
    run_layer("had_11008", 11008, 11008, 16, 8, 630, had=True, split=False)
    run_layer("had_8192", 8192, 8192, 16, 8, 635, had=True, split=False)
    # (iii) LLM down_proj: pad 18944 -> 19968, K = 156 x 128, canonical (no split) and split
    run_layer("down_19968", 18944, 19968, 16, 8, 640, had=True, split=False)
    run_layer("down_19968_split", 18944, 19968, 16, 8, 645, had=True, split=True)
    # InternVL2 w2: 14336 = 28 x 512
    run_layer("had_14336", 14336, 14336, 16, 8, 650, had=True, split=False)

    # ------------------------------------------------------------------ 10
    print("[10] invariance pairs (offline Hadamard vs online)")
    torch.Tensor.cuda = lambda self, *a, **k: self  # apply_exact_had_to_linear hard-codes .cuda()
    out = {}
    for n in [5120, 1280]:
        lin = torch.nn.Linear(n, 24, bias=True)
        W = make_w(700 + n, (24, n))
        lin.weight.data = torch.from_numpy(W.copy())
        b = lin.bias.data.clone()
        hu.apply_exact_had_to_linear(lin, had_dim=-1, output=False)
        out[f"W_in_{n}"] = lin.weight.data.numpy()
        lin2 = torch.nn.Linear(24, n, bias=True)
        W2 = make_w(710 + n, (n, 24))
        lin2.weight.data = torch.from_numpy(W2.copy())
        lin2.bias.data = torch.from_numpy(make_w(711 + n, (n,), std=0.1))
        hu.apply_exact_had_to_linear(lin2, had_dim=-1, output=True)
        out[f"W_out_{n}"] = lin2.weight.data.numpy()
        out[f"b_out_{n}"] = lin2.bias.data.numpy()
    lin = torch.nn.Linear(512, 24, bias=False)
    lin.weight.data = torch.from_numpy(make_w(720, (24, 512)))
    hu.apply_exact_had_to_linear(lin, had_dim=128, output=False)
    out["W_headin_128"] = lin.weight.data.numpy()
    lin = torch.nn.Linear(24, 512, bias=True)
    lin.weight.data = torch.from_numpy(make_w(721, (512, 24)))
    lin.bias.data = torch.from_numpy(make_w(722, (512,), std=0.1))
    hu.apply_exact_had_to_linear(lin, had_dim=128, output=True)
    out["W_headout_128"] = lin.weight.data.numpy()
    out["b_headout_128"] = lin.bias.data.numpy()
    save("offline_hadamard", **out)
    print("done")


if __name__ == "__main__":
    main()
