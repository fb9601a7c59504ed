"""Group-wise WEIGHT scales on the integer path (mq_gemm_w4a8_wgroupscale): the kernel against the oracle bit for bit, and the
wrapper over the reference's own GPTQ weights + per-group scales against the reference's forward
(tests/golden/wrapper_wgrp_*.npz, tools/gen_golden_wgroup.py)."""
import functools

import numpy as np
import pytest
import torch

import oracle
from golden_inputs import make_w, make_x
from test_wgroup_cpu import DT, TOL, cases, levels_of, load

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
torch.set_grad_enabled(False)
MODE = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}


def to_dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


@pytest.mark.parametrize("w_bits,out_dtype", [(4, torch.float16), (8, torch.float32), (4, torch.bfloat16), (4, torch.float32)])
@pytest.mark.parametrize("M,N,K,g", [(37, 200, 1280, 128), (130, 96, 2048, 64), (16, 48, 3584, 256), (300, 264, 1920, 64),
                                     (33, 72, 192, 64), (140, 200, 1984, 64), (768, 512, 1024, 1024)])
@pytest.mark.parametrize("act", ["static", "msq", "rows", "groups"])
def test_kernel_equals_the_oracle(w_bits, out_dtype, M, N, K, g, act):
    from mquant_amd import ops
    rng = np.random.default_rng(M + N + K + g + w_bits)
    K_pad = (K + 127) // 128 * 128
    G = K // g
    a = np.zeros((M, K_pad), np.int8)
    a[:, :K] = rng.integers(-128, 128, size=(M, K), dtype=np.int8)
    lim = 8 if w_bits == 4 else 128
    w = rng.integers(-lim, lim, size=(N, K), dtype=np.int8)
    wp = np.zeros((N, K_pad), np.int8)
    wp[:, :K] = w
    s_wg = (rng.random((G, N), dtype=np.float32) * 0.01 + 0.001).astype(np.float32)
    bias = rng.normal(size=N).astype(np.float32)
    kw, okw = {}, {}
    if act == "static":
        kw, okw = dict(s_x0=0.031), dict(sx0=0.031)
    elif act == "msq":
        sel = (rng.random(M) < 0.4).astype(np.uint8)
        kw, okw = dict(s_x0=0.031, s_x1=0.0077, row_sel=to_dev(sel)), dict(sx0=0.031, sx1=0.0077, row_sel=sel)
    elif act == "rows":
        rows = (rng.random(M, dtype=np.float32) * 0.05 + 0.01).astype(np.float32)
        kw, okw = dict(s_x_rows=to_dev(rows)), dict(sx_rows=rows)
    else:
        s_xg = (rng.random((M, G), dtype=np.float32) * 0.2 + 0.01).astype(np.float32)
        kw, okw = dict(s_x_groups=to_dev(s_xg)), dict(s_xg=s_xg)
    want = oracle.round_to(oracle.gemm_wgroup(a[:, :K], w, s_wg, g, bias=bias, **okw), MODE[out_dtype])
    img = ops.prepack(to_dev(wp), w_bits)
    for at in (ops.TiledAct.from_rows(to_dev(a)), to_dev(a)):                     # both activation layouts take the kernel
        y = ops.gemm_w4a8_wgroupscale(at, img, w_bits, N, to_dev(s_wg), g, bias=to_dev(bias), out_dtype=out_dtype, **kw)
        np.testing.assert_array_equal(y.float().cpu().numpy(), want)


@pytest.mark.parametrize("tile", [45, 46, 47, 48, 26])
@pytest.mark.parametrize("w_bits,out_dtype", [(4, torch.float16), (8, torch.bfloat16)])
@pytest.mark.parametrize("mode", ["w", "x", "wx"])
@pytest.mark.parametrize("M,N,K,g", [(300, 392, 1280, 128), (259, 136, 704, 64), (130, 260, 1024, 256)])
def test_every_tile_of_the_fold_and_the_round1_kernel_agree_with_the_oracle(tile, w_bits, out_dtype, mode, M, N, K, g):
    """The group scales fold inside the wave-specialised 16x16x64 tiles (gemm_ws.hip WG = 1 weights / 2 activations / 3 both);
    tile 26 keeps the round-1 kernel.  Every one of them against the oracle (ragged M and N, K_pad > K, groups of 64 / 128 / 256)."""
    from mquant_amd import ops
    rng = np.random.default_rng(M + N + K + g + w_bits + len(mode))
    K_pad = (K + 127) // 128 * 128
    G = K // g
    a = np.zeros((M, K_pad), np.int8)
    a[:, :K] = rng.integers(-128, 128, size=(M, K), dtype=np.int8)
    lim = 8 if w_bits == 4 else 128
    w = rng.integers(-lim, lim, size=(N, K), dtype=np.int8)
    wp = np.zeros((N, K_pad), np.int8)
    wp[:, :K] = w
    bias = rng.normal(size=N).astype(np.float32)
    s_wg = (rng.random((G, N), dtype=np.float32) * 0.01 + 0.001).astype(np.float32)
    s_xg = (rng.random((M, G), dtype=np.float32) * 0.2 + 0.01).astype(np.float32)
    s_w = (rng.random(N, dtype=np.float32) * 0.01 + 0.001).astype(np.float32)
    img = ops.prepack(to_dev(wp), w_bits)
    at = ops.TiledAct.from_rows(to_dev(a))
    try:
        ops.gemm_debug_force(tile, 0)
        if mode == "w":
            y = ops.gemm_w4a8_wgroupscale(at, img, w_bits, N, to_dev(s_wg), g, bias=to_dev(bias), out_dtype=out_dtype, s_x0=0.031)
            want = oracle.gemm_wgroup(a[:, :K], w, s_wg, g, bias=bias, sx0=0.031)
        elif mode == "wx":
            y = ops.gemm_w4a8_wgroupscale(at, img, w_bits, N, to_dev(s_wg), g, bias=to_dev(bias), out_dtype=out_dtype, s_x_groups=to_dev(s_xg))
            want = oracle.gemm_wgroup(a[:, :K], w, s_wg, g, bias=bias, s_xg=s_xg)
        else:
            y = ops.gemm_w4a8_groupscale(at, img, w_bits, N, to_dev(s_xg), g, to_dev(s_w), bias=to_dev(bias), out_dtype=out_dtype)
            acc = np.einsum("mgk,ngk->mgn", a[:, :K].reshape(M, G, g).astype(np.int64), w.reshape(N, G, g).astype(np.int64))
            f = np.zeros((M, N), np.float32)
            for gi in range(G):
                f = (f + (acc[:, gi, :].astype(np.float32) * s_xg[:, gi:gi + 1]).astype(np.float32)).astype(np.float32)
            want = ((f * s_w[None, :]).astype(np.float32) + bias[None, :]).astype(np.float32)
    finally:
        ops.gemm_debug_force(-1, 0)
    np.testing.assert_array_equal(y.float().cpu().numpy(), oracle.round_to(want, MODE[out_dtype]))


def test_entry_point_refuses_what_it_cannot_do():
    from mquant_amd import ops
    from mquant_amd._lib import MQuantHipError
    a = torch.zeros((16, 256), dtype=torch.int8, device=DEV)
    img = ops.prepack(torch.zeros((32, 256), dtype=torch.int8, device=DEV), 4)
    s_wg = torch.ones((2, 32), device=DEV)
    with pytest.raises(MQuantHipError):                                        # groups that do not cover K
        ops.gemm_w4a8_wgroupscale(a, img, 4, 32, torch.ones((1, 32), device=DEV), 64)
    with pytest.raises(MQuantHipError):                                        # group size
        ops.gemm_w4a8_wgroupscale(a, img, 4, 32, torch.ones((8, 32), device=DEV), 32)
    with pytest.raises(MQuantHipError):                                        # group-wise activations exclude per-row scales
        ops.gemm_w4a8_wgroupscale(a, img, 4, 32, s_wg, 128, s_x_rows=torch.ones(16, device=DEV), s_x_groups=torch.ones((16, 2), device=DEV))
    assert ops.gemm_w4a8_wgroupscale(a, img, 4, 32, s_wg, 128).shape == (16, 32)


def _wrapper_from_golden(g, c):
    from fake_quant import hadamard_utils as hu, quant_utils as qu, utils
    dt = DT[c["dtc"]]
    lin = torch.nn.Linear(c["K_pad"], c["N"], bias=c["bias"])
    lin.weight.data = torch.from_numpy(g["W"].copy())
    if c["bias"]:
        lin.bias.data = torch.from_numpy(make_w(c["seed"] + 1, (c["N"],), std=0.1))
    wrap = qu.ActQuantWrapper(lin.to(dt).to(DEV))
    if c["had"]:
        hadK, Kh = hu.get_hadK(c["K_pad"])
        wrap.online_full_had, wrap.had_K, wrap.K = True, hadK, Kh
    if c["K_pad"] != c["K_in"]:
        wrap.register_forward_pre_hook(functools.partial(utils.revise_down_input, new_size=c["K_pad"]))
    wq = qu.WeightQuantizer()
    wq.configure(c["w_bits"], perchannel=True, sym=True, mse=False)
    gs = torch.from_numpy(g["group_scales"].copy())
    wq.scale, wq.zero = gs[:, -1:].clone(), torch.zeros(c["N"], 1)            # what the reference's solver leaves behind ...
    wq.groupsize, wq.group_permuted, wq.group_scales = c["g"], False, gs      # ... and what this repository's keeps
    if "perm" in g:                                                            # --act_order: the solver's column permutation
        wq.group_permuted, wq.group_perm = True, torch.from_numpy(g["perm"].copy())
    qu.attach_weight_quantizer(wrap, "module", wq)
    return wrap


class Args:
    skip_names = []


def test_wrapper_over_the_references_gptq_weights_matches_the_references_forward(golden_dir):
    from fake_quant import quant_utils as qu
    from mquant_amd import ops
    paths = cases(golden_dir)
    assert len(paths) == 11
    for path in paths:
        g, c = load(path)
        dt = DT[c["dtc"]]
        wrap = _wrapper_from_golden(g, c)
        if c["mode"] == "static":
            wrap.quantizer.configure(bits=8, sym=True, static=True, observer_type="minmax")
            qu.calib_layer(wrap, [torch.from_numpy(make_x(c["seed"] + 10 + i, (c["M"], c["K_in"]))).to(dt).to(DEV) for i in range(3)], Args())
            np.testing.assert_array_equal(np.asarray(wrap.quantizer.quantizer.scale.cpu().numpy(), np.float32), g["s_x"], err_msg=path)
        elif c["mode"] == "dyn":
            wrap.quantizer.configure(bits=8, sym=True)
        else:
            wrap.quantizer.configure(bits=8, groupsize=c["g"], sym=True, clip_ratio=1.0)
        shape = (1, c["M"], c["K_in"]) if c["mode"] == "agrp" else (c["M"], c["K_in"])
        x = torch.from_numpy(make_x(c["seed"] + 20, shape)).to(dt).to(DEV)
        assert wrap._real_ready(x), (path, wrap.backend())
        assert "weight groups of %d" % c["g"] in wrap.backend()
        y = wrap(x)
        real = wrap._real
        assert real is not None and real.w_groups is not None and real.w_groups[1] == c["g"]
        tol = TOL[c["dtc"]] * float(np.abs(g["y"]).max())
        np.testing.assert_allclose(y.float().cpu().numpy().reshape(c["M"], c["N"]), g["y"], rtol=0, atol=tol, err_msg=path)
        # the integer levels the engine froze are the reference's, group by group
        lv = levels_of(g, c)
        img = ops.prepack(to_dev(lv), c["w_bits"])
        assert torch.equal(real.w_img, img), path
        np.testing.assert_array_equal(real.w_groups[0].cpu().numpy(), g["group_scales"].T, err_msg=path)
        # ... and the activation levels / scales equal the reference's own quantizer's
        rows = x.reshape(c["M"], c["K_in"])
        rows = torch.nn.functional.pad(rows, (0, c["K_pad"] - c["K_in"])) if c["K_pad"] != c["K_in"] else rows
        xr = ops.hadamard(rows, real.had.n, real.had.K, real.had.bits) if c["had"] else rows
        bias = None if not c["bias"] else wrap.module.bias.data.float().cpu().numpy()
        if "perm" in g:                                # --act_order: the engine gathers the (rotated) columns into the solver's order
            assert real.col_perm is not None and torch.equal(real.col_perm.cpu(), torch.from_numpy(g["perm"]))
            xr = xr.index_select(1, real.col_perm)
        if c["mode"] == "static":
            a, _ = real.quantize(rows)
            qx = a.to_rows() if isinstance(a, ops.TiledAct) else a
            np.testing.assert_array_equal(qx[:, :c["K_pad"]].cpu().numpy(), g["qx"], err_msg=path)
            want = oracle.gemm_wgroup(g["qx"], lv, g["group_scales"].T.copy(), c["g"], sx0=float(g["s_x"]), bias=bias)
        elif c["mode"] == "dyn":
            a, s_rows, _ = ops.quantize_act_dyn_i8(xr, 8, 1.0)
            np.testing.assert_array_equal(s_rows.cpu().numpy(), g["s_x_rows"], err_msg=path)
            np.testing.assert_array_equal(a[:, :c["K_pad"]].cpu().numpy(), g["qx"], err_msg=path)
            want = oracle.gemm_wgroup(g["qx"], lv, g["group_scales"].T.copy(), c["g"], sx_rows=g["s_x_rows"], bias=bias)
        else:
            a, s_g = ops.quantize_act_group_i8(xr, c["g"], 8, 1.0)
            np.testing.assert_array_equal(s_g.cpu().numpy(), g["s_x_groups"], err_msg=path)
            np.testing.assert_array_equal(a[:, :c["K_pad"]].cpu().numpy(), g["qx"], err_msg=path)
            want = oracle.gemm_wgroup(g["qx"], lv, g["group_scales"].T.copy(), c["g"], s_xg=g["s_x_groups"], bias=bias)
        # the wrapper's output IS the oracle's evaluation of the reference's integers, bit for bit
        np.testing.assert_array_equal(y.float().cpu().numpy().reshape(c["M"], c["N"]), oracle.round_to(want, c["dtc"]), err_msg=path)


@pytest.mark.parametrize("actorder", [False, True])
def test_gptq_with_weight_groups_runs_end_to_end_on_the_integer_path(actorder):
    """This repository's solver on the GPU (groups of 128, with and without --act_order) -> attach -> static calibration -> the wrapper
    runs mq_gemm_w4a8_wgroupscale and stays on the simulated evaluation of the same wrapper."""
    from fake_quant import quant_utils as qu
    from fake_quant.gptq.gptq_utils import GPTQ
    K, N, M = 1024, 96, 64
    lin = torch.nn.Linear(K, N, bias=True)
    lin.weight.data = torch.from_numpy(make_w(5, (N, K))) * 4.0
    wrap = qu.ActQuantWrapper(lin.to(DEV))
    solver = GPTQ(wrap.module)
    solver.quantizer = qu.WeightQuantizer()
    solver.quantizer.configure(4, perchannel=True, sym=True, mse=False)
    for i in range(3):
        solver.add_batch(torch.from_numpy(make_x(50 + i, (1, 80, K))).to(DEV))
    solver.fasterquant(percdamp=0.01, groupsize=128, actorder=actorder, static_groups=False)
    qu.attach_weight_quantizer(wrap, "module", solver.quantizer)
    wrap.quantizer.configure(bits=8, sym=True, static=True, observer_type="minmax")
    qu.calib_layer(wrap, [torch.from_numpy(make_x(60 + i, (M, K))).to(DEV) for i in range(2)], Args())
    x = torch.from_numpy(make_x(70, (M, K))).to(DEV)
    assert wrap._real_ready(x) and "weight groups of 128" in wrap.backend()
    y = wrap(x)
    assert (wrap._real.col_perm is not None) == actorder
    wrap.real_quant = False
    y_sim = wrap(x.clone())
    np.testing.assert_allclose(y.cpu().numpy(), y_sim.cpu().numpy(), rtol=0, atol=1e-3 * float(y_sim.abs().max()))


def test_sibling_fusion_with_weight_groups_equals_the_per_linear_evaluation():
    """q / k / v wrappers whose weights carry group scales of the same group size fuse into ONE quantize + ONE grouped GEMM
    (their [groups, channels] scale tables side by side): bit-identical to evaluating every Linear on its own."""
    import types
    from fake_quant import quant_utils as qu

    class Attn(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.q_proj = torch.nn.Linear(512, 256, bias=True)
            self.k_proj = torch.nn.Linear(512, 64, bias=True)
            self.v_proj = torch.nn.Linear(512, 64, bias=True)

        def forward(self, x):
            return self.q_proj(x), self.k_proj(x), self.v_proj(x)

    class Holder(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.attn = Attn()

        def forward(self, x):
            return self.attn(x)

    torch.manual_seed(3)
    model = Holder().to(DEV)
    qu.add_actquant(model)
    g = 128
    for name in ("q_proj", "k_proj", "v_proj"):
        wrap = getattr(model.attn, name)
        W = wrap.module.weight.data.float()
        N, K = W.shape
        gs = W.reshape(N, K // g, g).abs().amax(dim=2).clamp(min=1e-5) / 7
        wrap.module.weight.data = (torch.round(W.reshape(N, K // g, g) / gs[:, :, None]).clamp(-8, 7) * gs[:, :, None]).reshape(N, K)
        wq = qu.WeightQuantizer()
        wq.configure(4, perchannel=True, sym=True)
        wq.scale, wq.zero = gs[:, -1:].clone(), torch.zeros(N, 1, device=DEV)
        wq.groupsize, wq.group_permuted, wq.group_scales = g, False, gs
        qu.attach_weight_quantizer(wrap, "module", wq)
        wrap.quantizer.configure(bits=8, sym=True, static=True, observer_type="minmax")
    qu.calib_layer(model, [torch.from_numpy(make_x(40 + i, (24, 512))).to(DEV) for i in range(2)], Args())
    x = torch.from_numpy(make_x(50, (24, 512))).to(DEV)
    fused = [t.clone() for t in model(x)]
    grp = model.attn.q_proj.__dict__.get("_group")
    assert grp is not None and grp.enabled and grp.engine is not None and grp.engine.w_groups is not None and grp.launches == 1
    assert tuple(grp.engine.w_groups[0].shape) == (4, 256 + 64 + 64)
    qu.model_quant(model, types.SimpleNamespace(skip_names=[], no_sibling_fusion=True))
    plain = model(x)
    assert all(torch.equal(a, b) for a, b in zip(fused, plain))
    assert model.attn.k_proj._real is not None and model.attn.k_proj._real.w_groups is not None


def test_act_order_groups_round_trip_through_the_flat_checkpoint(golden_dir):
    """--act_order + --w_groupsize: the record keeps the column permutation (``col_perm``) and the rebuilt engine gives the wrapper's bits."""
    import os
    from fake_quant import quant_utils as qu
    from mquant_amd import checkpoint
    path = os.path.join(golden_dir, "wrapper_wgrp_ao_g128_static_1024_f32.npz")
    g, c = load(path)
    wrap = _wrapper_from_golden(g, c)
    wrap.quantizer.configure(bits=8, sym=True, static=True, observer_type="minmax")
    qu.calib_layer(wrap, [torch.from_numpy(make_x(c["seed"] + 10 + i, (c["M"], c["K_in"]))).to(DEV) for i in range(3)], Args())
    x = torch.from_numpy(make_x(c["seed"] + 20, (c["M"], c["K_in"]))).to(DEV)
    y = wrap(x)
    assert "weight groups of 128" in wrap.backend()
    rec = checkpoint.export_wrapper(wrap)
    assert "col_perm" in rec and rec["col_perm"].dtype == torch.int64 and rec["col_perm"].numel() == c["K_pad"]
    eng = checkpoint.build_linear(rec, torch.device(DEV))
    assert torch.equal(eng.forward(x), y)
