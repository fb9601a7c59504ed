"""GPTQ solver (fake_quant/gptq/gptq_utils.py) against the reference's GPTQ / GPTQConv outputs
(tests/golden/gptq_*.npz from tools/gen_golden_gptq.py), and the layer-sequential driver on toy
models."""
import os
import sys
import types

import numpy as np
import pandas as pd
import pytest
import torch

from golden_inputs import make_w, make_x

torch.set_grad_enabled(False)
CASES = ["plain", "actorder", "groups", "mse_w8", "wide", "conv2d"]


def build(kind, n_out, n_in, seed):
    if kind == "linear":
        layer = torch.nn.Linear(n_in, n_out, bias=False)
        layer.weight.data = torch.from_numpy(make_w(seed, (n_out, n_in))) * 4.0
        xs = [torch.from_numpy(make_x(seed + 1 + i, (2 * 10, n_in))).reshape(2, 10, n_in) for i in range(3)]
    else:
        layer = torch.nn.Conv2d(3, n_out, kernel_size=4, stride=4, bias=False)
        layer.weight.data = (torch.from_numpy(make_w(seed, (n_out, n_in))) * 4.0).reshape(n_out, 3, 4, 4)
        xs = [torch.from_numpy(make_x(seed + 1 + i, (12, n_in))).reshape(12, 3, 4, 4) for i in range(3)]
    return layer, xs


@pytest.mark.parametrize("case", CASES)
def test_solver_matches_reference(golden_dir, case):
    from fake_quant import quant_utils as qu
    from fake_quant.gptq.gptq_utils import GPTQ
    g = np.load(os.path.join(golden_dir, f"gptq_{case}.npz"))
    seed, n_out, n_in, bits, mse, actorder, groupsize = [int(v) for v in g["meta"]]
    layer, xs = build("conv" if case == "conv2d" else "linear", n_out, n_in, seed)
    w_before = layer.weight.data.clone()
    solver = GPTQ(layer)
    solver.quantizer = qu.WeightQuantizer()
    solver.quantizer.configure(bits, perchannel=True, sym=True, mse=bool(mse))
    for x in xs:
        solver.add_batch(x, layer(x))
    if g["H"].size:
        np.testing.assert_array_equal(solver.H.numpy(), g["H"])
    solver.fasterquant(percdamp=0.01, groupsize=groupsize, actorder=bool(actorder), static_groups=False)
    np.testing.assert_array_equal(solver.quantizer.scale.reshape(-1).numpy(), g["scale"])
    np.testing.assert_array_equal(layer.weight.data.reshape(n_out, -1).numpy(), g["Q"])
    assert layer.weight.shape == w_before.shape
    if groupsize == -1:                                            # on the int grid of its scale
        lv = layer.weight.data.reshape(n_out, -1).numpy() / g["scale"][:, None]
        np.testing.assert_allclose(lv, np.rint(lv), atol=1e-4)
    # the point of GPTQ: smaller output error than plain rounding on the calibration inputs
    rtn = qu.WeightQuantizer()
    rtn.configure(bits, perchannel=True, sym=True, mse=bool(mse))
    rtn.find_params(w_before)
    x = torch.cat([t.reshape(-1, n_in) for t in xs])
    def out_err(w):
        return float(((x @ (w.reshape(n_out, -1) - w_before.reshape(n_out, -1)).T) ** 2).sum())
    assert out_err(layer.weight.data) < out_err(rtn.quantize(w_before))


class OracleBlockGPTQ:
    """Mixin: routes the solver's per-block column loop through oracle.gptq_block on the CPU (the
    slot mq_gptq_block fills on the GPU)."""
    use_kernel = True
    calls = 0

    @staticmethod
    def _kernel_device(W):
        return True

    @classmethod
    def _block(cls, W, i1, i2, Hrows, scale, bits, Q, E1):
        import oracle
        q, e = oracle.gptq_block(W[:, i1:i2].numpy(), Hrows[i1:i2, i1:i2].numpy(), scale.numpy(), bits)
        Q[:, i1:i2] = torch.from_numpy(q)
        E1.copy_(torch.from_numpy(e))
        cls.calls += 1


@pytest.mark.parametrize("case", ["plain", "actorder", "mse_w8", "wide", "conv2d"])
def test_oracle_block_loop_reproduces_the_reference_goldens(golden_dir, case):
    """orc_gptq_block (oracle/mq_oracle.c, restating gptq_utils.py:249-286) pinned: with the column
    loop of every block computed by the oracle the solver lands on the reference's Q bit for bit."""
    from fake_quant import quant_utils as qu
    from fake_quant.gptq.gptq_utils import GPTQ
    g = np.load(os.path.join(golden_dir, f"gptq_{case}.npz"))
    seed, n_out, n_in, bits, mse, actorder, groupsize = [int(v) for v in g["meta"]]
    assert groupsize == -1
    layer, xs = build("conv" if case == "conv2d" else "linear", n_out, n_in, seed)
    Solver = type("Solver", (OracleBlockGPTQ, GPTQ), {})
    solver = Solver(layer)
    solver.quantizer = qu.WeightQuantizer()
    solver.quantizer.configure(bits, perchannel=True, sym=True, mse=bool(mse))
    for x in xs:
        solver.add_batch(x, None)
    solver.fasterquant(percdamp=0.01, groupsize=-1, actorder=bool(actorder), static_groups=False)
    assert Solver.calls == -(-n_in // 128)
    np.testing.assert_array_equal(layer.weight.data.reshape(n_out, -1).numpy(), g["Q"])


def test_oracle_block_loop_equals_torch_loop_with_errors():
    """Err1 is not in the goldens (it only feeds the trailing update): check it against the torch loop."""
    import oracle
    N, cols, bits = 37, 100, 4
    W = torch.from_numpy(make_w(5, (N, cols))) * 3.0
    X = torch.from_numpy(make_x(6, (cols + 64, cols)))
    H = X.T @ X / X.shape[0]
    H += 0.01 * torch.mean(torch.diag(H)) * torch.eye(cols)
    Hinv = torch.linalg.cholesky(torch.cholesky_inverse(torch.linalg.cholesky(H)), upper=True).contiguous()
    scale = W.abs().amax(1) / 7
    W1 = W.clone()
    Q1, E1 = torch.zeros_like(W1), torch.zeros_like(W1)
    for i in range(cols):
        w, d = W1[:, i], Hinv[i, i]
        q = scale * torch.clamp(torch.round(w / scale), -8, 7)
        Q1[:, i] = q
        err = (w - q) / d
        W1[:, i:] -= err.unsqueeze(1) @ Hinv[i, i:].unsqueeze(0)
        E1[:, i] = err
    q, e = oracle.gptq_block(W.numpy(), Hinv.numpy(), scale.numpy(), bits)
    np.testing.assert_array_equal(q, Q1.numpy())
    np.testing.assert_array_equal(e, E1.numpy())


# ------------------------------------------------------------------------------------ drivers
class ToyVlm:
    """VLMEvalKit-style wrapper around a toy HF module: .model, .generate(message=, dataset=)."""

    def __init__(self, kind, seed=5):
        import toy_models
        self.kind = kind
        self.model, self.pixels, self.ids = toy_models.build(kind, seed)
        self.model = self.model.float()
        self.pixels = self.pixels.float()
        self.calls = 0

    def generate(self, message, dataset):
        self.calls += 1
        g = torch.Generator().manual_seed(int(message))
        return self.model(self.pixels + 0.1 * torch.randn(self.pixels.shape, generator=g), self.ids)


class ToyDataset:
    def __init__(self, n):
        self.data = pd.DataFrame({"v": list(range(n))})

    def build_prompt(self, record):
        return int(record["v"])


def gptq_args(**over):
    base = dict(quant_llm=True, quant_visual_clip=True, quant_cross_attention=True, act_per_tensor=False,
                visual_w_rtn=False, llm_w_rtn=False, visual_w_bits=4, llm_w_bits=4, w_asym=False,
                visual_w_clip=False, llm_w_clip=False, visual_split=False, llm_split=False, nsamples=3,
                percdamp=0.01, w_groupsize=-1, act_order=False, skip_names=[], dataset_name="toy")
    base.update(over)
    return types.SimpleNamespace(**base)


def _on_grid(weight, quantizer):
    lv = weight.reshape(weight.shape[0], -1) / quantizer.scale.reshape(-1, 1).to(weight.dtype)
    return bool(torch.allclose(lv, torch.round(lv), atol=1e-3))


@pytest.mark.parametrize("kind", ["qwen2vl", "internvl"])
def test_layer_sequential_gptq_driver(kind):
    from fake_quant import quant_utils as qu
    from fake_quant.gptq import internvl_gptq_plus, qwen2vl_gptq_plus
    vlm = ToyVlm(kind)
    args = gptq_args()
    ref_out = vlm.generate(0, "toy")
    (qu.qwen2vl_add_act_qaunt if kind == "qwen2vl" else qu.internvl_add_act_qaunt)(vlm, args)
    driver = (qwen2vl_gptq_plus.qwen2vl_rtn_gptq_fwrd_plus if kind == "qwen2vl"
              else internvl_gptq_plus.internvl_rtn_gptq_fwrd_plus)
    quantizers = driver(vlm, ToyDataset(8), "cpu", "toy", args)
    wrappers = qu.find_qlayers(vlm.model, [qu.ActQuantWrapper])
    # every wrapped layer got its quantizer under the upstream key, weights sit on the int4 grid,
    # and the wrapper knows its quantizer (what the real W4A8 path needs)
    if kind == "qwen2vl":
        expect = {"model.visual.patch_embed.proj.module", "model.visual.blocks.1.mlp.fc2.module",
                  "model.visual.merger.mlp.2.module", "model.model.layers.0.self_attn.k_proj.module",
                  "model.model.layers.1.mlp.down_proj.module"}
    else:
        expect = {"model.vision_model.embeddings.patch_embedding", "model.vision_model.encoder.layers.0.attn.qkv.module",
                  "model.mlp1.3.module", "model.language_model.model.layers.1.attention.wqkv.module",
                  "model.language_model.model.layers.0.feed_forward.w2.module"}
    assert expect <= set(quantizers), sorted(quantizers)
    assert len(quantizers) == len(wrappers)
    for name, w in wrappers.items():
        assert "module" in w.weight_quantizers, name
        assert _on_grid(w.module.weight.data, w.weight_quantizers["module"]), name
    # forwards were aborted at the captured module: no full generate after the captures finished,
    # and the catcher is gone
    out = vlm.generate(0, "toy")
    assert torch.isfinite(out).all()
    rel = float((out - ref_out).norm() / ref_out.norm())
    assert rel < 0.5, rel


def test_gptq_beats_rtn_on_block_outputs_and_respects_split():
    """One toy tower: the error of the final logits is smaller with GPTQ than with RTN, and with
    --llm_split the solver works on ``down_proj.L2`` (column 0 stays in L1)."""
    from fake_quant import quant_utils as qu
    from fake_quant.gptq import qwen2vl_gptq_plus
    errs = {}
    for rtn in (True, False):
        vlm = ToyVlm("qwen2vl", seed=9)
        want = vlm.generate(3, "toy")
        args = gptq_args(quant_visual_clip=False, quant_cross_attention=False, llm_w_rtn=rtn, llm_split=not rtn,
                         nsamples=6)
        qu.qwen2vl_add_act_qaunt(vlm, args)
        if not rtn:
            for layer in vlm.model.model.layers:
                layer.mlp.down_proj.split = True
                layer.mlp.down_proj.split_weights()
        q = qwen2vl_gptq_plus.qwen2vl_rtn_gptq_fwrd_plus(vlm, ToyDataset(8), "cpu", "toy", args)
        if not rtn:
            assert "model.model.layers.0.mlp.down_proj.L2" in q and "L2" in vlm.model.model.layers[0].mlp.down_proj.weight_quantizers
        errs[rtn] = float((vlm.generate(3, "toy") - want).norm())
    assert errs[False] < errs[True], errs


def test_capture_restores_the_module_and_stops_early():
    from fake_quant.gptq import sequential as seq
    vlm = ToyVlm("qwen2vl")
    target = vlm.model.model.layers[0]
    args = gptq_args()
    samples = seq.capture_inputs(target, lambda enough: seq.run_calibration_prompts(vlm, ToyDataset(8), "toy", args, enough), 2)
    assert len(samples) == 2 and vlm.calls == 2 and "forward" not in target.__dict__
    assert samples[0][0][0].shape == (7, 64)


@pytest.mark.parametrize("rtn", [False, True])
def test_qwenvl_v1_driver(rtn):
    from fake_quant import quant_utils as qu
    from fake_quant.gptq import qwenvl_gptq_plus
    vlm = ToyVlm("qwenvl")
    args = gptq_args(visual_w_rtn=rtn, llm_w_rtn=rtn)
    ref_out = vlm.generate(0, "toy")
    qu.qwenvl_add_act_qaunt(vlm.model, args)
    quantizers = qwenvl_gptq_plus.qwenvl_rtn_gptq_fwrd_plus(vlm, ToyDataset(8), "cpu", args)
    wrappers = qu.find_qlayers(vlm.model, [qu.ActQuantWrapper])
    expect = {"model.transformer.visual.conv1", "model.transformer.visual.transformer.resblocks.1.mlp.c_proj.module",
              "model.model.transformer.visual.attn_pool.kv_proj.module", "model.model.transformer.visual.attn_pool.attn.out_proj.module",
              "model.model.transformer.visual.proj_fc", "model.transformer.h.0.attn.c_proj.module",
              "model.transformer.h.1.mlp.w2.module"}
    assert expect <= set(quantizers), sorted(quantizers)
    assert len(quantizers) == len(wrappers)
    for name, w in wrappers.items():
        assert "module" in w.weight_quantizers and _on_grid(w.module.weight.data, w.weight_quantizers["module"]), name
    out = vlm.generate(0, "toy")
    assert torch.isfinite(out).all() and float((out - ref_out).norm() / ref_out.norm()) < 0.5


def test_minicpmv_driver_walks_the_expected_modules():
    """No MiniCPM-V toy forward: a module tree with the right names, identity-like blocks."""
    import torch.nn as nn
    from fake_quant import quant_utils as qu
    from fake_quant.gptq import minicpmv_gptq_plus

    class Attn(nn.Module):
        def __init__(self, d, out_name):
            super().__init__()
            self.q_proj, self.k_proj, self.v_proj = nn.Linear(d, d), nn.Linear(d, d), nn.Linear(d, d)
            setattr(self, out_name, nn.Linear(d, d))
            self.out_name = out_name

        def forward(self, x):
            return getattr(self, self.out_name)(torch.tanh(self.q_proj(x) + self.k_proj(x)) * self.v_proj(x))

    class VisLayer(nn.Module):
        def __init__(self, d):
            super().__init__()
            self.self_attn, self.mlp = Attn(d, "out_proj"), nn.Module()
            self.mlp.fc1, self.mlp.fc2 = nn.Linear(d, 2 * d), nn.Linear(2 * d, d)

        def forward(self, x):
            x = x + self.self_attn(x)
            return (x + self.mlp.fc2(torch.relu(self.mlp.fc1(x))),)        # HF layers return tuples

    class LlmLayer(nn.Module):
        def __init__(self, d):
            super().__init__()
            self.self_attn, self.mlp = Attn(d, "o_proj"), nn.Module()
            self.mlp.gate_proj, self.mlp.up_proj, self.mlp.down_proj = nn.Linear(d, 2 * d), nn.Linear(d, 2 * d), nn.Linear(2 * d, d)

        def forward(self, x, position_ids=None):
            x = x + self.self_attn(x)
            return (x + self.mlp.down_proj(torch.sigmoid(self.mlp.gate_proj(x)) * self.mlp.up_proj(x)),)

    class Resampler(nn.Module):
        def __init__(self, d):
            super().__init__()
            self.kv_proj, self.attn, self.proj_fc = nn.Linear(d, d), Attn(d, "out_proj"), nn.Linear(d, d)

        def forward(self, x):
            return self.proj_fc(self.attn(self.kv_proj(x)))

    class Hf(nn.Module):
        def __init__(self, d=32):
            super().__init__()
            self.vpm, self.llm = nn.Module(), nn.Module()
            self.vpm.embeddings, self.vpm.encoder, self.llm.model = nn.Module(), nn.Module(), nn.Module()
            self.vpm.embeddings.patch_embedding = nn.Conv2d(3, d, kernel_size=4, stride=4)
            self.vpm.encoder.layers = nn.ModuleList(VisLayer(d) for _ in range(2))
            self.resampler = Resampler(d)
            self.llm.model.layers = nn.ModuleList(LlmLayer(d) for _ in range(2))

        def forward(self, pixels):
            x = self.vpm.embeddings.patch_embedding(pixels).flatten(1)
            for layer in self.vpm.encoder.layers:
                x = layer(x)[0]
            x = self.resampler(x)
            for layer in self.llm.model.layers:
                x = layer(x, position_ids=None)[0]
            return x

    class Vlm:
        def __init__(self):
            torch.manual_seed(0)
            self.model = Hf()

        def generate(self, message, dataset):
            g = torch.Generator().manual_seed(int(message))
            return self.model(torch.randn(12, 3, 4, 4, generator=g))

    vlm = Vlm()
    args = gptq_args()
    qu.minicpmv_add_act_qaunt(vlm.model, args)
    q = minicpmv_gptq_plus.minicpmv_rtn_gptq_fwrd_plus(vlm, ToyDataset(6), "cpu", "toy", args)
    wrappers = qu.find_qlayers(vlm.model, [qu.ActQuantWrapper])
    assert len(q) == len(wrappers) == 1 + 2 * 6 + 6 + 2 * 7
    assert {"model.vpm.embeddings.patch_embedding", "model.vpm.encoder.layers.0.self_attn.out_proj.module",
            "model.resampler.proj_fc.module", "model.llm.model.layers.1.mlp.down_proj.module"} <= set(q)
    assert all(_on_grid(w.module.weight.data, w.weight_quantizers["module"]) for w in wrappers.values())


def _llm_wrappers_after_gptq(**over):
    from fake_quant import gptq, quant_utils as qu
    vlm = ToyVlm("qwen2vl")
    args = gptq_args(quant_visual_clip=False, quant_cross_attention=False, **over)
    qu.qwen2vl_add_act_qaunt(vlm, args)
    q = gptq.qwen2vl_rtn_gptq_fwrd_plus(vlm, ToyDataset(4), "cpu", "toy", args)   # the name exam/quant_qwen2vl.py calls
    assert q
    wrappers = qu.find_qlayers(vlm.model, [qu.ActQuantWrapper])
    llm = [w for n, w in wrappers.items() if ".layers." in n]
    assert llm
    return llm


def test_group_wise_gptq_attaches_every_groups_scale():
    """--w_groupsize > 0 (static_groups=False): the reference's quantizer only remembers its LAST column group
    (gptq_utils.py:263-273).  The solver here keeps every group's scale (``group_scales`` [rows, groups]) and attaches the
    quantizer, so the wrappers can run mq_gemm_w4a8_wgroupscale; every weight is on ITS group's grid."""
    llm = _llm_wrappers_after_gptq(w_groupsize=8)
    assert all(w.weight_quantizers for w in llm)
    for w in llm:
        for sub, wq in w.weight_quantizers.items():
            W = getattr(w, sub).weight.data.float()
            assert wq.groupsize == 8 and not wq.group_permuted
            assert tuple(wq.group_scales.shape) == (W.shape[0], W.shape[1] // 8)
            np.testing.assert_array_equal(wq.group_scales[:, -1].numpy(), wq.scale.reshape(-1).float().numpy())   # what upstream keeps
            lv = W.reshape(W.shape[0], -1, 8) / wq.group_scales[:, :, None]
            np.testing.assert_allclose(lv.numpy(), np.rint(lv.numpy()), atol=1e-3)
            assert float(lv.abs().max()) <= 8.0
        # groups of 8 are below what the kernels take: the wrapper says it simulates, and why
        assert "weight group size 8" in w._simulated_because() or w._simulated_because()


def test_group_wise_gptq_with_act_order_keeps_the_permutation_with_the_scales():
    """--act_order + --w_groupsize: the groups are runs of permuted columns; the solver keeps the permutation next to every group's
    scale (the integer backend gathers the activation columns the same way).  Groups of 8 are below the kernels' 64: the wrapper
    says so instead of silently simulating."""
    llm = _llm_wrappers_after_gptq(w_groupsize=8, act_order=True)
    for w in llm:
        for wq in w.weight_quantizers.values():
            K = w.module.in_features
            assert wq.group_permuted and wq.group_scales is not None and tuple(wq.group_scales.shape)[1] == K // 8
            assert sorted(wq.group_perm.tolist()) == list(range(K))
            # the stored weight, gathered into the solver's order, sits on the groups' grids
            Wp = w.module.weight.data.float()[:, wq.group_perm]
            lv = torch.round(Wp.reshape(Wp.shape[0], -1, 8) / wq.group_scales[:, :, None])
            assert float((lv * wq.group_scales[:, :, None] - Wp.reshape(Wp.shape[0], -1, 8)).abs().max()) < 1e-5
    w = llm[0]
    w.quantizer.configure(bits=8, sym=True)
    assert "group size 8" in w._simulated_because() and "simulated" in w.backend() and "Backend:" in w.extra_repr()
    assert not w._real_ready(torch.zeros(2, w.module.in_features))


def test_per_channel_gptq_still_attaches():
    assert all(w.weight_quantizers for w in _llm_wrappers_after_gptq())
