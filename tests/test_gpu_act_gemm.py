"""mq_gemm_w4a8_act_ws: the consumer's activation in the PRODUCER GEMM's store (round 6, SURVEY 8(f3) taken to its end).

The activations are HF model code around the reference's wrapped Linears (fake_quant/quant_utils.py:330-391 evaluates each
Linear; Qwen2MLP computes down_proj(act_fn(gate_proj(x)) * up_proj(x)), the vision MLP fc2(quick_gelu(fc1(x)))).  The bar:
 * bit for bit the plain GEMM launch (already pinned to the oracle's int32 accumulators and epilogue) followed by the torch ops
   on its rounded output -- the fused store calls the same device expf;
 * the C oracle's epilogue + activation (library-free exp) up to its last-bit freedom."""
import numpy as np
import pytest
import torch

import oracle
from golden_inputs import make_x

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
torch.set_grad_enabled(False)
MODE = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}


def _case(M, N, K, seed=0, bias=True):
    from mquant_amd import ops
    g = torch.Generator(device=DEV).manual_seed(seed)
    levels = torch.randint(-8, 8, (N, K), generator=g, device=DEV, dtype=torch.int8)
    s_w = (torch.rand((N,), generator=g, device=DEV) * 0.004 + 0.001).float()
    b = (torch.randn((N,), generator=g, device=DEV) * 0.3).float() if bias else None
    x = torch.from_numpy(make_x(seed + 5, (M, K))).to(DEV).half()
    sel = (torch.arange(M, device=DEV) >= M // 3).to(torch.uint8)
    s0, s1 = 0.031, 0.047
    a, _ = ops.quantize_act_i8(x, s0, s1, row_sel=sel, tiled=True)
    img = ops.prepack(levels, 4)
    return ops, levels, s_w, b, a, img, sel, s0, s1


def _torch_act(y, act, ops):
    """The HF activation on the half tensor ``y`` with the roundings of the reference's CPU run: every torch op computes in fp32 and
    rounds ONCE to y's dtype.  Written with explicit fp32 intermediates: torch's own half kernels on this GPU fuse `a * b -> half`
    into one rounding (V_FMA_MIXLO_F16), which differs from the CPU's fp32-then-half on exact ties -- e.g. 1.702 * -0.18310546875
    (found by this test)."""
    dt = y.dtype
    f = y.float()
    if act == ops.ACT_SILU_MUL:
        H = y.shape[1] // 2
        s = torch.nn.functional.silu(f[:, :H]).to(dt)
        return (s.float() * f[:, H:]).to(dt)
    z = (1.702 * f).to(dt)                                   # QuickGELUActivation: x * sigmoid(1.702 x)
    sg = torch.sigmoid(z.float()).to(dt)
    return (f * sg.float()).to(dt)


def _torch_act_native(y, act, ops):
    """The same through torch's half kernels (what an fp16 model executes on this GPU): equal up to the tie cases above."""
    if act == ops.ACT_SILU_MUL:
        H = y.shape[1] // 2
        return torch.nn.functional.silu(y[:, :H]) * y[:, H:]
    return y * torch.sigmoid(1.702 * y)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16, torch.float32])
@pytest.mark.parametrize("act", [1, 2])
@pytest.mark.parametrize("tile", [-1, 14, 20, 19, 44, 45, 46, 47, 48, 41, 51, 52, 53, 54, (20, "slab"), (53, "slab"), (52, "slab")])
def test_act_store_equals_gemm_plus_torch_ops(dtype, act, tile):
    # (tile, "slab"): the slab form of the activation epilogue where a slab-free one exists (ping-pong silu, wave-specialised gelu)
    slab = isinstance(tile, tuple)
    tile = tile[0] if slab else tile
    M, N, K = 300, 448, 384            # N / 2 = 224 = 7 x 32: ragged against every tile width; M ragged against every tile height
    ops, levels, s_w, b, a, img, sel, s0, s1 = _case(M, N, K, seed=tile + 3 * act)
    y = ops.gemm_w4a8(a, img, 4, N, s0, s_w, s_x1=s1, row_sel=sel, bias=b, out_dtype=dtype)
    want = _torch_act(y, act, ops)
    ops.gemm_debug_force(tile, (1 << 16) if slab else 0)
    try:
        got = ops.gemm_w4a8_act(a, img, 4, N, s0, s_w, act, s_x1=s1, row_sel=sel, bias=b, out_dtype=dtype)
    finally:
        ops.gemm_debug_force(-1, 0)
    assert got.shape == want.shape and got.dtype == dtype
    assert torch.equal(got, want), float((got.float() - want.float()).abs().max())
    native = _torch_act_native(y, act, ops)
    off = got != native
    assert float(off.float().mean()) < 1e-3
    if bool(off.any()):                                       # one unit in the last place of the dtype, on ties only
        rel = (got.float() - native.float()).abs()[off] / native.float().abs()[off].clamp_min(1e-30)
        assert float(rel.max()) <= (2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10)


@pytest.mark.parametrize("act", [1, 2])
def test_act_store_against_the_oracle(act):
    M, N, K = 70, 192, 256
    ops, levels, s_w, b, a, img, sel, s0, s1 = _case(M, N, K, seed=11)
    got = ops.gemm_w4a8_act(a, img, 4, N, s0, s_w, act, s_x1=s1, row_sel=sel, bias=b, out_dtype=torch.float16).float().cpu().numpy()
    acc = oracle.gemm_i32(a.to_rows().cpu().numpy()[:, :K], levels.cpu().numpy())
    sx = np.where(sel.cpu().numpy() == 0, np.float32(s0), np.float32(s1)).astype(np.float32)
    y = oracle.round_to(oracle.epilogue(acc, sx, s_w.cpu().numpy(), bias=b.cpu().numpy()), 1)
    if act == 1:
        want = oracle.silu_mul(y[:, : N // 2].copy(), y[:, N // 2:].copy(), 1).reshape(M, N // 2)
    else:
        want = oracle.quick_gelu(y.copy(), 1).reshape(M, N)
    # the oracle's exp is correctly rounded, the device library's is not always: a last-bit freedom on a few elements
    bad = got != want
    assert bad.mean() < 2e-3, bad.mean()
    assert np.all(np.abs(got - want)[bad] <= np.maximum(np.abs(want[bad]) * 2.0 ** -10, 2.0 ** -24))


def test_full_size_gate_up_and_vision_fc1():
    """The two launches of the prefill: gate|up 768 x 37888 x 3584 (the 256 x 256 ping-pong tile, persistent over 444 ids) and the
    vision tower's fc1 1024 x 5120 x 1280 (wave-specialised 192 x 128)."""
    from mquant_amd import ops
    for (M, N, K, act, bias) in ((768, 37888, 3584, 1, False), (1024, 5120, 1280, 2, True)):
        ops_, levels, s_w, b, a, img, sel, s0, s1 = _case(M, N, K, seed=N % 97, bias=bias)
        y = ops.gemm_w4a8(a, img, 4, N, s0, s_w, s_x1=s1, row_sel=sel, bias=b, out_dtype=torch.float16)
        got = ops.gemm_w4a8_act(a, img, 4, N, s0, s_w, act, s_x1=s1, row_sel=sel, bias=b, out_dtype=torch.float16)
        assert torch.equal(got, _torch_act(y, act, ops))


def test_w8_image_and_per_row_scales():
    from mquant_amd import ops
    M, N, K = 130, 256, 256
    g = torch.Generator(device=DEV).manual_seed(3)
    levels = torch.randint(-128, 128, (N, K), generator=g, device=DEV, dtype=torch.int8)
    s_w = (torch.rand((N,), generator=g, device=DEV) * 0.0004 + 0.0001).float()
    x = torch.from_numpy(make_x(8, (M, K))).to(DEV).half()
    a, s_rows, _ = ops.quantize_act_dyn_i8(x, tiled=True)
    img = ops.prepack(levels, 8)
    y = ops.gemm_w4a8_rowscale(a, img, 8, N, s_rows, s_w, out_dtype=torch.float16)
    for act in (1, 2):
        got = ops.gemm_w4a8_act(a, img, 8, N, 1.0, s_w, act, s_x_rows=s_rows, out_dtype=torch.float16)
        assert torch.equal(got, _torch_act(y, act, ops))


def test_engine_gate_up_into_down_proj_equals_the_prologue_form(had_table):
    """A decoder MLP through the engines: gate|up storing silu(gate) * up, then down_proj's PLAIN Hadamard + quantize -- the same
    int8 levels and the same output as round 5's form (activation in the Hadamard kernel's prologue)."""
    from mquant_amd import ops
    from mquant_amd.engine import HadamardSpec, W4A8Linear
    M, D, H, Hp, K = 200, 256, 704, 768, 12      # intermediate 704 -> padded to 768 = 12 x 64 by the pad hook
    g = torch.Generator(device=DEV).manual_seed(1)
    w_gu = (torch.randn((2 * H, D), generator=g, device=DEV) * 0.05).half()
    w_dn = (torch.randn((D, Hp), generator=g, device=DEV) * 0.02).half()
    x = torch.from_numpy(make_x(4, (M, D))).to(DEV).half()
    words = torch.from_numpy(np.ascontiguousarray(had_table["words"][K])).to(DEV)
    gu = W4A8Linear.from_float(w_gu, 4, s_x0=0.03)
    dn = W4A8Linear.from_float(w_dn, 4, s_x0=0.02, had=HadamardSpec(Hp, K, words), in_features=H)
    a, _ = gu.quantize(x)
    assert gu.act_in_store_ok(ops.ACT_SILU_MUL)
    h = gu.gemm_act(a, ops.ACT_SILU_MUL, torch.float16)
    y_new = dn.forward(h)
    y2 = gu.gemm(a, None, torch.float16)
    qa, x0 = dn.quantize_act(y2[:, :H], y2[:, H:], ops.ACT_SILU_MUL)
    y_old = dn.gemm(qa, x0, torch.float16)
    assert torch.equal(h, torch.nn.functional.silu(y2[:, :H]) * y2[:, H:])
    assert torch.equal(y_new, y_old)


def test_argument_checks():
    from mquant_amd import ops
    from mquant_amd._lib import MQuantHipError
    ops_, levels, s_w, b, a, img, sel, s0, s1 = _case(40, 192, 128)
    with pytest.raises(MQuantHipError):
        ops.gemm_w4a8_act(a.to_rows(), img, 4, 192, s0, s_w, 1)                     # row-major activations
    with pytest.raises(MQuantHipError):
        ops.gemm_w4a8_act(a, img, 4, 192, s0, s_w, 7)                               # unknown activation
    ops2, levels2, s_w2, b2, a2, img2, *_ = _case(40, 96, 128)
    with pytest.raises(MQuantHipError):
        ops.gemm_w4a8_act(a2, img2, 4, 96, s0, s_w2, 1)                             # N / 2 = 48 is not a multiple of 32


def test_chained_prefill_logits_do_not_depend_on_where_the_activation_runs():
    """FullPrefill.act_in_gemm: the activation in the producing GEMM's store (round 6) against the Hadamard kernel's prologue
    (round 5) -- every rounding is the same, so the logits must be IDENTICAL."""
    from mquant_amd import workload
    from mquant_amd.full_prefill import FullPrefill
    specs = workload._qwen2vl_7b_specs(True, 2, 2)
    pf = workload.Prefill(specs, device=DEV, share_groups=True)
    outs = []
    for in_gemm in (False, True):
        fp = FullPrefill(pf, fused_glue=True)
        fp.act_in_gemm = in_gemm
        fp.calibrate()
        outs.append(fp.step().float().clone())
        fp.restore_hot_path_scales()
    assert torch.isfinite(outs[0]).all() and float(outs[0].abs().max()) > 0
    assert torch.equal(outs[0], outs[1])


def test_random_shapes_of_the_act_store():
    """Seeded sweep: ragged M, N / 2 any multiple of 32 (silu) or N any multiple of 8 (gelu), K any multiple of 128, W4 / W8, with and without
    bias, token-type scales or per-row scales, every kernel family the plan can send an activation to -- against GEMM + torch ops."""
    from mquant_amd import ops
    rng = np.random.default_rng(20261003)
    tiles = [-1, 14, 20, 19, 44, 45, 46, 47, 48, 51, 52, 53, 54]
    try:
        for case in range(60):
            act = 1 + case % 2
            w_bits = 8 if case % 5 == 0 else 4
            M = int(rng.integers(65, 600))
            N = int(rng.integers(1, 24)) * 64 if act == 1 else int(rng.integers(2, 160)) * 8
            K = int(rng.integers(1, 6)) * 128
            tile = int(rng.choice(tiles))
            if w_bits == 8 and tile in (14, 20, 19, 46, 52):
                tile = -1
            dtype = [torch.float16, torch.bfloat16, torch.float32][case % 3]
            g = torch.Generator(device=DEV).manual_seed(case)
            lim = 1 << (w_bits - 1)
            levels = torch.randint(-lim, lim, (N, K), generator=g, device=DEV, dtype=torch.int8)
            s_w = (torch.rand((N,), generator=g, device=DEV) * (0.004 if w_bits == 4 else 0.0003) + 0.0002).float()
            b = (torch.randn((N,), generator=g, device=DEV) * 0.3).float() if case % 3 else None
            x = torch.from_numpy(make_x(case, (M, K))).to(DEV).half()
            img = ops.prepack(levels, w_bits)
            if case % 4 == 0:
                a, s_rows, _ = ops.quantize_act_dyn_i8(x, tiled=True)
                y = ops.gemm_w4a8_rowscale(a, img, w_bits, N, s_rows, s_w, bias=b, out_dtype=dtype)
                kw = dict(s_x_rows=s_rows)
                s0 = 1.0
            else:
                sel = (torch.arange(M, device=DEV) % 3 == 0).to(torch.uint8)
                s0, s1 = 0.03, 0.05
                a, _ = ops.quantize_act_i8(x, s0, s1, row_sel=sel, tiled=True)
                y = ops.gemm_w4a8(a, img, w_bits, N, s0, s_w, s_x1=s1, row_sel=sel, bias=b, out_dtype=dtype)
                kw = dict(s_x1=s1, row_sel=sel)
            want = _torch_act(y, act, ops)
            ops.gemm_debug_force(tile, 0)
            got = ops.gemm_w4a8_act(a, img, w_bits, N, s0, s_w, act, bias=b, out_dtype=dtype, **kw)
            ops.gemm_debug_force(-1, 0)
            assert torch.equal(got, want), (case, act, w_bits, M, N, K, tile, dtype)
    finally:
        ops.gemm_debug_force(-1, 0)
