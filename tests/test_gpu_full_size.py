"""Size-independent properties at the FULL sizes of the benchmark configuration (Qwen2-VL-7B,
M_llm = 768): checksums of checksums, one-hot decoding of the pre-tiled weight image through the
GEMM, sampled exact dot products, transform round trips, quantizer idempotence.  The oracle cannot
run these sizes in seconds; the properties are exact integer identities (or fp32 round trips)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
torch.set_grad_enabled(False)

FULL = [("gate_up", 768, 37888, 3584), ("down_proj split-K", 768, 3584, 19968), ("vit fc1", 1024, 5120, 1280),
        ("72B gate_up", 768, 59136, 8192), ("72B down_proj", 768, 8192, 30720)]          # BASELINE config 5 shapes


def _levels(N, K, bits, seed):
    g = torch.Generator(device=DEV).manual_seed(seed)
    lim = 1 << (bits - 1)
    return torch.randint(-lim, lim, (N, K), generator=g, device=DEV, dtype=torch.int8)


@pytest.mark.parametrize("name,M,N,K", FULL)
@pytest.mark.parametrize("bits", [4, 8])
def test_gemm_checksums_and_sampled_dot_products(name, M, N, K, bits):
    from mquant_amd import ops
    if bits == 8 and N > 20000:
        pytest.skip("one W8 case at this size is enough")
    g = torch.Generator(device=DEV).manual_seed(N + K)
    a = torch.randint(-128, 128, (M, K), generator=g, device=DEV, dtype=torch.int8)
    w = _levels(N, K, bits, K)
    acc = ops.gemm_w4a8_i32(a, ops.prepack(w, bits), bits, N)
    a64, w64 = a.to(torch.int64), w.to(torch.int64)
    # checksum of checksums: row sums and column sums of the product from the factors' sums
    assert torch.equal(acc.to(torch.int64).sum(1), (a64 * w64.sum(0)[None, :]).sum(1))
    assert torch.equal(acc.to(torch.int64).sum(0), (w64 * a64.sum(0)[None, :]).sum(1))
    # 4096 sampled entries, exact
    mi = torch.randint(0, M, (4096,), generator=g, device=DEV)
    ni = torch.randint(0, N, (4096,), generator=g, device=DEV)
    assert torch.equal(acc[mi, ni].to(torch.int64), (a64[mi] * w64[ni]).sum(1))
    # linearity in the activations (integer arithmetic: exact)
    half = (a // 2).to(torch.int8)
    rest = (a - half).to(torch.int8)
    img = ops.prepack(w, bits)
    assert torch.equal(ops.gemm_w4a8_i32(half, img, bits, N) + ops.gemm_w4a8_i32(rest, img, bits, N), acc)


@pytest.mark.parametrize("name,M,N,K", FULL[:2])
def test_weight_image_decodes_through_the_gemm(name, M, N, K):
    """One-hot activation rows read single columns of the pre-tiled image back: acc[m, :] = w[:, k_m]."""
    from mquant_amd import ops
    w = _levels(N, K, 4, 3)
    img = ops.prepack(w, 4)
    g = torch.Generator(device=DEV).manual_seed(1)
    ks = torch.randint(0, K, (M,), generator=g, device=DEV)
    ks[:4] = torch.tensor([0, 1, K - 2, K - 1], device=DEV)
    a = torch.zeros((M, K), dtype=torch.int8, device=DEV)
    a[torch.arange(M, device=DEV), ks] = 1
    acc = ops.gemm_w4a8_i32(a, img, 4, N)
    assert torch.equal(acc, w[:, ks].t().to(torch.int32))
    # and the wire format round trip at this size
    assert torch.equal(ops.unpack_i4(ops.pack_i4(w)), w)


@pytest.mark.parametrize("n_in,n", [(18944, 19968), (5120, 5120), (29568, 30720)])
def test_hadamard_round_trip_and_norm(n_in, n):
    from fake_quant import hadamard_utils as hu
    from mquant_amd import ops
    M = 768
    hadK, K = hu.get_hadK(n)
    hadKt, _ = hu.get_hadK(n, transpose=True)
    g = torch.Generator(device=DEV).manual_seed(5)
    x = torch.randn((M, n_in), generator=g, device=DEV)
    y = ops.hadamard(x, n, K, hu._bits_for(hadK, K, torch.device(DEV)))
    # orthogonal: norms preserved (zero padding adds nothing)
    torch.testing.assert_close(y.double().pow(2).sum(1), x.double().pow(2).sum(1), rtol=1e-5, atol=0)
    # (H_K (x) H_m)^T = H_K^T (x) H_m undoes it
    back = ops.hadamard(y, n, K, hu._bits_for(hadKt, K, torch.device(DEV)))
    torch.testing.assert_close(back[:, :n_in], x, rtol=0, atol=2e-5)
    assert float(back[:, n_in:].abs().max()) < 2e-5 if n > n_in else True


def test_quantizers_are_idempotent_at_full_size():
    from mquant_amd import ops
    g = torch.Generator(device=DEV).manual_seed(9)
    x = (torch.randn((768, 18944), generator=g, device=DEV) * 3).half()
    s = 0.043
    q, _ = ops.quantize_act_i8(x, s)
    assert q.shape == (768, 18944) and int(q.abs().max()) <= 128
    deq = (q.float() * s)
    assert float((deq - x.float()).abs().max()) <= s / 2 * 1.001 or float(x.abs().max()) > 127 * s
    q2, _ = ops.quantize_act_i8(deq, s)
    assert torch.equal(q2, q)                                        # the grid is a fixed point
    fq = ops.fakequant_act(x, s)
    assert torch.equal(ops.fakequant_act(fq, s), fq)
    qd, sr, _ = ops.quantize_act_dyn_i8(x)
    assert int(qd.abs().max()) == 127 and torch.equal(qd.abs().amax(1), torch.full((768,), 127, device=DEV, dtype=torch.int8))


@pytest.mark.parametrize("name,M,N,K", [FULL[0], FULL[1], FULL[3]])
@pytest.mark.parametrize("g,mode", [(128, "w"), (64, "wx"), (256, "x")])
def test_group_scale_gemm_at_full_size(name, M, N, K, g, mode):
    """Group-wise scales at the benchmark's full sizes: the fold inside the wave-specialised tiles (csrc/gemm_ws.hip) against the
    round-1 grouped kernel (an independent implementation of the same arithmetic: other tile, other fold, conversions instead of the
    magic bias, scales by global loads) bit for bit, and 2048 sampled outputs against the arithmetic restated on the host."""
    from mquant_amd import ops
    gen = torch.Generator(device=DEV).manual_seed(N + K + g)
    a = torch.randint(-128, 128, (M, K), generator=gen, device=DEV, dtype=torch.int8)
    w = _levels(N, K, 4, K + g)
    G = K // g
    s_wg = torch.rand((G, N), generator=gen, device=DEV) * 0.01 + 0.001
    s_xg = torch.rand((M, G), generator=gen, device=DEV) * 0.2 + 0.01
    s_w = torch.rand((N,), generator=gen, device=DEV) * 0.01 + 0.001
    img = ops.prepack(w, 4)
    at = ops.TiledAct.from_rows(a)

    def run():
        if mode == "w":
            return ops.gemm_w4a8_wgroupscale(at, img, 4, N, s_wg, g, s_x0=0.031, out_dtype=torch.float32)
        if mode == "wx":
            return ops.gemm_w4a8_wgroupscale(at, img, 4, N, s_wg, g, s_x_groups=s_xg, out_dtype=torch.float32)
        return ops.gemm_w4a8_groupscale(at, img, 4, N, s_xg, g, s_w, out_dtype=torch.float32)
    y = run()
    try:
        ops.gemm_debug_force(26, 0)
        y_round1 = run()
    finally:
        ops.gemm_debug_force(-1, 0)
    assert torch.equal(y, y_round1)
    mi = torch.randint(0, M, (2048,), generator=gen, device=DEV)
    ni = torch.randint(0, N, (2048,), generator=gen, device=DEV)
    acc = (a[mi].to(torch.int64).reshape(-1, G, g) * w[ni].to(torch.int64).reshape(-1, G, g)).sum(2)      # exact group sums [2048, G]
    f = torch.zeros(2048, device=DEV, dtype=torch.float32)
    for gi in range(G):                                              # one fp32 rounding per product and per sum, ascending groups
        t = acc[:, gi].to(torch.float32)
        if mode != "w":
            t = t * s_xg[mi, gi]
        if mode != "x":
            t = t * s_wg[gi, ni]
        f = f + t
    want = f * (s_w[ni] if mode == "x" else (torch.tensor(0.031, device=DEV) if mode == "w" else torch.tensor(1.0, device=DEV)))
    assert torch.equal(y[mi, ni], want)
