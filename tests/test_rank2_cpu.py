"""Flag combinations that need the rank-1 epilogue term twice (--w_asym + --visual_split, --w_asym + --a_asym, --a_asym +
--visual_split): the integer restatement -- exact int32 accumulators over the stored levels plus two rank-1 terms -- against
the REFERENCE's own forward in those modes (tests/golden/wrapper_rank2_*.npz, tools/gen_golden_rank2.py).

  stored activation level a = q_a - 2^(ba-1) (asymmetric activations) or q_a (symmetric); stored weight level p = q_w - 2^(bw-1) or q_w
  x~[m][k] = s_a[m] (a + ha - z_a[m]),  W~[n][k] = s_w[n] (p + hw - z_w[n])          (ha = 2^(ba-1) or 0 with z_a = 0; same for hw)
  sum_k x~ W~ = s_a s_w acc                                   acc = sum_k a p   (int32, exact)
              + (s_a (ha - z_a))[m] * (s_w (sum_k p + Kq (hw - z_w)))[n]       activation zero points (Kq = quantized columns)
              + (s_a sum_k a)[m] * (s_w (hw - z_w))[n]                          weight zero points
              + x[m][0] * w0[n]                                                 the split column, in floating point
"""
import glob
import os

import numpy as np
import pytest

import oracle
from golden_inputs import make_w, make_x

ACT = {0: "static", 1: "dyn_sym", 2: "dyn_asym", 3: "pt_asym"}


def rank2_cases(golden_dir):
    return sorted(glob.glob(os.path.join(golden_dir, "wrapper_rank2_*.npz")))


def rank1_terms(g, s_rows, zero, qx_sum, qw_sum, Kq, w_sym, a_asym, split):
    """The (x[m], w[n]) pairs of the epilogue, in the order the engine applies them."""
    s_w = g["s_w"]
    terms = []
    if split:
        terms.append((g["x0"].astype(np.float32), g["w0"].astype(np.float32)))
    w_shift = None if w_sym else (s_w * (np.float32(8.0) - g["z_w"])).astype(np.float32)
    if a_asym:
        shift = (s_rows * (np.float32(128.0) - zero)).astype(np.float32)
        colsum = (qw_sum.astype(np.float32) * s_w).astype(np.float32)
        if w_shift is not None:
            colsum = (colsum + (np.float32(Kq) * w_shift).astype(np.float32)).astype(np.float32)
        terms.append((shift, colsum))
    if w_shift is not None:
        terms.append(((s_rows * qx_sum.astype(np.float32)).astype(np.float32), w_shift))
    return terms


def test_there_are_goldens(golden_dir):
    assert len(rank2_cases(golden_dir)) == 9                     # 7 two-term combinations + 2 with all three terms (round 5)


@pytest.mark.parametrize("path", rank2_cases(os.path.join(os.path.dirname(__file__), "golden")))
def test_two_rank1_terms_reproduce_the_reference_forward(had_table, path):
    g = np.load(path)
    K_in, K_pad, N, M, seed, had, split, bias, w_sym, act = [int(v) for v in g["meta"]]
    act = ACT[act]
    # weights: the oracle's quantizers are pinned to the reference's elsewhere; here the fixture's own scales / zero points
    W = make_w(seed, (N, K_pad))
    Wq = W[:, 1:] if split else W
    if w_sym:
        s_w, levels = oracle.wquant_sym(Wq, bits=4)
    else:
        s_w, z_w, levels = oracle.wquant_asym(Wq, bits=4)
        np.testing.assert_array_equal(z_w, g["z_w"])
    np.testing.assert_array_equal(s_w, g["s_w"])
    np.testing.assert_array_equal(levels.astype(np.int64).sum(axis=1), g["qw_sum"])
    x = make_x(seed + 20, (M, K_in))
    if K_pad != K_in and not had:
        x = np.pad(x, ((0, 0), (0, K_pad - K_in)))
    if had:
        K = had_table["n2k"][K_pad]
        x = oracle.hadamard(x, K_pad, K, had_table["mats"][K], mid_round=0, out_round=0)
    xq = x[:, 1:] if split else x
    a_asym = act in ("dyn_asym", "pt_asym")
    zero = np.zeros(M, np.float32)
    if act == "static":
        s_rows = np.full(M, np.float32(g["s_x"]), np.float32)
        q = oracle.quant_static(xq, np.float32(g["s_x"]))
    elif act == "dyn_sym":
        q, s_rows = oracle.quant_dyn(xq, bits=8, clip=1.0)
    elif act == "dyn_asym":
        q, s_rows, zero, _ = oracle.quant_dyn_asym(xq, bits=8, clip=1.0)
    else:
        q, s, z, _ = oracle.quant_tensor(xq, bits=8, clip=1.0, asym=True)
        s_rows, zero = np.full(M, s, np.float32), np.full(M, z, np.float32)
    if act != "static":
        np.testing.assert_array_equal(s_rows, g["s_rows"])
    if a_asym:
        np.testing.assert_array_equal(zero, g["zero"])
    np.testing.assert_array_equal(q[:, :64], g["qx_head"])
    acc = oracle.gemm_i32(q, levels)
    np.testing.assert_array_equal(acc, g["acc"])
    np.testing.assert_array_equal(q.astype(np.int64).sum(axis=1), g["qx_sum"])
    terms = rank1_terms(g, s_rows, zero, g["qx_sum"], g["qw_sum"], xq.shape[1], bool(w_sym), a_asym, bool(split))
    assert len(terms) == int(bool(split)) + int(not w_sym) + int(a_asym) and len(terms) in (2, 3), path
    b = make_w(seed + 1, (N,), std=0.1) if bias else None
    y = oracle.epilogue(acc, s_rows, s_w, bias=b, x0=terms[0][0], w0=terms[0][1], x1=terms[1][0], w1=terms[1][1])
    if len(terms) == 3:        # the third term continues the fp32 sum (mq_rank1_add_cast behind the two-slot GEMM)
        y = (y + (terms[2][0].reshape(-1, 1) * terms[2][1][None, :]).astype(np.float32)).astype(np.float32)
    np.testing.assert_allclose(y, g["y"], rtol=0, atol=1e-3)
