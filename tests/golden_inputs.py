"""Deterministic synthetic inputs shared by tools/gen_golden.py and the tests.

Large tensors are never stored in tests/golden/: both sides rebuild them from a
seed with numpy's legacy RandomState (bit-stable across numpy versions).
"""
import numpy as np


def make_x(seed, shape, outlier_frac=0.001, outlier_gain=20.0, dtype=np.float32):
    """Activations ~ N(0,1) with a few outlier channels."""
    rs = np.random.RandomState(seed)
    x = rs.standard_normal(shape).astype(np.float32)
    c = shape[-1]
    n_out = max(1, int(round(c * outlier_frac)))
    idx = rs.choice(c, size=n_out, replace=False)
    x[..., idx] *= np.float32(outlier_gain)
    return x.astype(dtype)


def make_w(seed, shape, std=0.02, dtype=np.float32):
    """Weights ~ N(0, std^2)."""
    rs = np.random.RandomState(seed)
    return (rs.standard_normal(shape) * std).astype(np.float32).astype(dtype)


def make_ties(scale, levels):
    """Exact x = (k + 0.5) * scale values: round-half-even probes."""
    k = np.asarray(levels, dtype=np.float32)
    return ((k + np.float32(0.5)) * np.float32(scale)).astype(np.float32)


def make_wquant_weights(seed, N=40, K=768):
    """Input of the wquant_* goldens (tools/gen_golden_wquant.py): fp32 before the dtype cast."""
    w = make_w(seed, (N, K))
    w[3] = 0.0
    w[5, 7] = 0.9
    w[6] *= 40.0
    return w
