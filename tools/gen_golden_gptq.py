#!/usr/bin/env python3
"""Goldens for the GPTQ solver: the REFERENCE's ``GPTQ`` / ``GPTQConv``
(fake_quant/gptq/gptq_utils.py:15-310) on CPU, small layers, seeded inputs.
Writes tests/golden/gptq_<case>.npz.  Build-container only.

Bridged for the import/run (nothing of the reference is replaced): the absent third-party modules
``fast_hadamard_transform`` (same stand-in as tools/gen_golden.py) and ``unfoldNd`` (only used for
Conv3d, not exercised here), and ``torch.cuda.synchronize`` (called unconditionally at
gptq_utils.py:287; there is no device in the build container)."""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_golden  # noqa: E402
from golden_inputs import make_w, make_x  # noqa: E402

# name: (kind, out, in, bits, mse, actorder, groupsize)
CASES = {
    "plain": ("linear", 40, 96, 4, False, False, -1),
    "actorder": ("linear", 40, 96, 4, False, True, -1),
    "groups": ("linear", 24, 128, 4, False, False, 32),
    "mse_w8": ("linear", 24, 160, 8, True, False, -1),
    "wide": ("linear", 16, 300, 4, False, False, -1),        # more than two 128-column blocks
    "conv2d": ("conv", 20, 3 * 4 * 4, 4, False, False, -1),
}


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


def main():
    gen_golden._install_shims()
    unf = types.ModuleType("unfoldNd")
    unf.UnfoldNd = object
    sys.modules["unfoldNd"] = unf
    torch.cuda.synchronize = lambda *a, **k: None
    torch.set_grad_enabled(False)
    from fake_quant import quant_utils as qu
    from fake_quant.gptq import gptq_utils as gu
    assert gu.__file__.startswith(gen_golden.REF)
    for i, (name, (kind, n_out, n_in, bits, mse, actorder, groupsize)) in enumerate(CASES.items()):
        seed = 900 + 10 * i
        layer, xs = build(kind, n_out, n_in, seed)
        solver = (gu.GPTQ if kind == "linear" else gu.GPTQConv)(layer)
        solver.quantizer = qu.WeightQuantizer()
        solver.quantizer.configure(bits, perchannel=True, sym=True, mse=mse)
        for x in xs:
            solver.add_batch(x, layer(x))
        H = solver.H.clone()
        solver.fasterquant(percdamp=0.01, groupsize=groupsize, actorder=actorder, static_groups=False)
        gen_golden.save(f"gptq_{name}", meta=np.array([seed, n_out, n_in, bits, int(mse), int(actorder), groupsize], np.int64),
                        H=(H.numpy() if n_in <= 128 else np.zeros((0, 0), np.float32)), Q=layer.weight.data.reshape(n_out, -1).numpy(),
                        scale=solver.quantizer.scale.reshape(-1).float().numpy())


if __name__ == "__main__":
    main()
