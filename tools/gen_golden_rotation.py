#!/usr/bin/env python3
"""Golden weights for the LayerNorm-fusion + rotation passes: runs the REFERENCE's
fuse_*_layer_norms / rotate_*_model (read-only at /root/reference) on the toy models of
tests/toy_models.py and stores the resulting state_dict and logits in
tests/golden/rotation_{qwen2vl,internvl,qwenvl,minicpmv}.npz.  Build-container only.

Two things the reference needs that this box lacks are bridged for the run, nothing else:
the `fast_hadamard_transform` import (same stand-in as tools/gen_golden.py) and a CUDA device
(the reference hard-codes ``.cuda()`` in apply_exact_had_to_linear / matmul_hadU_cuda, which is
mapped to a no-op so that the same torch ops run on the CPU).  ``internvl_rotation`` uses
``utils`` without importing it (NameError after the ViT loop); the module global is supplied.
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_golden  # noqa: E402

SEED, ROT_SEED = 7, 123


def main():
    gen_golden._install_shims()
    torch.set_grad_enabled(False)
    torch.Tensor.cuda = lambda self, *a, **k: self
    from fake_quant import hadamard_utils as hu
    from fake_quant import internvl_rotation, qwen2vl_rotation
    from fake_quant import utils as ref_utils
    assert hu.__file__.startswith(gen_golden.REF)
    if not hasattr(internvl_rotation, "utils"):
        internvl_rotation.utils = ref_utils
    import toy_models                      # its forward imports fake_quant.hadamard_utils lazily -> reference's

    from fake_quant import minicpmv_rotation, rotation_utils
    for kind, fuse, rotate in (
            ("minicpmv", minicpmv_rotation.fuse_minicpmv_layer_norms, minicpmv_rotation.rotate_minicpmv_model),
            ("qwen2vl", qwen2vl_rotation.fuse_qwen2vl_layer_norms, qwen2vl_rotation.rotate_qwen2vl_model),
            ("internvl", internvl_rotation.fuse_internvl_layer_norms, internvl_rotation.rotate_internvl2_model),
            ("qwenvl", rotation_utils.fuse_qwenvl_layer_norms, rotation_utils.rotate_model)):
        model, pixels, ids = toy_models.build(kind, SEED)
        args = toy_models.rotation_args()
        torch.manual_seed(ROT_SEED)
        fuse(model if kind in ("qwenvl", "minicpmv") else types.SimpleNamespace(model=model), args)
        rotate(model, args)
        model.online_visual = model.online_llm = True
        logits = model(pixels, ids)
        arrs = {k: v.numpy() for k, v in model.state_dict().items()}
        gen_golden.save(f"rotation_{kind}", seed=np.int64(SEED), rot_seed=np.int64(ROT_SEED),
                        logits=logits.numpy(), **arrs)


if __name__ == "__main__":
    main()
