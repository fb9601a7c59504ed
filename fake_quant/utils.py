"""Small runtime helpers used by the drivers (reference: ``fake_quant/utils.py``).

Only the live part of the upstream file is mirrored: ``DEV``, ``seed_everything``,
``cleanup_memory`` and the ``revise_down_input`` forward-pre-hook.  The QuaRot argument
parser and the accelerate-based ``distribute_model`` have no caller upstream.
"""
import gc
import logging
import random
from typing import Optional

import numpy as np
import torch

# TF32-style shortcuts stay off, as upstream (fake_quant/utils.py:26-27).
torch.backends.cuda.matmul.allow_tf32 = False
torch.backends.cudnn.allow_tf32 = False

DEV = torch.device("cuda:0") if torch.cuda.is_available() else torch.device("cpu")


def seed_everything(seed: Optional[int] = None) -> int:
    """Seed python / numpy / torch (all devices); returns the seed used."""
    if seed is None:
        seed = random.randint(0, 2 ** 32 - 1)
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
        torch.backends.cudnn.deterministic = True
        torch.backends.cudnn.benchmark = False
    return seed


def cleanup_memory(verbos: bool = True) -> None:
    """gc + empty the caching allocator; logs the reserved-memory delta when ``verbos``."""
    def reserved():
        return sum(torch.cuda.memory_reserved(i) for i in range(torch.cuda.device_count()))

    have_gpu = torch.cuda.is_available()
    before = reserved() if have_gpu else 0
    gc.collect()
    if have_gpu:
        torch.cuda.empty_cache()
        if verbos:
            after = reserved()
            gib = 1024 ** 3
            logging.info("GPU memory: %.2f -> %.2f GB (%.2f GB)", before / gib, after / gib,
                         (after - before) / gib)


def revise_down_input(m, i, new_size):
    """forward-pre-hook: zero-pad the last dim of the first positional input to ``new_size``.

    On the real-quant path ``ActQuantWrapper`` recognises this hook and folds the padding
    into the fused Hadamard kernel instead of materialising the padded tensor.
    """
    x = i[0]
    pad = new_size - x.shape[-1]
    return (torch.nn.functional.pad(x, (0, pad)),) + tuple(i[1:])


def tensor_version(t) -> int:
    """``t._version`` (the in-place write counter); inference tensors keep none and count as version 0 -- they CAN be
    rewritten in place inside ``torch.inference_mode()``, and writes through ``.data`` or through this repository's own
    in-place kernels bump no counter either, so callers must not rely on the counter alone (``SiblingGroup`` also drops its
    cache at every forward pass of the parent module)."""
    try:
        return t._version
    except RuntimeError:
        return 0
