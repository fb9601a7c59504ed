#!/usr/bin/env python3
"""Goldens for the on-device weight quantizer: the REFERENCE's
``WeightQuantizer.find_params`` + ``quantize`` (fake_quant/quant_utils.py:415-524) run on CPU
tensors of the dtypes the two weight passes feed it -- fp32 (GPTQ path, ``W.float()``) and
fp16 / bf16 (RTN path, the module's own dtype) -- with and without the MSE clip search.
Writes tests/golden/wquant_<dtype>_<bits>b_<rtn|mse>.npz.  Build-container only."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_golden  # noqa: E402
from golden_inputs import make_wquant_weights  # noqa: E402

N, K = 40, 768
CASES = [("f32", 4, 0), ("f32", 4, 1), ("f32", 8, 0), ("f32", 8, 1), ("f16", 4, 0), ("f16", 4, 1),
         ("bf16", 4, 0), ("bf16", 4, 1), ("f16", 8, 0)]
DT = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}


def weights(seed, dtype):
    # fp32 ~N(0, 0.02); row 3 dead (scale clamps at 1e-5), row 5 one outlier (the clip search cuts
    # it), row 6 scaled x40
    return torch.from_numpy(make_wquant_weights(seed, N, K)).to(dtype)


def main():
    gen_golden._install_shims()
    torch.set_grad_enabled(False)
    from fake_quant import quant_utils as qu
    assert qu.__file__.startswith(gen_golden.REF)
    for i, (dt, bits, mse) in enumerate(CASES):
        w = weights(500 + i, DT[dt])
        q = qu.WeightQuantizer()
        q.configure(bits, perchannel=True, sym=True, mse=bool(mse))
        q.find_params(w)
        wq = q.quantize(w)
        gen_golden.save(f"wquant_{dt}_{bits}b_{'mse' if mse else 'rtn'}",
                        meta=np.array([500 + i, bits, mse], np.int64),
                        scale=q.scale.float().numpy().reshape(-1), wq=wq.float().numpy())


if __name__ == "__main__":
    main()
